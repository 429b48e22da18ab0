// Developer tool: ablation timing of the pipelined fp16x3 3x3 kernels (kernels_conv3x3_x3.hip) on decoder-layer shapes.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/x3_ablate.hip -o tools/_x3_ablate
// ABL bits: 1 no global loads / LDS stores in the loop | 2 no MFMA | 4 no LDS fragment reads | 8 no barrier | 16 no epilogue | 256 __syncthreads()
//           (bit 512, the halo by LDS-DMA + zero page, was measured slower -- profiles/r02_x3_halo_dma_ab.txt -- and removed with its code path)
//   _x3_ablate        ablation table;  _x3_ablate c   clock probe;  _x3_ablate b   barrier A/B (lds_dma.hpp)
#include <cstdio>
#include <vector>

#include "../autoware_vision_pilot_amd/csrc/kernels_conv3x3_x3.hip"
#include "tool_ones.hpp"

using namespace vp;
// the tool's bias vector is all zeros (hipMemset): it doubles as the kernels' zero page
static const half_t* bias_zero_page(const float* bias) { return reinterpret_cast<const half_t*>(bias); }
namespace vp { hipError_t launch_splitk_finish(const ConvGemmParams&, hipStream_t) { return hipErrorInvalidValue; } }  // the tool never splits K

template <int TH, int WPX, bool HDB, int ABL>
static float time_variant(const ConvGemmParams& p, int iters) {
  constexpr int lds = (HDB ? 2 : 1) * 2 * ((TH + 2) * 18 * 80) + 6 * (128 * 64);
  auto k = conv3x3_x3_kernel<128, TH, 2, WPX, HDB, ACT_GELU, ABL>;
  hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  dim3 grid(((p.H + TH - 1) / TH) * ((p.W + 15) / 16) * (p.CoutW / 128));
  hipEvent_t a, b;
  hipEventCreate(&a);
  hipEventCreate(&b);
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(k, grid, dim3(128 * WPX), lds, 0, p);
  hipEventRecord(a, 0);
  for (int i = 0; i < iters; ++i) hipLaunchKernelGGL(k, grid, dim3(128 * WPX), lds, 0, p);
  hipEventRecord(b, 0);
  hipEventSynchronize(b);
  float ms = 0;
  hipEventElapsedTime(&ms, a, b);
  return ms * 1000.0f / iters;
}

template <int TH, int WPX, bool HDB>
static int run_shape(const char* name, int H, int W, int Cin, int Cout) {
  const size_t in_n = (size_t)H * W * Cin, out_n = (size_t)H * W * Cout, w_n = (size_t)9 * Cout * Cin;
  half_t *in, *inl, *out, *outl, *w, *wl;
  float* bias;
  hipMalloc(&in, in_n * 2); hipMalloc(&inl, in_n * 2); hipMalloc(&out, out_n * 2); hipMalloc(&outl, out_n * 2);
  hipMalloc(&w, w_n * 2); hipMalloc(&wl, w_n * 2); hipMalloc(&bias, Cout * 4);
  std::vector<half_t> h(in_n > w_n ? in_n : w_n);
  unsigned s = 12345;
  for (auto& v : h) {
    s = s * 1664525u + 1013904223u;
    v = (half_t)(((int)(s >> 9) % 2001 - 1000) * 0.001f);
  }
  hipMemcpy(in, h.data(), in_n * 2, hipMemcpyHostToDevice);
  hipMemcpy(w, h.data(), w_n * 2, hipMemcpyHostToDevice);
  for (auto& v : h) v = (half_t)((float)v * 0.0004f);
  hipMemcpy(inl, h.data(), in_n * 2, hipMemcpyHostToDevice);
  hipMemcpy(wl, h.data(), w_n * 2, hipMemcpyHostToDevice);
  hipMemset(bias, 0, Cout * 4);
  ConvGemmParams p{};
  p.in_hi = in; p.in_lo = inl; p.H = H; p.W = W; p.Cin = Cin; p.w_hi = w; p.w_lo = wl; p.bias = bias; p.wscale = tool_dev_ones(Cout); p.ks = 3; p.Ncols = Cout; p.CoutW = Cout;
  p.act = ACT_GELU; p.out_hi = out; p.out_lo = outl; p.Cstore = Cout; p.Creal = Cout; p.nsplit = 1; p.zeros = bias_zero_page(bias);
  const double gflop = 2.0 * H * W * (double)Cout * Cin * 9 / 1e9;
  const int it = 20;
  const float t0 = time_variant<TH, WPX, HDB, 0>(p, it), t16 = time_variant<TH, WPX, HDB, 16>(p, it), t1 = time_variant<TH, WPX, HDB, 1>(p, it),
              t4 = time_variant<TH, WPX, HDB, 4>(p, it), t8 = time_variant<TH, WPX, HDB, 8>(p, it), t2 = time_variant<TH, WPX, HDB, 2>(p, it),
              t29 = time_variant<TH, WPX, HDB, 1 | 4 | 8 | 16>(p, it), t21 = time_variant<TH, WPX, HDB, 1 | 4 | 16>(p, it),
              t17 = time_variant<TH, WPX, HDB, 1 | 16>(p, it), t20 = time_variant<TH, WPX, HDB, 4 | 16>(p, it),
              tA = time_variant<TH, WPX, HDB, 4096>(p, it), tA8 = time_variant<TH, WPX, HDB, 4096 | 8>(p, it);
  std::printf("%-30s %5.1f GF | full %6.1f us (%5.1f TF alg) | noEpi %6.1f | noGlobal %6.1f | noLdsRead %6.1f | noBarrier %6.1f | noMFMA %6.1f | "
              "noEpi+noGlobal %6.1f | noEpi+noLdsRead %6.1f | noEpi+noGlobal+noLdsRead %6.1f | MFMA+loop only %6.1f | weights global->registers (prototype) %6.1f, "
              "without the per-step barrier %6.1f\n",
              name, gflop, t0, gflop / t0 * 1e3, t16, t1, t4, t8, t2, t17, t20, t21, t29, tA, tA8);
  hipFree(in); hipFree(inl); hipFree(out); hipFree(outl); hipFree(w); hipFree(wl); hipFree(bias);
  return 0;
}

// sustained-load clock probe: 200 back-to-back launches, the last one's workgroup 0 reports shader-clock vs 100 MHz wall ticks of its K loop
template <int TH, int WPX, bool HDB, int ABL>
static void clock_probe(const char* name, int H, int W, int Cin, int Cout) {
  const size_t in_n = (size_t)H * W * Cin, out_n = (size_t)H * W * Cout, w_n = (size_t)9 * Cout * Cin;
  half_t *in, *inl, *out, *outl, *w, *wl;
  float* bias;
  unsigned long long* probe;
  hipMalloc(&in, in_n * 2); hipMalloc(&inl, in_n * 2); hipMalloc(&out, out_n * 2); hipMalloc(&outl, out_n * 2);
  hipMalloc(&w, w_n * 2); hipMalloc(&wl, w_n * 2); hipMalloc(&bias, Cout * 4); hipMalloc(&probe, 64);
  std::vector<half_t> h(in_n > w_n ? in_n : w_n);
  unsigned s = 777;
  for (auto& v : h) { s = s * 1664525u + 1013904223u; v = (half_t)(((int)(s >> 9) % 2001 - 1000) * 0.001f); }
  hipMemcpy(in, h.data(), in_n * 2, hipMemcpyHostToDevice); hipMemcpy(w, h.data(), w_n * 2, hipMemcpyHostToDevice);
  for (auto& v : h) v = (half_t)((float)v * 0.0004f);
  hipMemcpy(inl, h.data(), in_n * 2, hipMemcpyHostToDevice); hipMemcpy(wl, h.data(), w_n * 2, hipMemcpyHostToDevice);
  hipMemset(bias, 0, Cout * 4);
  ConvGemmParams p{};
  p.in_hi = in; p.in_lo = inl; p.H = H; p.W = W; p.Cin = Cin; p.w_hi = w; p.w_lo = wl; p.bias = bias; p.wscale = tool_dev_ones(Cout); p.ks = 3; p.Ncols = Cout; p.CoutW = Cout;
  p.act = ACT_GELU; p.out_hi = out; p.out_lo = outl; p.Cstore = Cout; p.Creal = Cout; p.nsplit = 1; p.partial = reinterpret_cast<float*>(probe); p.zeros = bias_zero_page(bias);
  constexpr int lds = (HDB ? 2 : 1) * 2 * ((TH + 2) * 18 * 80) + 6 * (128 * 64);
  auto k = conv3x3_x3_kernel<128, TH, 2, WPX, HDB, ACT_GELU, ABL>;
  hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  dim3 grid(((H + TH - 1) / TH) * ((W + 15) / 16) * (Cout / 128));
  hipEvent_t a, b;
  hipEventCreate(&a); hipEventCreate(&b);
  for (int i = 0; i < 50; ++i) hipLaunchKernelGGL(k, grid, dim3(128 * WPX), lds, 0, p);
  hipEventRecord(a, 0);
  for (int i = 0; i < 200; ++i) hipLaunchKernelGGL(k, grid, dim3(128 * WPX), lds, 0, p);
  hipEventRecord(b, 0);
  hipEventSynchronize(b);
  float ms = 0;
  hipEventElapsedTime(&ms, a, b);
  unsigned long long r[2] = {0, 0};
  hipMemcpy(r, probe, 16, hipMemcpyDeviceToHost);
  const double steps = (Cin / 32) * 9.0, mfma_cycles = steps * 2 * 24 * 32;  // two waves per SIMD, 24 MFMAs of 32 cycles per tap each
  std::printf("%-40s %6.1f us/launch sustained | K loop of workgroup 0: %llu shader ticks in %.2f us -> %.0f MHz; MFMA issue needs %.0f cycles = %.0f %% of them\n",
              name, ms * 1000.0f / 200, r[0], r[1] / 100.0, r[1] ? r[0] / (r[1] / 100.0) : 0.0, mfma_cycles, r[0] ? 100.0 * mfma_cycles / r[0] : 0.0);
  hipFree(in); hipFree(inl); hipFree(out); hipFree(outl); hipFree(w); hipFree(wl); hipFree(bias); hipFree(probe);
}

int main(int argc, char** argv) {
  if (argc > 1 && argv[1][0] == 'c') {
    clock_probe<16, 4, true, 32>("w8 dec4 full", 80, 160, 512, 512);
    clock_probe<16, 4, true, 32 | 1 | 4 | 8 | 16>("w8 dec4 MFMA + loop only", 80, 160, 512, 512);
    clock_probe<16, 4, true, 32 | 1>("w8 dec4 no global->LDS traffic", 80, 160, 512, 512);
    clock_probe<16, 4, true, 32 | 4>("w8 dec4 no LDS fragment reads", 80, 160, 512, 512);
    clock_probe<16, 4, true, 32 | 64>("w8 dec4 no halo staging (weights stream)", 80, 160, 512, 512);
    clock_probe<16, 4, true, 32 | 128>("w8 dec4 no weight DMA (halo streams)", 80, 160, 512, 512);
    clock_probe<16, 4, true, 32 | 16>("w8 dec4 no epilogue", 80, 160, 512, 512);
    clock_probe<16, 4, true, 32>("w8 dec6 full", 160, 320, 256, 256);
    clock_probe<8, 2, false, 32>("w4 dec6 full", 160, 320, 256, 256);
    clock_probe<8, 2, false, 32>("w4 dec8 full", 320, 640, 128, 128);
    clock_probe<8, 2, false, 32 | 1 | 4 | 8 | 16>("w4 dec8 MFMA + loop only", 320, 640, 128, 128);
    return 0;
  }
  if (argc > 1 && argv[1][0] == 'f') {  // ConvTranspose -> 3x3 prologue fusion, estimated BEFORE building it: the consumer layers of
    // upsample_layer_4 (-> decode_layer_8, 4-wave shape) and upsample_layer_3 + skip (-> decode_layer_6, 8-wave shape) with their halo read from a
    // quarter-resolution tensor (bit 512) and one / two extra taps' worth of MFMAs per chunk (bits 1024 / 2048: +11 % / +22 %; the in-kernel
    // up-sampling GEMM is +17 %).  What the fusion could save at best = the up-sampling launch (39.4 / 36.6 us) minus the growth of this launch.
    clock_probe<8, 2, false, 32>("w4 dec8 as shipped", 320, 640, 128, 128);
    clock_probe<8, 2, false, 32 | 512>("w4 dec8 halo from quarter-resolution tensor", 320, 640, 128, 128);
    clock_probe<8, 2, false, 32 | 512 | 1024>("w4 dec8 quarter-res halo, +11 % MFMAs", 320, 640, 128, 128);
    clock_probe<8, 2, false, 32 | 512 | 1024 | 2048>("w4 dec8 quarter-res halo, +22 % MFMAs", 320, 640, 128, 128);
    clock_probe<16, 4, true, 32>("w8 dec6 as shipped", 160, 320, 256, 256);
    clock_probe<16, 4, true, 32 | 512>("w8 dec6 halo from quarter-resolution tensor", 160, 320, 256, 256);
    clock_probe<16, 4, true, 32 | 512 | 1024>("w8 dec6 quarter-res halo, +11 % MFMAs", 160, 320, 256, 256);
    clock_probe<16, 4, true, 32 | 512 | 1024 | 2048>("w8 dec6 quarter-res halo, +22 % MFMAs", 160, 320, 256, 256);
    return 0;
  }
  if (argc > 1 && argv[1][0] == 'b') {  // barrier A/B: VP_LDS_BARRIER (library) vs __syncthreads() (drains vmcnt(0) at every tap, ABL bit 256)
    clock_probe<16, 4, true, 32>("w8 dec4 lds barrier", 80, 160, 512, 512);
    clock_probe<16, 4, true, 32 | 256>("w8 dec4 __syncthreads", 80, 160, 512, 512);
    clock_probe<16, 4, true, 32>("w8 dec6 lds barrier", 160, 320, 256, 256);
    clock_probe<16, 4, true, 32 | 256>("w8 dec6 __syncthreads", 160, 320, 256, 256);
    clock_probe<8, 2, false, 32>("w4 dec8 lds barrier", 320, 640, 128, 128);
    clock_probe<8, 2, false, 32 | 256>("w4 dec8 __syncthreads", 320, 640, 128, 128);
    clock_probe<8, 2, false, 32>("w4 dec6 lds barrier", 160, 320, 256, 256);
    clock_probe<8, 2, false, 32 | 256>("w4 dec6 __syncthreads", 160, 320, 256, 256);
    clock_probe<16, 4, true, 32 | 64>("w8 dec4 lds barrier, no halo staging", 80, 160, 512, 512);
    clock_probe<16, 4, true, 32 | 128>("w8 dec4 lds barrier, no weight DMA", 80, 160, 512, 512);
    return 0;
  }
  run_shape<16, 4, true>("w8 dec4 512->512 80x160", 80, 160, 512, 512);
  run_shape<8, 2, false>("w4 dec4 512->512 80x160", 80, 160, 512, 512);
  run_shape<16, 4, true>("w8 dec6 256->256 160x320", 160, 320, 256, 256);
  run_shape<8, 2, false>("w4 dec6 256->256 160x320", 160, 320, 256, 256);
  run_shape<16, 4, true>("w8 dec8 128->128 320x640", 320, 640, 128, 128);
  run_shape<8, 2, false>("w4 dec8 128->128 320x640", 320, 640, 128, 128);
  return 0;
}
