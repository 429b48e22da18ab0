// Developer tool: ablation timing of the LDS-DMA GEMM (kernels_gemm_dma.hip) on the neck's up-sampling shapes.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -w tools/gemm_dma_ablate.hip -o tools/_gemm_dma_ablate
// ABL bits: 1 no DMA in the loop | 2 no MFMA | 4 no LDS fragment reads | 8 no barrier / vmcnt wait | 16 no epilogue.
#include <cstdio>
#include <vector>

#include "../autoware_vision_pilot_amd/csrc/kernels_gemm_dma.hip"
#include "tool_ones.hpp"

using namespace vp;
namespace vp { hipError_t launch_splitk_finish(const ConvGemmParams&, hipStream_t) { return hipErrorInvalidValue; } }

template <int ABL>
static float time_variant(const ConvGemmParams& p, int iters) {
  constexpr int lds = 3 * 2 * (256 + 128) * 64;
  auto k = gemm_dma_kernel<true, ABL>;
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  const int M = p.H * p.W;
  dim3 grid(((M + 127) / 128) * (p.Ncols / 256) * p.nsplit);
  hipEvent_t a, b;
  (void)hipEventCreate(&a); (void)hipEventCreate(&b);
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(k, grid, dim3(512), lds, 0, p);
  (void)hipEventRecord(a, 0);
  for (int i = 0; i < iters; ++i) hipLaunchKernelGGL(k, grid, dim3(512), lds, 0, p);
  (void)hipEventRecord(b, 0);
  (void)hipEventSynchronize(b);
  float ms = 0;
  (void)hipEventElapsedTime(&ms, a, b);
  return ms * 1000.0f / iters;
}

static void run_shape(const char* name, int H, int W, int Cin, int Cin2, int Cout, int nsplit) {
  const int M = H * W, N = 4 * Cout, Kw = Cin + Cin2;
  const size_t in_n = (size_t)M * Cin, in2_n = (size_t)4 * M * Cin2, w_n = (size_t)N * Kw, out_n = (size_t)4 * M * Cout;
  half_t *in, *inl, *out, *outl, *w, *wl;
  float *bias, *partial;
  (void)hipMalloc(&in, (in_n + in2_n) * 2); (void)hipMalloc(&inl, (in_n + in2_n) * 2); (void)hipMalloc(&out, out_n * 2); (void)hipMalloc(&outl, out_n * 2);
  (void)hipMalloc(&w, w_n * 2); (void)hipMalloc(&wl, w_n * 2); (void)hipMalloc(&bias, N * 4); (void)hipMalloc(&partial, (size_t)nsplit * M * N * 4 + 1024 * 1024);
  std::vector<half_t> h(std::max(in_n + in2_n, w_n));
  unsigned s = 4242;
  for (auto& v : h) { s = s * 1664525u + 1013904223u; v = (half_t)(((int)(s >> 9) % 2001 - 1000) * 0.001f); }
  (void)hipMemcpy(in, h.data(), (in_n + in2_n) * 2, hipMemcpyHostToDevice); (void)hipMemcpy(w, h.data(), w_n * 2, hipMemcpyHostToDevice);
  (void)hipMemcpy(inl, h.data(), (in_n + in2_n) * 2, hipMemcpyHostToDevice); (void)hipMemcpy(wl, h.data(), w_n * 2, hipMemcpyHostToDevice);
  (void)hipMemset(bias, 0, N * 4);
  ConvGemmParams p{};
  p.in_hi = in; p.in_lo = inl; p.H = H; p.W = W; p.Cin = Cin; p.Cin2 = Cin2; p.in2_delta_hi = (long long)in_n; p.in2_delta_lo = (long long)in_n;
  p.w_hi = w; p.w_lo = wl; p.bias = bias; p.wscale = tool_dev_ones(N); p.ks = 1; p.Ncols = N; p.CoutW = N; p.store_mode = STORE_SHUFFLE2; p.out_hi = out; p.out_lo = outl; p.Cstore = Cout;
  p.Creal = Cout; p.nsplit = nsplit; p.partial = partial;
  const double gflop = 2.0 * M * (double)N * Kw / 1e9;
  const int it = 30;
  const float t0 = time_variant<0>(p, it), t16 = time_variant<16>(p, it), t1 = time_variant<1>(p, it), t4 = time_variant<4>(p, it), t8 = time_variant<8 | 1>(p, it),
              t2 = time_variant<2>(p, it), t5 = time_variant<1 | 4>(p, it), t29 = time_variant<1 | 4 | 8 | 16>(p, it), t6 = time_variant<2 | 4>(p, it);
  const int steps = Kw / 32 / nsplit;
  std::printf("%-34s %5.1f GF, %3d WGs x %2d steps | full %6.1f us (%5.1f TF alg) | noEpi %6.1f | noDMA %6.1f | noLdsRead %6.1f | noDMA+noBarrier %6.1f | noMFMA %6.1f | "
              "noDMA+noLdsRead %6.1f | noMFMA+noLdsRead (DMA + barriers only) %6.1f | MFMA + loop only %6.1f\n",
              name, gflop, ((M + 127) / 128) * (N / 256) * nsplit, steps, t0, gflop / t0 * 1e3, t16, t1, t4, t8, t2, t5, t6, t29);
  (void)hipFree(in); (void)hipFree(inl); (void)hipFree(out); (void)hipFree(outl); (void)hipFree(w); (void)hipFree(wl); (void)hipFree(bias); (void)hipFree(partial);
}

int main() {
  run_shape("up2 512+32 -> 512, 40x80", 40, 80, 512, 32, 512, 1);
  run_shape("up1 768+64 -> 768, 20x40", 20, 40, 768, 64, 768, 1);
  run_shape("up1 768+64 -> 768, 20x40, 2 slices", 20, 40, 768, 64, 768, 2);
  run_shape("up0 1280+96 -> 1280, 10x20", 10, 20, 1280, 96, 1280, 1);
  run_shape("up0 1280+96 -> 1280, 10x20, 4 slices", 10, 20, 1280, 96, 1280, 4);
  return 0;
}
