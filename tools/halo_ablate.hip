// Developer tool: ablation timing of the 3x3 halo kernel on decoder-layer shapes (not part of the library).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I autoware_vision_pilot_amd/csrc tools/halo_ablate.hip -o /tmp/halo_ablate
// Variants (template ABL bits): 0 full | 1 no global->LDS traffic | 2 no MFMA | 4 no LDS fragment reads | 8 no barrier.
#include <cstdio>
#include <vector>

#include "../autoware_vision_pilot_amd/csrc/kernels_conv3x3.hip"
#include "tool_ones.hpp"
#include "../autoware_vision_pilot_amd/csrc/kernels_conv.hip"

using namespace vp;

#define CK(x)                                                                     \
  do {                                                                            \
    hipError_t e_ = (x);                                                          \
    if (e_ != hipSuccess) {                                                       \
      std::printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); \
      return 1;                                                                   \
    }                                                                             \
  } while (0)

template <int CO, int TH, int TW, int ABL, bool FAST = false>
static float time_variant(const ConvGemmParams& p, int iters) {
  constexpr int lds_main = 2 * ((TH + 2) * (TW + 2) * 80 + CO * 64);
  constexpr int lds_a = lds_main > epilogue_stage_bytes<TH * TW, 2>() ? lds_main : epilogue_stage_bytes<TH * TW, 2>();
  constexpr int lds = lds_a > epilogue_fp16_stage_bytes<TH * TW, CO>() ? lds_a : epilogue_fp16_stage_bytes<TH * TW, CO>();
  auto k = conv3x3_halo_kernel<CO, TH, TW, 2, 2, false, ABL, FAST>;
  hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  dim3 grid(((p.H + TH - 1) / TH) * ((p.W + TW - 1) / TW) * (p.CoutW / CO) * p.nsplit);
  hipEvent_t a, b;
  hipEventCreate(&a);
  hipEventCreate(&b);
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(k, grid, dim3(256), lds, 0, p);
  hipEventRecord(a, 0);
  for (int i = 0; i < iters; ++i) hipLaunchKernelGGL(k, grid, dim3(256), lds, 0, p);
  hipEventRecord(b, 0);
  hipEventSynchronize(b);
  float ms = 0;
  hipEventElapsedTime(&ms, a, b);
  return ms * 1000.0f / iters;
}

template <int CO, int TH, int TW>
static int run_shape(const char* name, int H, int W, int Cin, int Cout) {
  const size_t in_n = (size_t)H * W * Cin, out_n = (size_t)H * W * Cout, w_n = (size_t)9 * Cout * Cin;
  half_t *in, *out, *w;
  float* bias;
  CK(hipMalloc(&in, in_n * 2));
  CK(hipMalloc(&out, out_n * 2));
  CK(hipMalloc(&w, w_n * 2));
  CK(hipMalloc(&bias, Cout * 4));
  std::vector<half_t> h(in_n > w_n ? in_n : w_n);
  unsigned s = 12345;
  for (auto& v : h) {
    s = s * 1664525u + 1013904223u;
    v = (half_t)(((int)(s >> 9) % 2001 - 1000) * 0.001f);
  }
  CK(hipMemcpy(in, h.data(), in_n * 2, hipMemcpyHostToDevice));
  CK(hipMemcpy(w, h.data(), w_n * 2, hipMemcpyHostToDevice));
  CK(hipMemset(bias, 0, Cout * 4));
  ConvGemmParams p{};
  p.in_hi = in;
  p.H = H;
  p.W = W;
  p.Cin = Cin;
  p.w_hi = w;
  p.bias = bias;
  p.wscale = tool_dev_ones(Cout);
  p.ks = 3;
  p.Ncols = Cout;
  p.CoutW = Cout;
  p.act = ACT_GELU_F16;
  p.out_hi = out;
  p.Cstore = Cout;
  p.Creal = Cout;
  p.nsplit = 1;
  const double gflop = 2.0 * H * W * (double)Cout * Cin * 9 / 1e9;
  const int it = 20;
  const float t0 = time_variant<CO, TH, TW, 0>(p, it), t1 = time_variant<CO, TH, TW, 1>(p, it), t2 = time_variant<CO, TH, TW, 2>(p, it),
              t4 = time_variant<CO, TH, TW, 4>(p, it), t8 = time_variant<CO, TH, TW, 8>(p, it), t5 = time_variant<CO, TH, TW, 5>(p, it),
              t13 = time_variant<CO, TH, TW, 13>(p, it), t7 = time_variant<CO, TH, TW, 7>(p, it), tf = time_variant<CO, TH, TW, 0, true>(p, it),
              tf7 = time_variant<CO, TH, TW, 7, true>(p, it);
  for (int g = 1; g <= 4; ++g) {
    ConvGemmParams q = p;
    q.ks = 3 + g;
    std::printf("  stagger sleep x%d: by TG_ID %7.1f us | by block>>8 %7.1f us\n", g, time_variant<CO, TH, TW, 16, true>(q, it), time_variant<CO, TH, TW, 32, true>(q, it));
  }
  std::printf("%-28s %6.1f GF | full %7.1f us (%6.1f TF) | noGlobal %7.1f | noMFMA %7.1f | noLdsRead %7.1f | noBarrier %7.1f | "
              "noGlobal+noLdsRead %7.1f | mfma+barrier-free-only %7.1f | nothing-but-loop %7.1f | FASTEPI full %7.1f (%6.1f TF) nothing %7.1f\n",
              name, gflop, t0, gflop / t0 * 1e3, t1, t2, t4, t8, t5, t13, t7, tf, gflop / tf * 1e3, tf7);
  hipFree(in);
  hipFree(out);
  hipFree(w);
  hipFree(bias);
  return 0;
}

// split-K neck shapes: kernel only (partials written), finish kernel timed separately
template <int CO, int TH, int TW>
static int run_split(const char* name, int H, int W, int Cin, int Cout, int nsplit) {
  const size_t in_n = (size_t)H * W * Cin, out_n = (size_t)H * W * Cout, w_n = (size_t)9 * Cout * Cin;
  half_t *in, *out, *w;
  float *bias, *partial;
  CK(hipMalloc(&in, in_n * 2));
  CK(hipMalloc(&out, out_n * 2));
  CK(hipMalloc(&w, w_n * 2));
  CK(hipMalloc(&bias, Cout * 4));
  CK(hipMalloc(&partial, (size_t)nsplit * H * W * Cout * 4));
  std::vector<half_t> h(in_n > w_n ? in_n : w_n);
  unsigned s = 12345;
  for (auto& v : h) {
    s = s * 1664525u + 1013904223u;
    v = (half_t)(((int)(s >> 9) % 2001 - 1000) * 0.001f);
  }
  CK(hipMemcpy(in, h.data(), in_n * 2, hipMemcpyHostToDevice));
  CK(hipMemcpy(w, h.data(), w_n * 2, hipMemcpyHostToDevice));
  CK(hipMemset(bias, 0, Cout * 4));
  ConvGemmParams p{};
  p.in_hi = in; p.H = H; p.W = W; p.Cin = Cin; p.w_hi = w; p.bias = bias; p.wscale = tool_dev_ones(Cout); p.ks = 3; p.Ncols = Cout; p.CoutW = Cout;
  p.act = ACT_GELU_F16; p.out_hi = out; p.Cstore = Cout; p.Creal = Cout; p.nsplit = nsplit; p.partial = partial;
  const double gflop = 2.0 * H * W * (double)Cout * Cin * 9 / 1e9;
  const int it = 20;
  const float t0 = time_variant<CO, TH, TW, 0>(p, it), t1 = time_variant<CO, TH, TW, 1>(p, it), t2 = time_variant<CO, TH, TW, 2>(p, it),
              t4 = time_variant<CO, TH, TW, 4>(p, it), t8 = time_variant<CO, TH, TW, 8>(p, it), t7 = time_variant<CO, TH, TW, 7>(p, it),
              t15 = time_variant<CO, TH, TW, 15>(p, it);
  hipEvent_t a, b;
  hipEventCreate(&a); hipEventCreate(&b);
  launch_splitk_finish(p, 0);
  hipEventRecord(a, 0);
  for (int i = 0; i < it; ++i) launch_splitk_finish(p, 0);
  hipEventRecord(b, 0);
  hipEventSynchronize(b);
  float fms = 0;
  hipEventElapsedTime(&fms, a, b);
  const int blocks = ((H + TH - 1) / TH) * ((W + TW - 1) / TW) * (Cout / CO) * nsplit;
  std::printf("%-34s ns=%2d blocks=%4d %5.1f GF | full %6.1f us (%5.1f TF) | noGlobal %6.1f | noMFMA %6.1f | noLdsRead %6.1f | noBarrier %6.1f | "
              "nothing-but-loop %6.1f | nothing, no barrier %6.1f | finish %5.1f us\n",
              name, nsplit, blocks, gflop, t0, gflop / t0 * 1e3, t1, t2, t4, t8, t7, t15, fms * 1000.0f / it);
  hipFree(in); hipFree(out); hipFree(w); hipFree(bias); hipFree(partial);
  return 0;
}

// 64 -> 3 head conv (co32 tile, 1x4 waves): where does its time go?
template <int ABL>
static float time_head(const ConvGemmParams& p, int iters) {
  constexpr int CO = 32, TH = 8, TW = 16;
  constexpr int lds_main = 2 * ((TH + 2) * (TW + 2) * 80 + CO * 64);
  constexpr int lds = lds_main > epilogue_stage_bytes<TH * TW, 1>() ? lds_main : epilogue_stage_bytes<TH * TW, 1>();
  auto k = conv3x3_halo_kernel<CO, TH, TW, 1, 4, false, ABL, false>;
  hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  dim3 grid(((p.H + TH - 1) / TH) * ((p.W + TW - 1) / TW) * (p.CoutW / CO));
  hipEvent_t a, b;
  hipEventCreate(&a);
  hipEventCreate(&b);
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(k, grid, dim3(256), lds, 0, p);
  hipEventRecord(a, 0);
  for (int i = 0; i < iters; ++i) hipLaunchKernelGGL(k, grid, dim3(256), lds, 0, p);
  hipEventRecord(b, 0);
  hipEventSynchronize(b);
  float ms = 0;
  hipEventElapsedTime(&ms, a, b);
  return ms * 1000.0f / iters;
}
static int run_head() {
  const int H = 320, W = 640, Cin = 64, Cout = 32;
  half_t *in, *w;
  float *bias, *outf;
  CK(hipMalloc(&in, (size_t)H * W * Cin * 2));
  CK(hipMalloc(&w, (size_t)9 * Cout * Cin * 2));
  CK(hipMalloc(&bias, Cout * 4));
  CK(hipMalloc(&outf, (size_t)3 * H * W * 4));
  CK(hipMemset(in, 0, (size_t)H * W * Cin * 2));
  CK(hipMemset(w, 0, (size_t)9 * Cout * Cin * 2));
  CK(hipMemset(bias, 0, Cout * 4));
  ConvGemmParams p{};
  p.in_hi = in; p.H = H; p.W = W; p.Cin = Cin; p.w_hi = w; p.bias = bias; p.wscale = tool_dev_ones(Cout); p.ks = 3; p.Ncols = Cout; p.CoutW = Cout;
  p.act = ACT_NONE; p.store_mode = STORE_NCHW_F32; p.out_f32 = outf; p.Creal = 3; p.nsplit = 1;
  const int it = 20;
  std::printf("head 64->3 320x640: full %6.1f us | noGlobal %6.1f | noMFMA %6.1f | noLdsRead %6.1f | noBarrier %6.1f | nothing-but-loop %6.1f | nothing, no barrier %6.1f\n",
              time_head<0>(p, it), time_head<1>(p, it), time_head<2>(p, it), time_head<4>(p, it), time_head<8>(p, it), time_head<7>(p, it), time_head<15>(p, it));
  return 0;
}

int main(int argc, char** argv) {
  if (argc > 1 && argv[1][0] == 'h') return run_head();
  if (argc > 1) {
    for (int ns : {1, 2, 4, 8, 15}) run_split<128, 8, 16>("dec0 1920->512 20x40 t8x16", 20, 40, 1920, 512, ns);
    for (int ns : {1, 2, 4, 8, 15}) run_split<64, 8, 16>("dec0 1920->512 20x40 co64 t8x16", 20, 40, 1920, 512, ns);
    for (int ns : {1, 2, 4, 8}) run_split<128, 8, 16>("dec1 512->512 20x40 t8x16", 20, 40, 512, 512, ns);
    for (int ns : {1, 2, 3, 6}) run_split<128, 8, 16>("dec2 768->512 40x80 t8x16", 40, 80, 768, 512, ns);
    for (int ns : {1, 2, 3, 6}) run_split<128, 8, 16>("dec3 512->512 40x80 t8x16", 40, 80, 512, 512, ns);
    for (int ns : {1, 2, 3}) run_split<128, 8, 16>("dec5 512->256 80x160 t8x16", 80, 160, 512, 256, ns);
    for (int ns : {1, 2, 3}) run_split<64, 8, 16>("dec5 512->256 80x160 co64 t8x16", 80, 160, 512, 256, ns);
    for (int ns : {1, 2, 3}) run_split<64, 8, 16>("dec2 768->512 40x80 co64 t8x16", 40, 80, 768, 512, ns);
    for (int ns : {1, 2, 3}) run_split<64, 8, 16>("dec3 512->512 40x80 co64 t8x16", 40, 80, 512, 512, ns);
    for (int ns : {2, 4, 6, 10}) run_split<64, 8, 16>("dec0 1280->768 20x40 co64 t8x16", 20, 40, 1280, 768, ns);
    for (int ns : {2, 4, 6, 10}) run_split<128, 8, 16>("dec0 1280->768 20x40 co128 t8x16", 20, 40, 1280, 768, ns);
    for (int ns : {2, 4, 6}) run_split<64, 8, 16>("dec1 768->768 20x40 co64 t8x16", 20, 40, 768, 768, ns);
    for (int ns : {2, 4, 6}) run_split<128, 8, 16>("dec1 768->768 20x40 co128 t8x16", 20, 40, 768, 768, ns);
    return 0;
  }
  run_shape<128, 16, 16>("dec8 128->128 320x640 t16x16", 320, 640, 128, 128);
  run_shape<128, 8, 16>("dec8 128->128 320x640 t8x16", 320, 640, 128, 128);
  run_shape<128, 16, 16>("dec6 256->256 160x320 t16x16", 160, 320, 256, 256);
  run_shape<128, 8, 16>("dec4 512->512 80x160 t8x16", 80, 160, 512, 512);
  return 0;
}
