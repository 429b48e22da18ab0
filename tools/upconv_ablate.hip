// Developer tool (round 6): ablation timing of the composed up-sampling kernel (kernels_upconv.hip) on the three big stage shapes of a scene network.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/upconv_ablate.hip -o tools/_upconv_ablate && tools/_upconv_ablate
// ABL bits: 1 no halo loads / stores after the prologue | 2 no MFMA | 16 no epilogue | 32 epilogue without its global stores | 128 no weight DMA after
// the prologue.  Values are meaningless (random planes, unpacked weights): timing only.
#include <cstdio>
#include <vector>

#include "../autoware_vision_pilot_amd/csrc/kernels_upconv.hip"

using namespace vp;

template <int TH, int WPX, bool HDB, int ABL>
static float time_variant(const UpconvParams& p, int iters) {
  constexpr int lds = (HDB ? 2 : 1) * 2 * ((TH + 2) * 18 * 80) + 6 * (128 * 64);
  auto k = upconv_x3_kernel<128, TH, 2, WPX, HDB, ACT_GELU, false, false, ABL>;
  hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  dim3 grid(((p.H + TH - 1) / TH) * ((p.W + 15) / 16) * 4 * (p.CoutW / 128));
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
static void run_shape(const char* name, int H, int W, int Cin, int Cs, int Cout) {
  const size_t in_n = (size_t)H * W * Cin, sk_n = (size_t)4 * H * W * (Cs ? Cs : 1), out_n = (size_t)4 * H * W * Cout;
  const int S = upconv_steps(Cin, Cs);
  const size_t w_n = (size_t)4 * S * Cout * 32;
  half_t *in, *inl, *sk, *skl, *out, *outl, *w, *wl;
  float *bias, *wsc, *partial;
  hipMalloc(&in, in_n * 2); hipMalloc(&inl, in_n * 2); hipMalloc(&sk, sk_n * 2); hipMalloc(&skl, sk_n * 2); hipMalloc(&out, out_n * 2); hipMalloc(&outl, out_n * 2);
  hipMalloc(&w, w_n * 2); hipMalloc(&wl, w_n * 2); hipMalloc(&bias, 9 * Cout * 4); hipMalloc(&wsc, 4 * Cout * 4); hipMalloc(&partial, 64 * 1024 * 4);
  std::vector<half_t> h(std::max(std::max(in_n, sk_n), w_n));
  unsigned s = 12345;
  for (auto& v : h) {
    s = s * 1664525u + 1013904223u;
    v = (half_t)(((int)(s >> 9) % 2001 - 1000) * 0.001f);
  }
  hipMemcpy(in, h.data(), in_n * 2, hipMemcpyHostToDevice);
  hipMemcpy(sk, h.data(), sk_n * 2, hipMemcpyHostToDevice);
  hipMemcpy(w, h.data(), w_n * 2, hipMemcpyHostToDevice);
  for (auto& v : h) v = (half_t)((float)v * 0.0004f);
  hipMemcpy(inl, h.data(), in_n * 2, hipMemcpyHostToDevice);
  hipMemcpy(skl, h.data(), sk_n * 2, hipMemcpyHostToDevice);
  hipMemcpy(wl, h.data(), w_n * 2, hipMemcpyHostToDevice);
  hipMemset(bias, 0, 9 * Cout * 4);
  std::vector<float> ones(4 * Cout, 1.0f);
  hipMemcpy(wsc, ones.data(), ones.size() * 4, hipMemcpyHostToDevice);
  UpconvParams p{};
  p.in_hi = in; p.in_lo = inl; p.sk_hi = Cs ? sk : nullptr; p.sk_lo = Cs ? skl : nullptr; p.H = H; p.W = W; p.Cin = Cin; p.Cs = Cs; p.w_hi = w; p.w_lo = wl;
  p.bias = bias; p.wscale = wsc; p.CoutW = Cout; p.Ncols = Cout; p.Cstore = Cout; p.out_hi = out; p.out_lo = outl; p.act = ACT_GELU; p.nsplit = 1; p.partial = partial;
  const double gflop = 2.0 * 4.0 * H * W * Cout * (4.0 * Cin + 9.0 * Cs) / 1e9;
  const int it = 20;
  const float t0 = time_variant<TH, WPX, HDB, 0>(p, it), t16 = time_variant<TH, WPX, HDB, 16>(p, it), t32 = time_variant<TH, WPX, HDB, 32>(p, it),
              t1 = time_variant<TH, WPX, HDB, 1>(p, it), t128 = time_variant<TH, WPX, HDB, 128>(p, it), t129 = time_variant<TH, WPX, HDB, 129>(p, it),
              t2 = time_variant<TH, WPX, HDB, 2>(p, it), t18 = time_variant<TH, WPX, HDB, 18>(p, it), t145 = time_variant<TH, WPX, HDB, 145>(p, it),
              t147 = time_variant<TH, WPX, HDB, 147>(p, it);
  std::printf("%s\t%s\t%.1f GFLOP executed, %d steps\tfull %.1f (%.0f TFLOP/s executed)\tno-epilogue %.1f\tepilogue-no-stores %.1f\tno-halo %.1f\tno-weights %.1f\t"
              "neither %.1f\tno-MFMA %.1f\tno-MFMA-no-epilogue %.1f\tMFMA+LDS only (no streams, no epilogue) %.1f\tnothing but LDS reads + barriers %.1f\n",
              name, HDB ? "w8" : "w4", gflop, S, t0, gflop / t0 * 1e3, t16, t32, t1, t128, t129, t2, t18, t145, t147);
  for (void* q : {(void*)in, (void*)inl, (void*)sk, (void*)skl, (void*)out, (void*)outl, (void*)w, (void*)wl, (void*)bias, (void*)wsc, (void*)partial}) hipFree(q);
}

int main() {
  std::printf("# composed up-sampling kernel, parity mode, us per launch (20 launches back to back), random planes\n");
  run_shape<8, 2, false>("dec4 40x80 512+32->512", 40, 80, 512, 32, 512);
  run_shape<16, 4, true>("dec4 40x80 512+32->512", 40, 80, 512, 32, 512);
  run_shape<8, 2, false>("dec6 80x160 256+32->256", 80, 160, 256, 32, 256);
  run_shape<16, 4, true>("dec6 80x160 256+32->256", 80, 160, 256, 32, 256);
  run_shape<8, 2, false>("dec8 160x320 128->128", 160, 320, 128, 0, 128);
  run_shape<16, 4, true>("dec8 160x320 128->128", 160, 320, 128, 0, 128);
  return 0;
}
