// Developer tool: ablation timing of the small-map parity kernel (kernels_conv3x3_map.hip) on the neck's layer shapes.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/map_ablate.hip -o tools/_map_ablate
// ABL bits: 1 no DMA inside the loop | 2 no MFMA | 4 tap-invariant fragment addresses | 8 no barrier / vmcnt wait in the loop
#include <cstdio>
#include <vector>

#include "../autoware_vision_pilot_amd/csrc/kernels_conv3x3_map.hip"

using namespace vp;
namespace vp { hipError_t launch_splitk_finish(const ConvGemmParams&, hipStream_t) { return hipSuccess; } }  // the tool times the main kernel only

template <int ABL>
static float time_variant(const ConvGemmParams& p, int iters) {
  auto k = conv3x3_map_kernel<ABL>;
  hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, mapk::Neck::LDS);
  dim3 grid((p.H / 20) * (p.W / 40) * (p.CoutW / 32) * p.nsplit);
  hipEvent_t a, b;
  hipEventCreate(&a);
  hipEventCreate(&b);
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(k, grid, dim3(512), mapk::Neck::LDS, 0, p);
  hipEventRecord(a, 0);
  for (int i = 0; i < iters; ++i) hipLaunchKernelGGL(k, grid, dim3(512), mapk::Neck::LDS, 0, p);
  hipEventRecord(b, 0);
  hipEventSynchronize(b);
  float ms = 0;
  hipEventElapsedTime(&ms, a, b);
  return ms * 1000.0f / iters;
}

static void run_shape(const char* name, int H, int W, int Cin, int Cout, int nsplit) {
  const size_t in_n = (size_t)H * W * Cin, w_n = (size_t)9 * Cout * Cin;
  half_t *in, *inl, *w, *wl, *zeros;
  float* partial;
  hipMalloc(&in, in_n * 2); hipMalloc(&inl, in_n * 2); hipMalloc(&w, w_n * 2); hipMalloc(&wl, w_n * 2); hipMalloc(&zeros, 256);
  hipMalloc(&partial, (size_t)nsplit * H * W * Cout * 4);
  hipMemset(zeros, 0, 256);
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
  ConvGemmParams p{};
  p.in_hi = in; p.in_lo = inl; p.H = H; p.W = W; p.Cin = Cin; p.w_hi = w; p.w_lo = wl; p.ks = 3; p.Ncols = Cout; p.CoutW = Cout;
  p.nsplit = nsplit; p.partial = partial; p.zeros = zeros;
  const double gflop = 2.0 * H * W * (double)Cout * Cin * 9 / 1e9;
  const int it = 20;
  const float t0 = time_variant<0>(p, it), t1 = time_variant<1>(p, it), t2 = time_variant<2>(p, it), t4 = time_variant<4>(p, it), t8 = time_variant<8>(p, it),
              t9 = time_variant<1 | 8>(p, it), t13 = time_variant<1 | 4 | 8>(p, it), t14 = time_variant<2 | 4 | 8>(p, it), t15 = time_variant<1 | 2 | 4 | 8>(p, it);
  std::printf("%-34s %5.1f GF nsplit %2d grid %3d | full %6.1f us (%5.1f TF alg) | noDMA %6.1f | noMFMA %6.1f | flatAddr %6.1f | noBarrier %6.1f | noDMA+noBarrier %6.1f | MFMA+LDS reads only %6.1f | DMA+LDS reads only(noMFMA,flat,noBar) %6.1f | loop+reads only %6.1f\n",
              name, gflop, nsplit, (H / 20) * (W / 40) * (Cout / 32) * nsplit, t0, gflop / t0 * 1e-3 * 1e3, t1, t2, t4, t8, t9, t13, t14, t15);
  hipFree(in); hipFree(inl); hipFree(w); hipFree(wl); hipFree(partial); hipFree(zeros);
}

int main() {
  run_shape("dec0 1280->768 20x40", 20, 40, 1280, 768, 10);
  run_shape("dec1 768->768 20x40", 20, 40, 768, 768, 10);
  run_shape("dec2 768->512 40x80", 40, 80, 768, 512, 4);
  run_shape("dec3 512->512 40x80", 40, 80, 512, 512, 4);
  run_shape("dec3 512->512 40x80 nsplit 2", 40, 80, 512, 512, 2);
  return 0;
}
