// Developer tool: where the time of the fused MBConv front half (kernels_mbconv.hip, mbconv_front_kernel) goes on the encoder's shapes:
// time per launch (with and without the per-channel pool sums), and with the expand loop / the depthwise taps / the pool atomics / the SiLUs removed (results are then wrong: timing only).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/mbf_check.hip -o tools/_mbf_check
#include <cstdio>
#include <vector>

#include "../autoware_vision_pilot_amd/csrc/kernels_mbconv.hip"
#include "tool_ones.hpp"

using namespace vp;

template <class T>
static T* dev(size_t n, float fill_scale = 0.0f) {
  std::vector<T> h(n);
  unsigned s = 777u;
  for (auto& v : h) {
    s = s * 1664525u + 1013904223u;
    v = (T)(((int)(s >> 9) % 2001 - 1000) * 0.001f * fill_scale);
  }
  T* d;
  hipMalloc(&d, n * sizeof(T));
  hipMemcpy(d, h.data(), n * sizeof(T), hipMemcpyHostToDevice);
  return d;
}

template <int K, int S>
static void run(int cin, int cexp, int H, int W) {
  const int Cin = (cin + 31) / 32 * 32, Cexp = (cexp + 31) / 32 * 32, OH = H / S, OW = W / S, replicas = 8, sq = cin / 4 > 0 ? cin / 4 : 1;
  MbFrontParams p{};
  p.in = ActView{dev<half_t>((size_t)H * W * Cin, 1.0f), dev<half_t>((size_t)H * W * Cin, 0.0005f), H, W, Cin};
  p.w_hi = dev<half_t>((size_t)Cexp * Cin, 0.2f);
  p.w_lo = dev<half_t>((size_t)Cexp * Cin, 0.0001f);
  p.b_exp = dev<float>(Cexp, 0.1f);
  p.s_exp = tool_dev_ones(Cexp);
  p.w_dw = dev<float>((size_t)K * K * Cexp, 0.3f);
  p.b_dw = dev<float>(Cexp, 0.1f);
  p.out = ActView{dev<half_t>((size_t)OH * OW * Cexp), dev<half_t>((size_t)OH * OW * Cexp), OH, OW, Cexp};
  p.k = K;
  p.stride = S;
  p.sums = dev<unsigned long long>((size_t)replicas * Cexp);
  p.replicas = replicas;
  p.w1 = dev<float>((size_t)sq * Cexp, 0.2f);
  p.sq = sq;
  p.zsums = dev<unsigned long long>((size_t)replicas * 64);
  auto time_it = [&](auto go) {
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    for (int i = 0; i < 5; ++i) go();
    hipEventRecord(a, 0);
    for (int i = 0; i < 50; ++i) go();
    hipEventRecord(b, 0);
    hipEventSynchronize(b);
    float ms = 0;
    hipEventElapsedTime(&ms, a, b);
    return ms * 1000.0f / 50;
  };
#define T_(ABL) time_it([&] { return launch_mb<K, S, ABL>(p, 0); })
  using T = MbTile<K, S>;
  const int grid = ((OH + T::TH - 1) / T::TH) * ((OW + T::TW - 1) / T::TW) * (Cexp / 32);
  MbFrontParams pz = p;
  pz.sums = nullptr;   // the engine's form: the back half starts from the squeeze sums, no per-channel sums
  const float t_nosums = time_it([&] { return launch_mb<K, S, 0>(pz, 0); });
  std::printf("k%d s%d  %4d -> %4d  %3dx%-3d  %4d workgroups, %d K chunks | engine form (squeeze sums only) %5.1f us | with per-channel sums %5.1f | no expand loop %5.1f | no taps %5.1f | no pool atomics %5.1f | no SiLU %5.1f | "
              "no loop, no taps %5.1f | nothing but loads of the tile and the store %5.1f | loop without its loads %5.1f | loop with loads only %5.1f\n", K, S, cin, cexp, H, W, grid, Cin / 32, t_nosums, T_(0), T_(1), T_(2), T_(4), T_(8), T_(3), T_(15), T_(16), T_(32));
}

int main() {
  run<3, 2>(16, 96, 160, 320);
  run<3, 1>(24, 144, 80, 160);
  run<5, 2>(24, 144, 80, 160);
  run<5, 1>(40, 240, 40, 80);
  run<3, 2>(40, 240, 40, 80);
  run<3, 1>(80, 480, 20, 40);
  run<5, 1>(80, 480, 20, 40);
  run<5, 1>(112, 672, 20, 40);
  run<5, 2>(112, 672, 20, 40);
  run<5, 1>(192, 1152, 10, 20);
  run<3, 1>(192, 1152, 10, 20);
  return 0;
}
