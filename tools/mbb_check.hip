// Developer tool: the fused MBConv back half (kernels_mbconv.hip, mbconv_back_kernel) on the encoder's shapes, on the GPU, against a
// float64 host evaluation of the same tensors; and its time per launch.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/mbb_check.hip -o tools/_mbb_check
#include <cmath>
#include <cstdio>
#include <vector>

#include "../autoware_vision_pilot_amd/csrc/kernels_mbconv.hip"
#include "tool_ones.hpp"

using namespace vp;

static unsigned g_seed = 1u;
static float rnd() {  // ~N(0,1)
  float s = 0;
  for (int i = 0; i < 4; ++i) {
    g_seed = g_seed * 1664525u + 1013904223u;
    s += (float)(g_seed >> 8) * (1.0f / 16777216.0f) - 0.5f;
  }
  return s * 1.7320508f;
}
template <class T>
static T* up(const std::vector<T>& v) {
  T* d;
  hipMalloc(&d, v.size() * sizeof(T));
  hipMemcpy(d, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice);
  return d;
}

static void run(int cexp, int cout, int sq, int H, int W, bool residual, float wscale) {
  const int C = (cexp + 31) / 32 * 32, Cout = (cout + 31) / 32 * 32, M = H * W, replicas = 8, sqp = (sq + 3) / 4 * 4;
  std::vector<half_t> xh((size_t)M * C, (half_t)0.f), xl(xh.size(), (half_t)0.f), rh((size_t)M * Cout, (half_t)0.f), rl(rh.size(), (half_t)0.f);
  std::vector<double> xv((size_t)M * C, 0.0), rv((size_t)M * Cout, 0.0);
  std::vector<unsigned long long> sums((size_t)replicas * C, 0ull);
  for (int m = 0; m < M; ++m)
    for (int c = 0; c < cexp; ++c) {
      const float v = rnd() * 3.0f;
      const half_t h = (half_t)v, l = (half_t)(v - (float)h);
      xh[(size_t)m * C + c] = h;
      xl[(size_t)m * C + c] = l;
      const double d = (double)(float)h + (double)(float)l;
      xv[(size_t)m * C + c] = d;
      sums[(size_t)(m % replicas) * C + c] += (unsigned long long)(long long)std::llrint(d * 16777216.0);
    }
  for (int m = 0; m < M; ++m)
    for (int c = 0; c < cout; ++c) {
      const float v = rnd() * 2.0f;
      const half_t h = (half_t)v, l = (half_t)(v - (float)h);
      rh[(size_t)m * Cout + c] = h;
      rl[(size_t)m * Cout + c] = l;
      rv[(size_t)m * Cout + c] = (double)(float)h + (double)(float)l;
    }
  std::vector<float> w1((size_t)sq * C, 0.f), b1(sq), w2((size_t)C * sq, 0.f), b2(C, 0.f), w2q((size_t)sqp * C, 0.f), w((size_t)Cout * C, 0.f), bias(Cout, 0.f);
  for (int q = 0; q < sq; ++q) {
    for (int c = 0; c < cexp; ++c) w1[(size_t)q * C + c] = rnd() * 0.2f;
    b1[q] = rnd() * 0.1f;
  }
  for (int c = 0; c < cexp; ++c) {
    for (int q = 0; q < sq; ++q) {
      w2[(size_t)c * sq + q] = rnd() * 0.5f;
      w2q[((size_t)(q >> 2) * C + c) * 4 + (q & 3)] = w2[(size_t)c * sq + q];
    }
    b2[c] = rnd() * 0.2f;
  }
  for (int n = 0; n < cout; ++n) {
    for (int c = 0; c < cexp; ++c) w[(size_t)n * C + c] = rnd() * wscale / std::sqrt((float)cexp);
    bias[n] = rnd() * 0.2f;
  }
  MbBackParams p{};
  half_t *oh, *ol;
  hipMalloc(&oh, (size_t)M * Cout * 2);
  hipMalloc(&ol, (size_t)M * Cout * 2);
  p.in = ActView{up(xh), up(xl), H, W, C};
  p.se.sums = up(sums); p.se.replicas = replicas; p.se.C = C; p.se.Creal = cexp; p.se.sq = sq; p.se.inv_hw = 1.0f / M; p.se.w1 = up(w1); p.se.b1 = up(b1); p.se.frames = 1;
  p.w2q = up(w2q); p.b2 = up(b2); p.sqp = sqp; p.w = up(w); p.bias = up(bias); p.wscale = tool_dev_ones(bias.size());
  if (residual) p.res = ActView{up(rh), up(rl), H, W, Cout};
  p.out = ActView{oh, ol, H, W, Cout};
  if (launch_mbconv_back(p, 0) != hipSuccess) { std::printf("launch failed\n"); return; }
  hipDeviceSynchronize();
  std::vector<half_t> goh((size_t)M * Cout), gol(goh.size());
  hipMemcpy(goh.data(), oh, goh.size() * 2, hipMemcpyDeviceToHost);
  hipMemcpy(gol.data(), ol, gol.size() * 2, hipMemcpyDeviceToHost);
  // float64 reference
  std::vector<double> mean(C, 0.0), s1(sq), gate(C, 0.0);
  for (int c = 0; c < cexp; ++c) {
    long long t = 0;
    for (int r = 0; r < replicas; ++r) t += (long long)sums[(size_t)r * C + c];
    mean[c] = (double)t / 16777216.0 / M;
  }
  for (int q = 0; q < sq; ++q) {
    double z = b1[q];
    for (int c = 0; c < cexp; ++c) z += (double)w1[(size_t)q * C + c] * mean[c];
    s1[q] = z / (1.0 + std::exp(-z));
  }
  for (int c = 0; c < cexp; ++c) {
    double z = b2[c];
    for (int q = 0; q < sq; ++q) z += (double)w2[(size_t)c * sq + q] * s1[q];
    gate[c] = 1.0 / (1.0 + std::exp(-z));
  }
  double emax = 0, rmax = 0, pad = 0;
  for (int m = 0; m < M; ++m)
    for (int n = 0; n < Cout; ++n) {
      const double got = (double)(float)goh[(size_t)m * Cout + n] + (double)(float)gol[(size_t)m * Cout + n];
      if (n >= cout) { pad = std::fmax(pad, std::fabs(got)); continue; }
      double y = bias[n];
      for (int c = 0; c < cexp; ++c) y += xv[(size_t)m * C + c] * gate[c] * (double)w[(size_t)n * C + c];
      if (residual) y += rv[(size_t)m * Cout + n];
      emax = std::fmax(emax, std::fabs(got - y));
      rmax = std::fmax(rmax, std::fabs(y));
    }
  const bool wide = M >= 12800;
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
  // the squeeze FC handed over by the front half
  std::vector<unsigned long long> zs((size_t)replicas * 64, 0ull);
  {
    std::vector<double> z(sq, 0.0);
    for (int q = 0; q < sq; ++q)
      for (int c = 0; c < cexp; ++c) z[q] += (double)w1[(size_t)q * C + c] * mean[c] * M;
    for (int r = 0; r < replicas; ++r)
      for (int q = 0; q < sq; ++q) zs[(size_t)r * 64 + q] = (unsigned long long)(long long)std::llrint(z[q] / replicas * 16777216.0);
  }
  MbBackParams pz = p;
  pz.zsums = up(zs);
  launch_mbconv_back(pz, 0);
  hipDeviceSynchronize();
  hipMemcpy(goh.data(), oh, goh.size() * 2, hipMemcpyDeviceToHost);
  hipMemcpy(gol.data(), ol, gol.size() * 2, hipMemcpyDeviceToHost);
  double ezmax = 0;
  for (int m = 0; m < M; ++m)
    for (int n = 0; n < cout; ++n) {
      double y = bias[n];
      for (int c = 0; c < cexp; ++c) y += xv[(size_t)m * C + c] * gate[c] * (double)w[(size_t)n * C + c];
      if (residual) y += rv[(size_t)m * Cout + n];
      ezmax = std::fmax(ezmax, std::fabs((double)(float)goh[(size_t)m * Cout + n] + (double)(float)gol[(size_t)m * Cout + n] - y));
    }
  const float tz = wide ? time_it([&] { return launch_mbb<4, 4, 0>(pz, 0); }) : time_it([&] { return launch_mbb<1, 8, 0>(pz, 0); });
  const float tzg = wide ? time_it([&] { return launch_mbb<4, 4, 2>(pz, 0); }) : time_it([&] { return launch_mbb<1, 8, 2>(pz, 0); });
  if (M >= 12800 && M < 32768)   // the three workgroup shapes on the 80x160 maps (the library takes <4, 4>)
    std::printf("      80x160: <WM 4, 4 waves> %5.1f us | <2, 8> %5.1f | <1, 8> %5.1f\n", time_it([&] { return launch_mbb<4, 4, 0>(pz, 0); }), time_it([&] { return launch_mbb<2, 8, 0>(pz, 0); }),
                time_it([&] { return launch_mbb<1, 8, 0>(pz, 0); }));
#define T_(ABL) (wide ? time_it([&] { return launch_mbb<4, 4, ABL>(p, 0); }) : time_it([&] { return launch_mbb<1, 8, ABL>(p, 0); }))
  std::printf("%4d -> %3d  sq %2d  %3dx%-3d res %d ws %4.2f | err %.1e rel %.1e pad %.0e | zsums: err %.1e %5.1f us (noGEMM %5.1f) | here: %5.1f us | noGate %5.1f | noGEMM %5.1f | neither %5.1f | noGEMM: -means %5.1f  -fc1 %5.1f  -fc2 %5.1f  "
              "means only %5.1f  fc1 only %5.1f  fc2 only %5.1f\n", cexp, cout, sq, H, W, (int)residual, wscale, emax, emax / rmax, pad, ezmax, tz, tzg, T_(0), T_(1), T_(2), T_(3), T_(2 + 4), T_(2 + 8), T_(2 + 16),
              T_(2 + 8 + 16), T_(2 + 4 + 16), T_(2 + 4 + 8));
}

int main() {
  run(32, 16, 8, 160, 320, false, 1.f);
  run(96, 24, 4, 80, 160, false, 1.f);
  run(144, 24, 6, 80, 160, true, 1.f);
  run(144, 40, 6, 40, 80, false, 1.f);
  run(240, 40, 10, 40, 80, true, 1.f);
  run(240, 80, 10, 20, 40, false, 1.f);
  run(480, 80, 20, 20, 40, true, 1.f);
  run(480, 112, 20, 20, 40, false, 1.f);
  run(672, 112, 28, 20, 40, true, 1.f);
  run(672, 192, 28, 10, 20, false, 1.f);
  run(1152, 192, 48, 10, 20, true, 1.f);
  run(1152, 192, 48, 10, 20, false, 1.f);
  run(1152, 192, 48, 10, 20, true, 0.1f);
  run(1152, 192, 48, 10, 20, true, 0.01f);
  run(1152, 320, 48, 10, 20, false, 1.f);
  return 0;
}
