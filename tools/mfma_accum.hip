// Developer tool (round 6): HOW does v_mfma_f32_32x32x16_f16 round when it accumulates?  The parity sweep's worst rows are
// farther from an fp64 evaluation than the fp32 reference is, and the storage format explains only a part of it (profiles/r06_pair_storage_study.tsv).
// One wave accumulates D = sum over N chained MFMAs of A_i (32 x 16) * B_i (16 x 32), fp16 operands with products that are exact in fp32; the host forms
//   exact   : the same sum in fp64 (every product of two fp16 values is exact in fp64),
//   rne_seq : fp32, one correctly rounded addition per PRODUCT in k order (what a scalar fp32 loop does),
// and prints the error of the device result and of rne_seq against `exact`, as max and as mean SIGNED error in units of 2^-24 x |exact| -- a rounding
// rule that truncates shows up as a signed bias growing with N, round-to-nearest as a zero-mean walk growing with sqrt(N).
// Result (profiles/r06_mfma_accum.txt): all-positive operands, N = 4096: device max 92 / bias -10.7, scalar loop max 185 / bias -10.1 -- the matrix
// pipe accumulates at least as accurately as a scalar fp32 loop and shows no truncation bias.
//   hipcc --offload-arch=gfx950 -O2 -std=c++17 tools/mfma_accum.hip -o tools/_mfma_accum && tools/_mfma_accum
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <vector>

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f16v __attribute__((ext_vector_type(16)));

__global__ void chain(const _Float16* A, const _Float16* B, float* D, int n) {
  // A: [n][32 rows][16 k], B: [n][32 cols][16 k]; lane l: row / column l % 32, k = 8 * (l / 32) + 0..7
  const int lane = threadIdx.x;
  f16v c = {};
  for (int i = 0; i < n; ++i) {
    const h8 a = *reinterpret_cast<const h8*>(A + ((size_t)i * 32 + (lane & 31)) * 16 + 8 * (lane >> 5));
    const h8 b = *reinterpret_cast<const h8*>(B + ((size_t)i * 32 + (lane & 31)) * 16 + 8 * (lane >> 5));
    c = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
  }
  for (int r = 0; r < 16; ++r) D[(8 * (r / 4) + 4 * (lane >> 5) + (r % 4)) * 32 + (lane & 31)] = c[r];   // D[row m][col n]
}

int main() {
  for (int positive = 0; positive < 2; ++positive)
    for (int n : {16, 64, 256, 1024, 4096}) {
      std::vector<_Float16> A((size_t)n * 32 * 16), B(A.size());
      unsigned s = 99u + n + 7919u * positive;
      auto rnd = [&] {
        s = s * 1664525u + 1013904223u;
        const float v = ((int)(s >> 9) % 2001 - 1000) * 0.001f;
        return (_Float16)(positive ? std::fabs(v) + 0.01f : v);   // all-positive products: a truncating adder shows a one-sided bias at once
      };
      for (auto& v : A) v = rnd();
      for (auto& v : B) v = rnd();
      _Float16 *dA, *dB;
      float* dD;
      hipMalloc(&dA, A.size() * 2);
      hipMalloc(&dB, B.size() * 2);
      hipMalloc(&dD, 32 * 32 * 4);
      hipMemcpy(dA, A.data(), A.size() * 2, hipMemcpyHostToDevice);
      hipMemcpy(dB, B.data(), B.size() * 2, hipMemcpyHostToDevice);
      hipLaunchKernelGGL(chain, dim3(1), dim3(64), 0, 0, dA, dB, dD, n);
      std::vector<float> D(32 * 32);
      hipMemcpy(D.data(), dD, D.size() * 4, hipMemcpyDeviceToHost);
      double emax[2] = {0, 0}, ebias[2] = {0, 0};
      for (int m = 0; m < 32; ++m)
        for (int c = 0; c < 32; ++c) {
          double exact = 0.0;
          float seq = 0.0f;
          for (int i = 0; i < n; ++i) {
            double part = 0.0;
            for (int k = 0; k < 16; ++k) {
              const float p = (float)A[((size_t)i * 32 + m) * 16 + k] * (float)B[((size_t)i * 32 + c) * 16 + k];   // exact in fp32
              part += (double)p;
              seq += p;
            }
            exact += part;
          }
          const double unit = std::ldexp(std::fabs(exact), -24) + 1e-300;
          const double e[2] = {(double)D[m * 32 + c] - exact, (double)seq - exact};
          for (int q = 0; q < 2; ++q) {
            emax[q] = std::fmax(emax[q], std::fabs(e[q]) / unit);
            ebias[q] += e[q] / unit / 1024.0 * (exact >= 0 ? 1.0 : -1.0);   // signed towards / away from zero: negative = magnitude lost
          }
        }
      if (n == 16) std::printf("# %s operands\n# mfmas\tdevice max / bias\trne per product max / bias   (units of 2^-24 |exact|)\n", positive ? "all-positive" : "signed");
      std::printf("%d\t%.2f / %+.2f\t%.2f / %+.2f\n", n, emax[0], ebias[0], emax[1], ebias[1]);
      hipFree(dA);
      hipFree(dB);
      hipFree(dD);
    }
  return 0;
}
