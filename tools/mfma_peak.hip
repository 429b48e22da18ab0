// Developer tool: what does a bare MFMA loop reach on this GPU, and at which shader clock?
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/mfma_peak.hip -o tools/_mfma_peak
// Prints TFLOP/s of v_mfma_f32_32x32x16_f16 with 4 independent accumulators per wave for 1..3 waves per SIMD, and the
// shader clock derived from s_memtime (core cycles) vs s_memrealtime (100 MHz).
#include <hip/hip_runtime.h>

#include <cstdio>

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f16v __attribute__((ext_vector_type(16)));

__global__ __launch_bounds__(256) void mfma_loop(float* out, long long* clk, int iters) {
  h8 a, b;
  for (int e = 0; e < 8; ++e) {
    a[e] = (_Float16)(0.001f * (threadIdx.x + e));
    b[e] = (_Float16)(0.002f * (threadIdx.x - e));
  }
  f16v c0 = {}, c1 = {}, c2 = {}, c3 = {};
  const long long t0 = __builtin_readcyclecounter(), r0 = wall_clock64();
  for (int i = 0; i < iters; ++i) {
    c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c0, 0, 0, 0);
    c1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c1, 0, 0, 0);
    c2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c2, 0, 0, 0);
    c3 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c3, 0, 0, 0);
  }
  const long long t1 = __builtin_readcyclecounter(), r1 = wall_clock64();
  float s = 0;
  for (int e = 0; e < 16; ++e) s += c0[e] + c1[e] + c2[e] + c3[e];
  out[blockIdx.x * 256 + threadIdx.x] = s;
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    clk[0] = t1 - t0;
    clk[1] = r1 - r0;
  }
}

int main() {
  float* out;
  long long* clk;
  hipMalloc(&out, 256 * 16 * 256 * 4);
  hipMallocManaged(&clk, 16);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  for (int wps = 1; wps <= 3; ++wps) {
    const int blocks = 256 * wps, iters = 20000;
    hipLaunchKernelGGL(mfma_loop, dim3(blocks), dim3(256), 0, 0, out, clk, 2000);
    hipDeviceSynchronize();
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL(mfma_loop, dim3(blocks), dim3(256), 0, 0, out, clk, iters);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double flop = (double)blocks * 4 * iters * 4 * 32 * 32 * 16 * 2;
    std::printf("waves/SIMD %d: %.3f ms  %.1f TFLOP/s  | counter ticks %lld, realtime ticks %lld (100 MHz) -> counter at %.1f MHz\n", wps, ms,
                flop / ms * 1e-9, clk[0], clk[1], (double)clk[0] / clk[1] * 100.0);
  }
  return 0;
}
