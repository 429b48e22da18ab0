// Developer tool: do a wave's VALU instructions overlap another wave's MFMAs on the same SIMD?
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/mfma_valu_overlap.hip -o tools/_mfma_valu_overlap
// 512 workgroups x 256 threads = 2 per CU, one wave of each per SIMD.  mode 0: both run the MFMA loop; 1: both VALU;
// 2: the workgroup in threadgroup slot 0 runs MFMA, the one in slot 1 VALU (HW_ID.TG_ID parity); 3: MFMA only on slot 0
// (slot 1 exits); 4: VALU only on slot 1; 5: transcendental loop on slot 1 + MFMA on slot 0.
#include <hip/hip_runtime.h>

#include <cstdio>

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f16v __attribute__((ext_vector_type(16)));

__device__ float run_mfma(int iters) {
  h8 a, b;
  for (int e = 0; e < 8; ++e) {
    a[e] = (_Float16)(0.001f * (threadIdx.x + e));
    b[e] = (_Float16)(0.002f * (threadIdx.x - e));
  }
  f16v c0 = {}, c1 = {}, c2 = {}, c3 = {};
  for (int i = 0; i < iters; ++i) {
    c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c0, 0, 0, 0);
    c1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c1, 0, 0, 0);
    c2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c2, 0, 0, 0);
    c3 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c3, 0, 0, 0);
  }
  float s = 0;
  for (int e = 0; e < 16; ++e) s += c0[e] + c1[e] + c2[e] + c3[e];
  return s;
}
__device__ float run_valu(int iters) {  // 32 independent fp32 FMAs per iteration = 128 issue cycles
  float x[8];
  for (int e = 0; e < 8; ++e) x[e] = 0.001f * (threadIdx.x + e);
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int e = 0; e < 8; ++e) x[e] = __builtin_fmaf(x[e], 0.999f, 0.001f);
  }
  float s = 0;
  for (int e = 0; e < 8; ++e) s += x[e];
  return s;
}
__device__ float run_trans(int iters) {  // 8 exp + 8 rcp per iteration
  float x[8];
  for (int e = 0; e < 8; ++e) x[e] = 0.001f * (threadIdx.x + e);
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int e = 0; e < 8; ++e) x[e] = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-x[e]));
  }
  float s = 0;
  for (int e = 0; e < 8; ++e) s += x[e];
  return s;
}

__device__ float run_both(int iters, int nvalu) {  // same wave: 4 MFMAs + 8*nvalu independent FMAs per iteration
  h8 a, b;
  for (int e = 0; e < 8; ++e) {
    a[e] = (_Float16)(0.001f * (threadIdx.x + e));
    b[e] = (_Float16)(0.002f * (threadIdx.x - e));
  }
  f16v c0 = {}, c1 = {}, c2 = {}, c3 = {};
  float x[8];
  for (int e = 0; e < 8; ++e) x[e] = 0.001f * (threadIdx.x + e);
  for (int i = 0; i < iters; ++i) {
    c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c0, 0, 0, 0);
    if (nvalu > 0)
#pragma unroll
      for (int e = 0; e < 8; ++e) x[e] = __builtin_fmaf(x[e], 0.999f, 0.001f);
    c1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c1, 0, 0, 0);
    if (nvalu > 1)
#pragma unroll
      for (int e = 0; e < 8; ++e) x[e] = __builtin_fmaf(x[e], 0.999f, 0.001f);
    c2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c2, 0, 0, 0);
    if (nvalu > 2)
#pragma unroll
      for (int e = 0; e < 8; ++e) x[e] = __builtin_fmaf(x[e], 0.999f, 0.001f);
    c3 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c3, 0, 0, 0);
    if (nvalu > 3)
#pragma unroll
      for (int e = 0; e < 8; ++e) x[e] = __builtin_fmaf(x[e], 0.999f, 0.001f);
  }
  float s = 0;
  for (int e = 0; e < 16; ++e) s += c0[e] + c1[e] + c2[e] + c3[e];
  for (int e = 0; e < 8; ++e) s += x[e];
  return s;
}

__global__ __launch_bounds__(256) void overlap(float* out, int mode, int iters) {
  const unsigned tg = __builtin_amdgcn_s_getreg((3 << 11) | (16 << 6) | 4) & 1;
  float s = 0;
  if (mode == 0) s = run_mfma(iters);
  if (mode == 1) s = run_valu(iters);
  if (mode == 2) s = tg ? run_valu(iters) : run_mfma(iters);
  if (mode == 3 && !tg) s = run_mfma(iters);
  if (mode == 4 && tg) s = run_valu(iters);
  if (mode == 5) s = tg ? run_trans(iters) : run_mfma(iters);
  if (mode == 6 && tg) s = run_trans(iters);
  if (mode == 7 && !tg) s = run_both(iters, 4);
  if (mode == 8) s = run_both(iters, 4);
  if (mode == 9 && !tg) s = run_both(iters, 2);
  if (mode == 10) s = run_both(iters, 2);
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

int main() {
  float* out;
  hipMalloc(&out, 512 * 256 * 4);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  const char* names[] = {"MFMA + MFMA", "VALU + VALU", "MFMA + VALU", "MFMA alone (1 wave/SIMD)", "VALU alone (1 wave/SIMD)", "MFMA + transcendental",
                         "transcendental alone", "same wave 4 MFMA + 32 FMA, 1 wave/SIMD", "same wave 4 MFMA + 32 FMA, 2 waves/SIMD",
                         "same wave 4 MFMA + 16 FMA, 1 wave/SIMD", "same wave 4 MFMA + 16 FMA, 2 waves/SIMD"};
  for (int mode = 0; mode < 11; ++mode) {
    const int iters = 20000;
    hipLaunchKernelGGL(overlap, dim3(512), dim3(256), 0, 0, out, mode, 1000);
    hipDeviceSynchronize();
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL(overlap, dim3(512), dim3(256), 0, 0, out, mode, iters);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    std::printf("mode %d %-28s %.3f ms\n", mode, names[mode], ms);
  }
  return 0;
}
