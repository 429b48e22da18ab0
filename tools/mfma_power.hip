// Developer tool: is the fp16 matrix pipe of THIS GPU clock-limited by power?  A bare v_mfma_f32_32x32x16_f16 loop (random fp16 operands,
// 2 waves per SIMD, 4 independent accumulators per wave: the pipe issues back to back) on N of the 256 CUs, N = 32 .. 256: TFLOP/s, TFLOP/s
// per active CU and the shader clock read inside the kernel (s_memtime ticks against the 100 MHz s_memrealtime).  If the chip holds its
// clock, TFLOP/s grows linearly with N at 9.8 TFLOP/s per CU (2.4 GHz x 4 SIMDs x 1024 FLOP/cycle); if it is power-limited, the clock
// falls as N grows and the last CUs add little -- the regime in which filling idle CUs (stream-K, more frames in flight) cannot pay.
// Also: the same loop with all-zero operands (no toggling -> no throttle) as the control.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/mfma_power.hip -o tools/_mfma_power
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstring>
#include <vector>

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f16v __attribute__((ext_vector_type(16)));

__global__ __launch_bounds__(512, 2) void mfma_loop(const h8* src, float* out, long long* clk, int iters) {
  const h8 a = src[threadIdx.x], b = src[512 + threadIdx.x];
  f16v c0 = {}, c1 = {}, c2 = {}, c3 = {};
  const long long t0 = __builtin_readcyclecounter(), r0 = wall_clock64();
  for (int i = 0; i < iters; ++i) {
    c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c0, 0, 0, 0);
    c1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(b, a, c1, 0, 0, 0);
    c2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c2, 0, 0, 0);
    c3 = __builtin_amdgcn_mfma_f32_32x32x16_f16(b, a, c3, 0, 0, 0);
  }
  const long long t1 = __builtin_readcyclecounter(), r1 = wall_clock64();
  float s = 0;
  for (int e = 0; e < 16; ++e) s += c0[e] + c1[e] + c2[e] + c3[e];
  out[blockIdx.x * 512 + threadIdx.x] = s;
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    clk[0] = t1 - t0;
    clk[1] = r1 - r0;
  }
}

// Round 4: the same question for v_mfma_f32_16x16x32_f16 (half the FLOPs per instruction, a quarter of the accumulator registers): does the other
// tile shape of the fp16 pipe draw less per FLOP, i.e. sustain a higher rate under the power limit?  Eight independent accumulators per wave.
typedef float f4v __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(512, 2) void mfma16_loop(const h8* src, float* out, long long* clk, int iters) {
  const h8 a = src[threadIdx.x], b = src[512 + threadIdx.x];
  f4v c[8] = {};
  const long long t0 = __builtin_readcyclecounter(), r0 = wall_clock64();
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int k = 0; k < 8; ++k) c[k] = __builtin_amdgcn_mfma_f32_16x16x32_f16((k & 1) ? b : a, (k & 1) ? a : b, c[k], 0, 0, 0);
  }
  const long long t1 = __builtin_readcyclecounter(), r1 = wall_clock64();
  float s = 0;
  for (int k = 0; k < 8; ++k)
    for (int e = 0; e < 4; ++e) s += c[k][e];
  out[blockIdx.x * 512 + threadIdx.x] = s;
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    clk[0] = t1 - t0;
    clk[1] = r1 - r0;
  }
}


// Round 6: the parity mode's product as the pipelined kernels issue it -- per accumulator tile lo*hi, hi*lo, hi*hi, three
// dependent MFMAs back to back -- with (hi, lo) planes of random fp32 values, the `lo` planes' mantissas truncated to KEEP explicit bits (10 = as
// split today).  If the sustained rate on 256 CUs rises as the low planes carry fewer toggling bits, truncating them at load / in the epilogues buys
// clock on a power-limited pipe (product error 2^-(11 + KEEP + 1) relative instead of 2^-22).  src: [a_hi | a_lo | b_hi | b_lo] x 512 lanes.
__global__ __launch_bounds__(512, 2) void mfma_x3_loop(const h8* src, float* out, long long* clk, int iters) {
  const h8 ah = src[threadIdx.x], al = src[512 + threadIdx.x], bh = src[1024 + threadIdx.x], bl = src[1536 + threadIdx.x];
  f16v c0 = {}, c1 = {}, c2 = {}, c3 = {};
  const long long t0 = __builtin_readcyclecounter(), r0 = wall_clock64();
  for (int i = 0; i < iters; ++i) {
    c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh, c0, 0, 0, 0);
    c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl, c0, 0, 0, 0);
    c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, c0, 0, 0, 0);
    c1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(bl, ah, c1, 0, 0, 0);
    c1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(bh, al, c1, 0, 0, 0);
    c1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(bh, ah, c1, 0, 0, 0);
    c2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh, c2, 0, 0, 0);
    c2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl, c2, 0, 0, 0);
    c2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, c2, 0, 0, 0);
    c3 = __builtin_amdgcn_mfma_f32_32x32x16_f16(bl, ah, c3, 0, 0, 0);
    c3 = __builtin_amdgcn_mfma_f32_32x32x16_f16(bh, al, c3, 0, 0, 0);
    c3 = __builtin_amdgcn_mfma_f32_32x32x16_f16(bh, ah, c3, 0, 0, 0);
  }
  const long long t1 = __builtin_readcyclecounter(), r1 = wall_clock64();
  float s = 0;
  for (int e = 0; e < 16; ++e) s += c0[e] + c1[e] + c2[e] + c3[e];
  out[blockIdx.x * 512 + threadIdx.x] = s;
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    clk[0] = t1 - t0;
    clk[1] = r1 - r0;
  }
}

// The same loop with R 16-byte LDS reads (ds_read_b128, conflict-free, random data: the operands of the MFMAs) per 4 MFMAs -- what a
// register tile costs in LDS traffic: the parity 3x3 kernel reads 8 fragments per 12 MFMAs (R = 2.7); a 2x larger register tile would
// read 1.3.  If the pipe is POWER-limited, LDS energy comes out of the MFMA budget and the rate falls with R.
template <int R>
__global__ __launch_bounds__(512, 2) void mfma_lds_loop(const h8* src, float* out, long long* clk, int iters) {
  __shared__ h8 lds[4096];
  for (int i = threadIdx.x; i < 4096; i += 512) lds[i] = src[i & 1023];
  __syncthreads();
  h8 a = src[threadIdx.x], b = src[512 + threadIdx.x], a2 = b, b2 = a;
  f16v c0 = {}, c1 = {}, c2 = {}, c3 = {};
  const long long t0 = __builtin_readcyclecounter(), r0 = wall_clock64();
  for (int i = 0; i < iters; ++i) {
    if (R >= 1) a = lds[(i * 67 + threadIdx.x) & 4095];
    if (R >= 2) b = lds[(i * 131 + 1024 + threadIdx.x) & 4095];
    if (R >= 3) a2 = lds[(i * 197 + 2048 + threadIdx.x) & 4095];
    if (R >= 4) b2 = lds[(i * 263 + 3072 + threadIdx.x) & 4095];
    c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c0, 0, 0, 0);
    c1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(b2, a2, c1, 0, 0, 0);
    c2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a2, b, c2, 0, 0, 0);
    c3 = __builtin_amdgcn_mfma_f32_32x32x16_f16(b2, a, c3, 0, 0, 0);
  }
  const long long t1 = __builtin_readcyclecounter(), r1 = wall_clock64();
  float s = 0;
  for (int e = 0; e < 16; ++e) s += c0[e] + c1[e] + c2[e] + c3[e];
  out[blockIdx.x * 512 + threadIdx.x] = s;
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    clk[0] = t1 - t0;
    clk[1] = r1 - r0;
  }
}

template <int R>
static void lds_row(const h8* src, float* out, long long* clk, hipEvent_t e0, hipEvent_t e1, int n) {
  const int iters = 40000;
  hipLaunchKernelGGL(mfma_lds_loop<R>, dim3(n), dim3(512), 0, 0, src, out, clk, 4000);
  hipDeviceSynchronize();
  hipEventRecord(e0, 0);
  hipLaunchKernelGGL(mfma_lds_loop<R>, dim3(n), dim3(512), 0, 0, src, out, clk, iters);
  hipEventRecord(e1, 0);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  const double flop = (double)n * 8 * iters * 4 * 32 * 32 * 16 * 2;
  std::printf("%d\t%d\t%.2f\t%.3f\t%.1f\t%.2f\t%.0f\n", n, R, R / 4.0, ms, flop / ms * 1e-9, flop / ms * 1e-9 / n, (double)clk[0] / clk[1] * 100.0);
}

int main() {
  float* out;
  long long* clk;
  h8* src;
  hipMalloc(&out, 256 * 512 * 4);
  hipMallocManaged(&clk, 16);
  hipMalloc(&src, 1024 * sizeof(h8));
  std::vector<_Float16> h(1024 * 8);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  for (int zero = 0; zero < 2; ++zero) {
    unsigned s = 12345;
    for (auto& v : h) {
      s = s * 1664525u + 1013904223u;
      v = zero ? (_Float16)0.0f : (_Float16)(((int)(s >> 9) % 2001 - 1000) * 0.001f);
    }
    hipMemcpy(src, h.data(), h.size() * 2, hipMemcpyHostToDevice);
    std::printf("# %s operands\n# active_CUs\tms\tTFLOP/s\tTFLOP/s_per_CU\tshader_clock_MHz\n", zero ? "all-zero" : "random fp16");
    for (int n : {32, 64, 128, 192, 200, 224, 256}) {
      const int iters = 40000;
      hipLaunchKernelGGL(mfma_loop, dim3(n), dim3(512), 0, 0, src, out, clk, 4000);
      hipDeviceSynchronize();
      hipEventRecord(e0, 0);
      hipLaunchKernelGGL(mfma_loop, dim3(n), dim3(512), 0, 0, src, out, clk, iters);
      hipEventRecord(e1, 0);
      hipEventSynchronize(e1);
      float ms;
      hipEventElapsedTime(&ms, e0, e1);
      const double flop = (double)n * 8 * iters * 4 * 32 * 32 * 16 * 2;
      std::printf("%d\t%.3f\t%.1f\t%.2f\t%.0f\n", n, ms, flop / ms * 1e-9, flop / ms * 1e-9 / n, (double)clk[0] / clk[1] * 100.0);
    }
  }
  {
    unsigned s = 12345;
    for (auto& v : h) {
      s = s * 1664525u + 1013904223u;
      v = (_Float16)(((int)(s >> 9) % 2001 - 1000) * 0.001f);
    }
    hipMemcpy(src, h.data(), h.size() * 2, hipMemcpyHostToDevice);
    std::printf("# random fp16 operands, v_mfma_f32_16x16x32_f16 (8 accumulators per wave)\n# active_CUs\tms\tTFLOP/s\tTFLOP/s_per_CU\tshader_clock_MHz\n");
    for (int n : {64, 256}) {
      const int iters = 40000;
      hipLaunchKernelGGL(mfma16_loop, dim3(n), dim3(512), 0, 0, src, out, clk, 4000);
      hipDeviceSynchronize();
      hipEventRecord(e0, 0);
      hipLaunchKernelGGL(mfma16_loop, dim3(n), dim3(512), 0, 0, src, out, clk, iters);
      hipEventRecord(e1, 0);
      hipEventSynchronize(e1);
      float ms;
      hipEventElapsedTime(&ms, e0, e1);
      const double flop = (double)n * 8 * iters * 8 * 16 * 16 * 32 * 2;
      std::printf("%d\t%.3f\t%.1f\t%.2f\t%.0f\n", n, ms, flop / ms * 1e-9, flop / ms * 1e-9 / n, (double)clk[0] / clk[1] * 100.0);
    }
  }
  {
    unsigned s = 12345;
    for (auto& v : h) {
      s = s * 1664525u + 1013904223u;
      v = (_Float16)(((int)(s >> 9) % 2001 - 1000) * 0.001f);
    }
    hipMemcpy(src, h.data(), h.size() * 2, hipMemcpyHostToDevice);
    std::printf("# random fp16 operands read from LDS: R ds_read_b128 per 4 MFMAs\n# active_CUs\tR\treads_per_MFMA\tms\tTFLOP/s\tTFLOP/s_per_CU\tshader_clock_MHz\n");
    for (int n : {64, 256}) {
      lds_row<0>(src, out, clk, e0, e1, n);
      lds_row<1>(src, out, clk, e0, e1, n);
      lds_row<2>(src, out, clk, e0, e1, n);
      lds_row<3>(src, out, clk, e0, e1, n);
      lds_row<4>(src, out, clk, e0, e1, n);
    }
  }
  {
    // round 6: (hi, lo) planes, low planes truncated to KEEP explicit mantissa bits; -1 = low planes all zero (control: two of three MFMAs multiply by 0)
    h8* src4;
    hipMalloc(&src4, 2048 * sizeof(h8));
    std::vector<_Float16> q(2048 * 8);
    std::printf("# parity-mode product (lo*hi, hi*lo, hi*hi per accumulator), random fp32 values split into (hi, lo) fp16 planes, low planes truncated\n"
                "# keep_bits\tactive_CUs\tms\tTFLOP/s_issued\tTFLOP/s_per_CU\tshader_clock_MHz\n");
    for (int keep : {10, 8, 6, 5, 4, 2, 0, -1}) {
      unsigned s = 777;
      for (int plane = 0; plane < 2; ++plane)        // a, b
        for (int i = 0; i < 512 * 8; ++i) {
          s = s * 1664525u + 1013904223u;
          const float x = ((int)(s >> 8) % 2000001 - 1000000) * 1e-6f;
          const _Float16 hi = (_Float16)x;
          _Float16 lo = (_Float16)(x - (float)hi);
          unsigned short bits;
          std::memcpy(&bits, &lo, 2);
          if (keep < 0) bits = 0;
          else bits &= (unsigned short)~((1u << (10 - keep)) - 1u);
          std::memcpy(&lo, &bits, 2);
          q[(size_t)(plane * 2 + 0) * 512 * 8 + i] = hi;
          q[(size_t)(plane * 2 + 1) * 512 * 8 + i] = lo;
        }
      hipMemcpy(src4, q.data(), q.size() * 2, hipMemcpyHostToDevice);
      for (int n : {64, 256}) {
        const int iters = 14000;
        hipLaunchKernelGGL(mfma_x3_loop, dim3(n), dim3(512), 0, 0, src4, out, clk, 1400);
        hipDeviceSynchronize();
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL(mfma_x3_loop, dim3(n), dim3(512), 0, 0, src4, out, clk, iters);
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        const double flop = (double)n * 8 * iters * 12 * 32 * 32 * 16 * 2;
        std::printf("%d\t%d\t%.3f\t%.1f\t%.2f\t%.0f\n", keep, n, ms, flop / ms * 1e-9, flop / ms * 1e-9 / n, (double)clk[0] / clk[1] * 100.0);
      }
    }
    hipFree(src4);
  }
  return 0;
}
