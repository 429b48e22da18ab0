// Squeeze-excite, phases 1 and 2, shared by se_gate_scale_kernel (kernels_backbone.hip) and mbconv_back_kernel (kernels_mbconv.hip): a
// workgroup of NT (256 or 512) threads rebuilds the channel means from the replica rows of the fused pool and ALL squeeze units itself
// (torchvision SqueezeExcitation: avgpool -> fc1 -> SiLU; Models/model_components/backbone.py:9-22).
#pragma once
#include "kernels.hpp"

namespace vp {

// acc: [acc_rows][C] u64 (acc_rows = se_acc_rows(C, NT)), mean: [C] floats (16-byte aligned), red: [NT] floats, s1: [64] floats.
// On return (after the trailing barrier) mean[] and s1[0 .. sq) are valid for every thread; s1[sq .. 64) is zero.
__host__ __device__ static inline int se_acc_rows(int C, int NT = 256) { return (C >> 1) >= NT ? 1 : NT / (C >> 1); }

template <int NT = 256>
__device__ __forceinline__ void se_means(const SeParams& se, unsigned long long* acc, float* mean) {
  const int tid = threadIdx.x, C = se.C;
  // ---- 1: means.  Thread = (channel pair, slice of the replica rows): every load of a thread is independent (one or two round
  // trips), the slices of a pair meet in LDS; integer sums, so any grouping gives the same bits.  (LDS atomics per loaded pair
  // serialise -- load -> ds_add_u64 chains, 11 us in round 1 -- and a thread walking its channel's 8..64 rows alone is 8 trips.)
  {
    typedef unsigned long long u64x2 __attribute__((ext_vector_type(2)));
    const int CP = C >> 1;                                  // 16-byte pairs per row
    const int TPC = CP >= NT ? 1 : NT / CP;               // threads per pair
    const int rows_per = (se.replicas + TPC - 1) / TPC;
    const u64x2* src = reinterpret_cast<const u64x2*>(se.sums);
    for (int cp = tid % (TPC == 1 ? NT : CP); cp < CP; cp += NT) {
      const int g = TPC == 1 ? 0 : tid / CP;
      if (g >= TPC) break;
      const int r_begin = g * rows_per, r_end = min(se.replicas, r_begin + rows_per);
      unsigned long long a0 = 0ull, a1 = 0ull;
      int r = r_begin;
      for (; r + 8 <= r_end; r += 8) {
        u64x2 v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = src[(size_t)(r + u) * CP + cp];
#pragma unroll
        for (int u = 0; u < 8; ++u) { a0 += v[u][0]; a1 += v[u][1]; }
      }
      for (; r < r_end; ++r) {
        const u64x2 v = src[(size_t)r * CP + cp];
        a0 += v[0];
        a1 += v[1];
      }
      acc[(size_t)g * C + 2 * cp] = a0;
      acc[(size_t)g * C + 2 * cp + 1] = a1;
      if (TPC > 1) break;  // one pair per thread in the sliced form
    }
    __syncthreads();
    for (int c = tid; c < C; c += NT) {
      unsigned long long t = acc[c];
      for (int g = 1; g < TPC; ++g) t += acc[(size_t)g * C + c];
      mean[c] = (float)((double)(long long)t * (1.0 / 16777216.0)) * se.inv_hw;
    }
  }
  __syncthreads();
}

template <int NB = 16, int NT = 256>  // NB: 16-byte loads of the squeeze FC in flight per thread; NT: threads of the workgroup
__device__ __forceinline__ void se_means_squeeze(const SeParams& se, unsigned long long* acc, float* mean, float* red, float* s1) {
  const int tid = threadIdx.x, C = se.C;
  se_means<NT>(se, acc, mean);
  // ---- 2: squeeze FC
  {
    const int nseg = NT / se.sq;                       // >= 4 (sq <= 64)
    const int j = tid / nseg, sg = tid - j * nseg;
    const int C4 = C >> 2, per = (C4 + nseg - 1) / nseg;
    float s = 0.f;
    if (j < se.sq) {
      const f32x4_t* wr = reinterpret_cast<const f32x4_t*>(se.w1 + (size_t)j * C);
      const f32x4_t* m4 = reinterpret_cast<const f32x4_t*>(mean);
      const int q1 = min(C4, (sg + 1) * per);
      // up to 58 16-byte loads per thread (sq = 48, C = 1152): 16 in flight at a time -- at 4 the phase was 15 serial round trips
      int q = sg * per;
      for (; q + NB <= q1; q += NB) {
        f32x4_t a[NB];
#pragma unroll
        for (int u = 0; u < NB; ++u) a[u] = wr[q + u];
#pragma unroll
        for (int u = 0; u < NB; ++u) {
          const f32x4_t m = m4[q + u];
          s += (a[u][0] * m[0] + a[u][1] * m[1]) + (a[u][2] * m[2] + a[u][3] * m[3]);
        }
      }
#pragma unroll 4
      for (; q < q1; ++q) {
        const f32x4_t a = wr[q], m = m4[q];
        s += (a[0] * m[0] + a[1] * m[1]) + (a[2] * m[2] + a[3] * m[3]);
      }
    }
    red[tid] = s;
    __syncthreads();
    if (tid < se.sq) {
      float t = 0.f;
      for (int g = 0; g < nseg; ++g) t += red[tid * nseg + g];
      s1[tid] = silu_f(t + se.b1[tid]);
    } else if (tid < 64) {
      s1[tid] = 0.0f;
    }
  }
  __syncthreads();
}

}  // namespace vp
