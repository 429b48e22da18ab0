// 16-byte activation accessors shared by the HBM-bound kernels: 8 channels of one pixel, value = hi (+ lo).
#pragma once
#include "kernels.hpp"

namespace vp {

__device__ __forceinline__ void load8(const ActView& a, size_t off, float v[8]) {
  const h8_t h = *reinterpret_cast<const h8_t*>(a.hi + off);
#pragma unroll
  for (int i = 0; i < 8; ++i) v[i] = (float)h[i];
  if (a.lo) {
    const h8_t l = *reinterpret_cast<const h8_t*>(a.lo + off);
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] += (float)l[i];
  }
}
__device__ __forceinline__ void store8(const ActView& a, size_t off, const float v[8]) {
  h8_t h;
#pragma unroll
  for (int i = 0; i < 8; ++i) h[i] = (half_t)v[i];
  *reinterpret_cast<h8_t*>(a.hi + off) = h;
  if (a.lo) {
    h8_t l;
#pragma unroll
    for (int i = 0; i < 8; ++i) l[i] = (half_t)(v[i] - (float)h[i]);
    *reinterpret_cast<h8_t*>(a.lo + off) = l;
  }
}

#define VP_LAUNCH(kern, grid, block, shmem, st, ...)               \
  do {                                                             \
    hipLaunchKernelGGL(kern, grid, block, shmem, st, __VA_ARGS__); \
    return hipGetLastError();                                      \
  } while (0)

static inline unsigned nblk(long long n) { return (unsigned)((n + 255) / 256); }

// lane layout of the pixel-slab kernels (depthwise+pool, pool): channel-octet lanes x pixel lanes = 256
static inline int slab_cgl(int C) {
  const int CG = C >> 3;
  return CG >= 32 ? 32 : (CG >= 16 ? 16 : (CG >= 8 ? 8 : 4));
}

}  // namespace vp
