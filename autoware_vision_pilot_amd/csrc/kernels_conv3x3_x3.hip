// 3x3 / stride 1 / pad 1 convolution of the PARITY mode (VP_FP16X3) on large maps: 8-wave workgroups, software-pipelined
// MFMA fragments.
//
// In fp16x3 every tensor is a (hi, lo) fp16 pair and every product is three MFMAs, so the halo kernel's LDS plan
// (kernels_conv3x3.hip) doubles: its 128-channel tile needs 90 KiB, ONE 4-wave workgroup per CU, one wave per SIMD --
// and that lone wave alternates "ds_read fragments -> wait -> MFMA -> stage -> barrier", leaving the matrix pipe idle for
// every LDS round trip (measured 265-290 TFLOP/s algorithmic = 32 % of the fp16 MFMA peak on the six big decoder layers,
// which are 45 % of a SceneSeg + Scene3D frame).  This kernel keeps the data flow (halo tile resident in LDS, one weight
// tile per tap, XCD-aware tile map, identical K order => bit-identical accumulation) and changes the execution shape:
//   * 512 threads = 8 waves on a 16x16-pixel x 128-channel tile: two waves per SIMD share the matrix pipe, each wave owns
//     64 channels x 64 pixels (2 x 2 MFMA tiles, 12 MFMAs per 16-channel K step);
//   * fragments are prefetched ONE K SUB-STEP AHEAD (the 8 ds_read_b128 of sub-step k+1 are issued before the 12 MFMAs
//     of sub-step k), across tap boundaries too: a THIRD weight buffer in LDS makes tap t+1's weights resident before
//     tap t starts, the next tap's pixel operand is the same halo image at a shifted address, and the next chunk's halo
//     is complete five taps before it is needed.  Only the barrier is left at a tap boundary;
//   * weights and halo pieces travel global -> registers (3-deep rings, issued 3 taps ahead) -> LDS as before;
//   * register epilogue: bias + exact-erf GELU + (hi, lo) split on the accumulators, BOTH fp16 planes staged once in LDS
//     and written as 256 contiguous bytes per pixel and plane.
// LDS: 2 x 2 x 25 920 (halo, double-buffered, two planes) + 3 x 2 x 8 192 (weights) = 152 832 B: one workgroup per CU.
#include <type_traits>

#include "conv_epilogue.hpp"

namespace vp {

template <int CO_TILE, int WCO, int WPX, int ACT>
__global__ __launch_bounds__(512) void conv3x3_x3w8_kernel(const ConvGemmParams p) {
  static_assert(WCO * WPX == 8, "8 waves");
  constexpr int TH = 16, TW = 16, ROWB = 80, HWD = TW + 2, HPX = (TH + 2) * HWD, PX = TH * TW;
  constexpr int HALO_BYTES = HPX * ROWB, WROW = 64, W_BYTES = CO_TILE * WROW;
  constexpr int HCHUNKS = HPX * 4, HP = (HCHUNKS + 511) / 512;
  constexpr int WCHUNKS = CO_TILE * 4, WP = (WCHUNKS + 511) / 512;
  constexpr int MT = CO_TILE / WCO / 32, NT = PX / WPX / 32;
  static_assert(MT >= 1 && NT >= 1 && HP <= 3 && WP == 1, "tile shape");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* const halo_base = smem;                       // [2][2 planes][HALO_BYTES]
  char* const w_base = smem + 4 * HALO_BYTES;         // [3][2 planes][W_BYTES]

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wco = wave / WPX, wpx = wave % WPX;
  const int tiles_x = (p.W + TW - 1) / TW;
  int vid;  // XCD-aware workgroup -> tile map (see kernels_conv3x3.hip)
  {
    const int nwg = gridDim.x, q = nwg >> 3, r = nwg & 7, xcd = blockIdx.x & 7;
    vid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (blockIdx.x >> 3);
  }
  const int n_px_tiles = tiles_x * ((p.H + TH - 1) / TH);
  const int tile_px = vid % n_px_tiles, tile_co = vid / n_px_tiles;
  const int tyi = tile_px / tiles_x, txi = tile_px - tyi * tiles_x;
  const int y0 = tyi * TH, x0 = txi * TW;
  const int co0 = tile_co * CO_TILE;
  const int KC = p.Cin >> 5;

  // ---- staging assignment
  int h_goff[HP], h_lds[HP];
#pragma unroll
  for (int pc = 0; pc < HP; ++pc) {
    const int hidx = tid + 512 * pc;
    const int hp = hidx >> 2, ch = hidx & 3;
    const int hy = hp / HWD, hx = hp - hy * HWD;
    const int gy = y0 - 1 + hy, gx = x0 - 1 + hx;
    const bool in_tile = hidx < HCHUNKS;
    const bool ok = in_tile && (unsigned)gy < (unsigned)p.H && (unsigned)gx < (unsigned)p.W;
    h_goff[pc] = ok ? (gy * p.W + gx) * p.Cin + ch * 8 : -1;
    h_lds[pc] = in_tile ? hp * ROWB + ch * 16 : -1;
  }
  const bool w_ok = tid < WCHUNKS;
  const int w_row = tid >> 2, w_ch = tid & 3;
  const int w_goff = w_ok ? (co0 + w_row) * 32 + w_ch * 8 : 0;
  const int w_lds = w_row * WROW + ((w_ch ^ ((w_row >> 2) & 3)) << 4);
  const size_t w_step = (size_t)p.CoutW * 32;
  typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
  const u32x4 zero4 = {0u, 0u, 0u, 0u};

  // ---- fragment addressing (same LDS image and lane maps as the halo kernel)
  int b_ofs[NT];
#pragma unroll
  for (int j = 0; j < NT; ++j) {
    const int q = (wpx * NT + j) * 32 + (lane & 31);
    int rowbit, px;
    lane_to_px16(q & 31, rowbit, px);
    b_ofs[j] = ((2 * (q >> 5) + rowbit) * HWD + px) * ROWB + (lane >> 5) * 16;
  }
  const int a_ofs = (wco * 32 + (lane & 31)) * WROW;
  const int a_swz = ((lane & 31) >> 2) & 3;
  const int a_sw[2] = {(((lane >> 5)) ^ a_swz) << 4, ((2 + (lane >> 5)) ^ a_swz) << 4};

  f32x16_t acc[MT][NT];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

  // fragment sets [K sub-step parity]: set 0 = channels 0..15 of the tap's 32, set 1 = channels 16..31
  h8_t fa[2][MT], fal[2][MT], fb[2][NT], fbl[2][NT];
  // staging rings (compile-time slots)
  u32x4 rw_hi[3], rw_lo[3], rh_hi[3], rh_lo[3];
#pragma unroll
  for (int r = 0; r < 3; ++r) rw_hi[r] = rw_lo[r] = rh_hi[r] = rh_lo[r] = zero4;
  const int s_last = KC * 9 - 1;

#define VP_LOAD_W(SLOT, SIDX)                                                                \
  if (w_ok) {                                                                                \
    const int si_ = (SIDX) < s_last ? (SIDX) : s_last;                                       \
    const size_t base_ = (size_t)si_ * w_step + w_goff;                                      \
    rw_hi[SLOT] = *reinterpret_cast<const u32x4*>(p.w_hi + base_);                           \
    rw_lo[SLOT] = *reinterpret_cast<const u32x4*>(p.w_lo + base_);                           \
  }
#define VP_STORE_W(SLOT, BUF)                                                                \
  if (w_ok) {                                                                                \
    char* dst_ = w_base + (BUF) * 2 * W_BYTES + w_lds;                                       \
    *reinterpret_cast<u32x4*>(dst_) = rw_hi[SLOT];                                           \
    *reinterpret_cast<u32x4*>(dst_ + W_BYTES) = rw_lo[SLOT];                                 \
  }
#define VP_LOAD_H(SLOT, PC, C)                                                               \
  {                                                                                          \
    const int g_ = h_goff[PC];                                                               \
    const int o_ = (g_ >= 0 ? g_ : 0) + (C) * 32;                                            \
    const u32x4 v_ = *reinterpret_cast<const u32x4*>(p.in_hi + o_);                          \
    const u32x4 l_ = *reinterpret_cast<const u32x4*>(p.in_lo + o_);                          \
    rh_hi[SLOT] = g_ >= 0 ? v_ : zero4;                                                      \
    rh_lo[SLOT] = g_ >= 0 ? l_ : zero4;                                                      \
  }
#define VP_STORE_H(SLOT, PC, BUF)                                                            \
  if (h_lds[PC] >= 0) {                                                                      \
    char* dst_ = halo_base + (BUF) * 2 * HALO_BYTES + h_lds[PC];                             \
    *reinterpret_cast<u32x4*>(dst_) = rh_hi[SLOT];                                           \
    *reinterpret_cast<u32x4*>(dst_ + HALO_BYTES) = rh_lo[SLOT];                              \
  }
#define VP_READ_FRAGS(SET, WBUF, HBUF, TAPOFS)                                               \
  {                                                                                          \
    const char* wsrc_ = (WBUF) + a_ofs + a_sw[SET];                                          \
    _Pragma("unroll") for (int i = 0; i < MT; ++i) {                                         \
      fa[SET][i] = *reinterpret_cast<const h8_t*>(wsrc_ + i * WCO * 32 * WROW);              \
      fal[SET][i] = *reinterpret_cast<const h8_t*>(wsrc_ + W_BYTES + i * WCO * 32 * WROW);   \
    }                                                                                        \
    _Pragma("unroll") for (int j = 0; j < NT; ++j) {                                         \
      fb[SET][j] = *reinterpret_cast<const h8_t*>((HBUF) + b_ofs[j] + (TAPOFS) + (SET) * 32); \
      fbl[SET][j] = *reinterpret_cast<const h8_t*>((HBUF) + HALO_BYTES + b_ofs[j] + (TAPOFS) + (SET) * 32); \
    }                                                                                        \
  }
#define VP_MFMA(SET)                                                                         \
  _Pragma("unroll") for (int i = 0; i < MT; ++i) _Pragma("unroll") for (int j = 0; j < NT; ++j) { \
    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fal[SET][i], fb[SET][j], acc[i][j], 0, 0, 0); \
    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[SET][i], fbl[SET][j], acc[i][j], 0, 0, 0); \
    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[SET][i], fb[SET][j], acc[i][j], 0, 0, 0);  \
  }
  // One tap step.  On entry fragment set 0 of THIS step is in flight / in registers (read during the previous step).
#define VP_TAP(T)                                                                            \
  {                                                                                          \
    constexpr int tap_ofs_ = (((T) / 3) * HWD + ((T) % 3)) * ROWB;                           \
    constexpr int tn_ = ((T) + 1) % 9;                                                       \
    constexpr int tap_next_ = ((tn_ / 3) * HWD + (tn_ % 3)) * ROWB;                          \
    const char* wcur_ = w_base + ((T) % 3) * 2 * W_BYTES;                                    \
    const char* wnext_ = w_base + (((T) + 1) % 3) * 2 * W_BYTES;                             \
    const char* hnext_ = (T) == 8 ? hbuf_other : hbuf;                                       \
    /* ---- K sub-step 0: set 1 of this step is fetched while set 0 multiplies */           \
    VP_READ_FRAGS(1, wcur_, hbuf, tap_ofs_)                                                  \
    __builtin_amdgcn_sched_barrier(0); /* keep the prefetch AHEAD of the MFMAs (the scheduler sinks it otherwise) */ \
    if constexpr ((T) < HP) VP_LOAD_H((T) % 3, (T) < HP ? (T) : 0, next_chunk ? c + 1 : c)   \
    VP_MFMA(0)                                                                               \
    /* ---- K sub-step 1: set 0 of the NEXT step is fetched while set 1 multiplies */       \
    VP_READ_FRAGS(0, wnext_, hnext_, tap_next_)                                              \
    __builtin_amdgcn_sched_barrier(0);                                                       \
    if (next_chunk || (T) < 7) VP_STORE_W(((T) + 2) % 3, ((T) + 2) % 3)                       \
    VP_LOAD_W(((T) + 2) % 3, c * 9 + (T) + 5)                                                \
    if constexpr ((T) >= 2 && (T) - 2 < HP) {                                                \
      if (next_chunk) VP_STORE_H(((T) + 1) % 3 /* == (T - 2) % 3 */, (T) >= 2 ? (T) - 2 : 0, hb ^ 1) \
    }                                                                                        \
    VP_MFMA(1)                                                                               \
    __syncthreads();                                                                         \
  }

  // ---- prologue: halo(chunk 0) and weight tiles 0, 1 -> LDS; tiles 2, 3, 4 -> ring slots 2, 0, 1
#pragma unroll
  for (int pc = 0; pc < HP; ++pc) {
    VP_LOAD_H(0, pc, 0)
    VP_STORE_H(0, pc, 0)
  }
  VP_LOAD_W(0, 0)
  VP_LOAD_W(1, 1)
  VP_STORE_W(0, 0)
  VP_STORE_W(1, 1)
  VP_LOAD_W(2, 2)
  VP_LOAD_W(0, 3)
  VP_LOAD_W(1, 4)
  __syncthreads();
  VP_READ_FRAGS(0, w_base, halo_base, 0)

  int hb = 0;
  for (int c = 0; c < KC; ++c) {
    const bool next_chunk = (c + 1 < KC);
    const char* hbuf = halo_base + hb * 2 * HALO_BYTES;
    const char* hbuf_other = halo_base + (hb ^ 1) * 2 * HALO_BYTES;
    VP_TAP(0) VP_TAP(1) VP_TAP(2) VP_TAP(3) VP_TAP(4) VP_TAP(5) VP_TAP(6) VP_TAP(7) VP_TAP(8)
    hb ^= 1;
  }
#undef VP_TAP
#undef VP_MFMA
#undef VP_READ_FRAGS
#undef VP_STORE_H
#undef VP_LOAD_H
#undef VP_STORE_W
#undef VP_LOAD_W

  // ---- register epilogue: bias + activation + (hi, lo) split, both planes staged as [pixel][CO_TILE] fp16, 16-byte stores.
  // (the loop's last barrier has passed: no wave still reads the main buffers the stage aliases; the fragment prefetch
  // issued in the last step is dead)
  constexpr int PITCH = CO_TILE * 2 + 16, STAGE_PLANE = PX * PITCH;
  static_assert(2 * STAGE_PLANE <= 4 * HALO_BYTES + 6 * W_BYTES, "stage fits the main buffers");
  const PixPatch<TW> pix{y0, x0, p.H, p.W};
#pragma unroll
  for (int i = 0; i < MT; ++i) {
    const int cl = (i * WCO + wco) * 32 + 4 * (lane >> 5);
    f32x4_t b[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) b[g] = *reinterpret_cast<const f32x4_t*>(p.bias + co0 + cl + 8 * g);
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      char* row = smem + ((wpx * NT + j) * 32 + (lane & 31)) * PITCH + cl * 2;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        h4_t h, l;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float x = apply_act(acc[i][j][4 * g + r] + b[g][r], ACT);
          h[r] = (half_t)x;
          l[r] = (half_t)(x - (float)h[r]);
        }
        *reinterpret_cast<h4_t*>(row + g * 16) = h;
        *reinterpret_cast<h4_t*>(row + STAGE_PLANE + g * 16) = l;
      }
    }
  }
  __syncthreads();
  constexpr int CPR = CO_TILE / 8, RPI = 512 / CPR;
  static_assert(512 % CPR == 0 && PX % RPI == 0, "row loop shape");
  const int c8 = tid % CPR, r0 = tid / CPR;
  const int co = co0 + c8 * 8;
  if (co >= p.Ncols) return;
#pragma unroll 4
  for (int r = r0; r < PX; r += RPI) {
    const int m = pix(r);
    if (m < 0) continue;
    const size_t o = (size_t)m * p.Cstore + co;
    *reinterpret_cast<h8_t*>(p.out_hi + o) = *reinterpret_cast<const h8_t*>(smem + r * PITCH + c8 * 16);
    *reinterpret_cast<h8_t*>(p.out_lo + o) = *reinterpret_cast<const h8_t*>(smem + STAGE_PLANE + r * PITCH + c8 * 16);
  }
}

bool conv3x3_x3w8_supported(const ConvGemmParams& p) {
  return p.ks == 3 && p.stride <= 1 && p.in_lo && p.w_lo && p.out_lo && p.nsplit == 1 && p.store_mode == STORE_NHWC &&
         p.res_mode == RES_NONE && p.post_act == ACT_NONE && (p.act == ACT_GELU || p.act == ACT_NONE) && p.CoutW % 128 == 0 &&
         p.Cin % 32 == 0 && p.Cin2 == 0;
}

hipError_t launch_conv3x3_x3w8(const ConvGemmParams& p, hipStream_t st) {
  if (!conv3x3_x3w8_supported(p)) return hipErrorInvalidValue;
  constexpr int lds = 4 * (18 * 18 * 80) + 6 * (128 * 64);
  static_assert(lds <= 160 * 1024, "LDS budget");
  auto k = p.act == ACT_GELU ? conv3x3_x3w8_kernel<128, 2, 4, ACT_GELU> : conv3x3_x3w8_kernel<128, 2, 4, ACT_NONE>;
  static LdsAttrOnce attr_once[2];
  if (hipError_t e = set_max_dynamic_lds(attr_once[p.act == ACT_GELU], reinterpret_cast<const void*>(k), lds); e != hipSuccess) return e;
  dim3 grid(((p.H + 15) / 16) * ((p.W + 15) / 16) * (p.CoutW / 128));
  hipLaunchKernelGGL(k, grid, dim3(512), lds, st, p);
  return hipGetLastError();
}

}  // namespace vp
