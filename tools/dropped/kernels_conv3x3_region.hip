// 3x3 / stride 1 / pad 1 convolution for SMALL maps with LONG K: the neck layers at 20x40 and 40x80 (K = 4.6k..11.5k,
// M = 800 / 3200 pixels; scene_neck.py:13-19) and the AutoDrive head at 16x32.
//
// The patch kernel (kernels_conv3x3.hip) gives a workgroup 128 pixels x 128 channels; on these layers that means
// 9-25 pixel tiles re-streaming every weight tile (159 MB of weight traffic for 17.7 MB of weights on the first neck
// layer), 44 % of the MFMA work spent on tile padding at 20x40, a barrier per tap and one or two workgroups per CU
// fully exposed to latency: 230-490 TFLOP/s where the large layers reach 750+.
//
// Here a workgroup owns a REGION of RH x RW pixels (10x40 or 16x32: <= 512 pixels = 16 MFMA column fragments over LINEAR
// pixel indices, so a 20x40 map is two regions with no padding rows or columns), a slice of 32 output channels and a
// K slice (split-K over 32-channel chunks).  Per chunk the whole (RH+2) x (RW+2) halo AND all nine 32x32 weight tiles
// sit in LDS, so there is ONE barrier per chunk (9 taps x 2 k-steps x 4 MFMAs per wave between barriers); the next
// chunk's halo and weights are fetched into registers at the top of the chunk (one workgroup per CU = one wave per
// SIMD, the register file is all ours) and written to the other LDS buffers at its end.  Every weight element is
// read by (number of regions) workgroups instead of (number of pixel tiles): 2 / 8 instead of 9 / 25.
#include "conv_epilogue.hpp"

namespace vp {

// local pixel index of the region -> linear pixel of the image, -1 outside
template <int RW>
struct PixRegion {
  int y0, x0, H, W, npx;
  __device__ __forceinline__ int operator()(int q) const {
    const int ly = q / RW, lx = q - ly * RW;
    const int y = y0 + ly, x = x0 + lx;
    return (q < npx && y < H && x < W) ? y * W + x : -1;
  }
};

template <int RH, int RW, int CO, bool SPLIT>
__global__ __launch_bounds__(256) void conv3x3_region_kernel(const ConvGemmParams p) {
  constexpr int MT = CO / 32, NT = 4, PXCAP = 4 * NT * 32;  // 512 pixel slots; all four waves share the CO channels
  static_assert(RH * RW <= PXCAP, "region must fit 16 column fragments");
  constexpr int ROWB = 80, HW2 = RW + 2, HPX = (RH + 2) * HW2;
  constexpr int HPIECES = HPX * 4, HP = (HPIECES + 255) / 256;
  constexpr int WPIECES = 9 * CO * 4, WP = (WPIECES + 255) / 256;
  constexpr int NPL = SPLIT ? 2 : 1;
  constexpr int HALO_BYTES = HPX * ROWB, W_BYTES = 9 * CO * 64;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* const halo_base = smem;                           // [2][NPL][HALO_BYTES]
  char* const w_base = smem + 2 * NPL * HALO_BYTES;       // [2][NPL][W_BYTES]

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  // XCD-aware decode (see kernels_conv3x3.hip): regions of one (channel tile, K slice) share its weights -> fastest index
  int vid;
  {
    const int nwg = gridDim.x, q = nwg >> 3, r = nwg & 7, xcd = blockIdx.x & 7;
    vid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (blockIdx.x >> 3);
  }
  const int regs_x = (p.W + RW - 1) / RW, n_regs = regs_x * ((p.H + RH - 1) / RH), n_co = p.CoutW / CO;
  const int reg = vid % n_regs, rest = vid / n_regs;
  const int tile_co = rest % n_co, zsplit = rest / n_co;
  const int ry = reg / regs_x, rx = reg - ry * regs_x;
  const int y0 = ry * RH, x0 = rx * RW, co0 = tile_co * CO;
  const int KC = p.Cin >> 5;
  const int c_begin = (int)(((long long)KC * zsplit) / p.nsplit), c_end = (int)(((long long)KC * (zsplit + 1)) / p.nsplit);
  const int M = p.H * p.W;

  // ---- staging assignment
  typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
  const u32x4 zero4 = {0u, 0u, 0u, 0u};
  int h_goff[HP], h_lds[HP];
#pragma unroll
  for (int i = 0; i < HP; ++i) {
    const int idx = tid + 256 * i, hp = idx >> 2, ch = idx & 3;
    const int hy = hp / HW2, hx = hp - hy * HW2;
    const int gy = y0 - 1 + hy, gx = x0 - 1 + hx;
    const bool in_halo = idx < HPIECES;
    const bool ok = in_halo && (unsigned)gy < (unsigned)p.H && (unsigned)gx < (unsigned)p.W;
    h_goff[i] = ok ? (gy * p.W + gx) * p.Cin + ch * 8 : -1;
    h_lds[i] = in_halo ? hp * ROWB + ch * 16 : -1;
  }
  int w_goff[WP], w_lds[WP];
#pragma unroll
  for (int i = 0; i < WP; ++i) {
    const int idx = tid + 256 * i, tap = idx / (CO * 4), rem = idx - tap * (CO * 4), row = rem >> 2, ch = rem & 3;
    const bool ok = idx < WPIECES;
    // packed weights [chunk][tap][CoutW][32]: element offset of (tap, co0 + row, ch*8) inside one chunk
    w_goff[i] = ok ? (tap * p.CoutW + co0 + row) * 32 + ch * 8 : -1;
    w_lds[i] = ok ? (tap * CO + row) * 64 + ((ch ^ ((row >> 2) & 3)) << 4) : -1;
  }
  const size_t w_chunk = (size_t)9 * p.CoutW * 32;  // elements per chunk

  // ---- fragment addressing: this wave's NT column fragments over linear region pixels
  int b_ofs[NT];
#pragma unroll
  for (int j = 0; j < NT; ++j) {
    int q = (wave * NT + j) * 32 + (lane & 31);
    if (q >= RH * RW) q = 0;  // padding slots read a valid pixel; their results are dropped by the pixel map
    const int ly = q / RW, lx = q - ly * RW;
    b_ofs[j] = (ly * HW2 + lx) * ROWB + (lane >> 5) * 16;
  }
  const int a_row = lane & 31, a_swz = (a_row >> 2) & 3;
  const int a_ofs = a_row * 64;
  const int a_sw[2] = {((lane >> 5) ^ a_swz) << 4, ((2 + (lane >> 5)) ^ a_swz) << 4};

  f32x16_t acc[MT][NT];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

  u32x4 rh_hi[HP], rh_lo[SPLIT ? HP : 1], rw_hi[WP], rw_lo[SPLIT ? WP : 1];
#define VP_FETCH(C)                                                                                    \
  {                                                                                                   \
    _Pragma("unroll") for (int i = 0; i < HP; ++i) {                                                   \
      const int g_ = h_goff[i];                                                                        \
      const int o_ = (g_ >= 0 ? g_ : 0) + (C) * 32; /* clamped: the load stays unconditional */        \
      rh_hi[i] = *reinterpret_cast<const u32x4*>(p.in_hi + o_); /* zero padding is applied at commit time: */ \
      if constexpr (SPLIT) rh_lo[i] = *reinterpret_cast<const u32x4*>(p.in_lo + o_); /* a select here waits */ \
    }                                                                                                  \
    const size_t wb_ = (size_t)(C) * w_chunk;                                                          \
    _Pragma("unroll") for (int i = 0; i < WP; ++i) {                                                   \
      const int g_ = w_goff[i] >= 0 ? w_goff[i] : 0;                                                   \
      rw_hi[i] = *reinterpret_cast<const u32x4*>(p.w_hi + wb_ + g_);                                   \
      if constexpr (SPLIT) rw_lo[i] = *reinterpret_cast<const u32x4*>(p.w_lo + wb_ + g_);              \
    }                                                                                                  \
  }
#define VP_COMMIT(BUF)                                                                                 \
  {                                                                                                   \
    char* hd_ = halo_base + (BUF) * NPL * HALO_BYTES;                                                  \
    char* wd_ = w_base + (BUF) * NPL * W_BYTES;                                                        \
    _Pragma("unroll") for (int i = 0; i < HP; ++i) if (h_lds[i] >= 0) {                                \
      *reinterpret_cast<u32x4*>(hd_ + h_lds[i]) = h_goff[i] >= 0 ? rh_hi[i] : zero4;                   \
      if constexpr (SPLIT) *reinterpret_cast<u32x4*>(hd_ + HALO_BYTES + h_lds[i]) = h_goff[i] >= 0 ? rh_lo[i] : zero4; \
    }                                                                                                  \
    _Pragma("unroll") for (int i = 0; i < WP; ++i) if (w_lds[i] >= 0) {                                \
      *reinterpret_cast<u32x4*>(wd_ + w_lds[i]) = rw_hi[i];                                            \
      if constexpr (SPLIT) *reinterpret_cast<u32x4*>(wd_ + W_BYTES + w_lds[i]) = rw_lo[i];             \
    }                                                                                                  \
  }

  if (c_begin < c_end) {
    VP_FETCH(c_begin)
    VP_COMMIT(0)
  }
  __syncthreads();
  int buf = 0;
  for (int c = c_begin; c < c_end; ++c) {
    const bool more = c + 1 < c_end;
    if (more) VP_FETCH(c + 1)
    const char* hbuf = halo_base + buf * NPL * HALO_BYTES;
    const char* wbuf = w_base + buf * NPL * W_BYTES;
    // 18 (tap, k-half) groups of 1 weight + NT pixel fragments and NT MFMAs.  The fragments of group g+1 are read while
    // the MFMAs of group g run (two register sets, compile-time alternation; sched_barriers keep the order: left alone
    // the scheduler emits read-wait-MFMA per group and exposes the full LDS latency 18 times per chunk).
    h8_t fa0[MT], fa1[MT], fb0[NT], fb1[NT], fal0[SPLIT ? MT : 1], fal1[SPLIT ? MT : 1], fbl0[SPLIT ? NT : 1], fbl1[SPLIT ? NT : 1];
#define VP_RG_READ(FA, FB, FAL, FBL, G)                                                                \
  {                                                                                                   \
    constexpr int tap_ = (G) / 2, kk_ = (G) % 2;                                                       \
    constexpr int tap_ofs_ = ((tap_ / 3) * HW2 + (tap_ % 3)) * ROWB;                                   \
    _Pragma("unroll") for (int i = 0; i < MT; ++i) {                                                   \
      FA[i] = *reinterpret_cast<const h8_t*>(wbuf + (tap_ * CO + i * 32) * 64 + a_ofs + a_sw[kk_]);    \
      if constexpr (SPLIT) FAL[i] = *reinterpret_cast<const h8_t*>(wbuf + W_BYTES + (tap_ * CO + i * 32) * 64 + a_ofs + a_sw[kk_]); \
    }                                                                                                  \
    _Pragma("unroll") for (int j = 0; j < NT; ++j) {                                                   \
      FB[j] = *reinterpret_cast<const h8_t*>(hbuf + b_ofs[j] + tap_ofs_ + kk_ * 32);                   \
      if constexpr (SPLIT) FBL[j] = *reinterpret_cast<const h8_t*>(hbuf + HALO_BYTES + b_ofs[j] + tap_ofs_ + kk_ * 32); \
    }                                                                                                  \
  }
#define VP_RG_MMA(FA, FB, FAL, FBL)                                                                    \
  _Pragma("unroll") for (int i = 0; i < MT; ++i) _Pragma("unroll") for (int j = 0; j < NT; ++j) {      \
    if constexpr (SPLIT) {                                                                             \
      acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(FAL[i], FB[j], acc[i][j], 0, 0, 0);           \
      acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(FA[i], FBL[j], acc[i][j], 0, 0, 0);           \
    }                                                                                                  \
    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(FA[i], FB[j], acc[i][j], 0, 0, 0);              \
  }
#define VP_RG_PAIR(G) /* groups G (set 0) and G+1 (set 1) */                                           \
  VP_RG_READ(fa1, fb1, fal1, fbl1, (G) + 1)                                                            \
  __builtin_amdgcn_sched_barrier(0);                                                                   \
  VP_RG_MMA(fa0, fb0, fal0, fbl0)                                                                      \
  __builtin_amdgcn_sched_barrier(0);                                                                   \
  if constexpr ((G) + 2 < 18) VP_RG_READ(fa0, fb0, fal0, fbl0, (G) + 2)                                \
  __builtin_amdgcn_sched_barrier(0);                                                                   \
  VP_RG_MMA(fa1, fb1, fal1, fbl1)                                                                      \
  __builtin_amdgcn_sched_barrier(0);
    VP_RG_READ(fa0, fb0, fal0, fbl0, 0)
    VP_RG_PAIR(0) VP_RG_PAIR(2) VP_RG_PAIR(4) VP_RG_PAIR(6) VP_RG_PAIR(8) VP_RG_PAIR(10) VP_RG_PAIR(12) VP_RG_PAIR(14) VP_RG_PAIR(16)
#undef VP_RG_PAIR
#undef VP_RG_MMA
#undef VP_RG_READ
    if (more) VP_COMMIT(buf ^ 1)
    __syncthreads();
    buf ^= 1;
  }
#undef VP_FETCH
#undef VP_COMMIT

  // ---- epilogue: the shared fp32-staged pass (bias / activation / residual / split-K partials), 32 channels x PXCAP slots
  const PixRegion<RW> pix{y0, x0, p.H, p.W, RH * RW};
#pragma unroll
  for (int i = 0; i < MT; ++i) epilogue_pass<PXCAP, 1, NT>(p, smem, acc[i], co0 + i * 32, 0, wave, pix, M, zsplit);
}

template <int RH, int RW, int CO, bool SPLIT>
static hipError_t launch_region_cfg(const ConvGemmParams& p, hipStream_t st) {
  constexpr int lds_main = 2 * (SPLIT ? 2 : 1) * ((RH + 2) * (RW + 2) * 80 + 9 * CO * 64);
  constexpr int lds = lds_main > epilogue_stage_bytes<512, 1>() ? lds_main : epilogue_stage_bytes<512, 1>();
  static_assert(lds <= 160 * 1024, "LDS budget");
  auto k = conv3x3_region_kernel<RH, RW, CO, SPLIT>;
  static LdsAttrOnce attr_once;
  if (hipError_t e = set_max_dynamic_lds(attr_once, reinterpret_cast<const void*>(k), lds); e != hipSuccess) return e;
  const int n_regs = ((p.H + RH - 1) / RH) * ((p.W + RW - 1) / RW);
  dim3 grid(n_regs * (p.CoutW / CO) * p.nsplit);
  hipLaunchKernelGGL(k, grid, dim3(256), lds, st, p);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return e;
  if (p.nsplit > 1) e = launch_splitk_finish(p, st);
  return e;
}

// region shapes: 0 = 10 x 40 (20x40 and 40x80 maps), 1 = 16 x 32 (AutoDrive P5 maps)
bool region_shape_fits(int shape, int H, int W) {
  const int rh = shape == 0 ? 10 : 16, rw = shape == 0 ? 40 : 32;
  return H % rh == 0 && W % rw == 0;
}
int region_co(int shape, int CoutW) { return (shape == 0 && CoutW % 64 == 0) ? 64 : 32; }
int region_count(int shape, int H, int W) {
  const int rh = shape == 0 ? 10 : 16, rw = shape == 0 ? 40 : 32;
  return ((H + rh - 1) / rh) * ((W + rw - 1) / rw);
}
hipError_t launch_conv3x3_region(const ConvGemmParams& p, int shape, bool split, hipStream_t st) {
  if (split) return hipErrorInvalidValue;  // the (hi, lo) planes of halo + 9 weight tiles, double-buffered, exceed 160 KiB
  // 64 channels per workgroup where the LDS plan allows (10x40: 154 KiB) and the layer has them: half the fragment
  // reads and half the halo staging per MFMA
  if (shape == 0) return p.CoutW % 64 == 0 ? launch_region_cfg<10, 40, 64, false>(p, st) : launch_region_cfg<10, 40, 32, false>(p, st);
  if (shape == 1) return launch_region_cfg<16, 32, 32, false>(p, st);
  return hipErrorInvalidValue;
}

}  // namespace vp
