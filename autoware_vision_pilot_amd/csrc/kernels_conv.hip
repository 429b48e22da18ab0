// Implicit-GEMM convolution on the gfx950 matrix cores (v_mfma_f32_32x32x16_f16), im2col-free.
//
// Covers every dense conv of the hot path:
//   * 3x3 / stride 1 / pad 1 convs of context, neck and heads (scene_context.py:19-22, scene_neck.py:13-24,
//     scene_seg_head.py:13-19, ...),
//   * 1x1 convs (skip links scene_neck.py:12,17,22; EfficientNet-B0 expand / project / head convs),
//   * ConvTranspose2d(k2,s2) as a GEMM over input pixels with N = 4*Cout and a pixel-shuffle store
//     (scene_neck.py:11,16,21; scene_seg_head.py:11,16).
//
// GEMM view (swapped so that the MFMA result registers hold consecutive output channels of ONE pixel, which
// is the contiguous direction of NHWC):   D[co][px] = sum_k  Wt[co][k] * X[px][k],   k = (tap, cin).
//   MFMA A operand  = weight tile  [CO_TILE][BK]   (rows  -> output channels, result register index)
//   MFMA B operand  = pixel  tile  [PX_TILE][BK]   (cols  -> pixels, lane & 31)
// Pixels are the linear index m = y*W + x, so a tile is any PX_TILE consecutive pixels (ragged edges and
// images narrower than the tile are handled by predication, zero padding by predicated loads).
//
// Pipeline: register-staged global->LDS double buffer, one barrier per K step; LDS rows padded by 16 B
// (80 B / 144 B row pitch: conflict-free for ds_read_b128's 16-lane groups).
// fp16x3 mode (SPLIT): both operands carry a hi and a lo fp16 plane; three MFMAs per tile pair.
#include "conv_epilogue.hpp"

namespace vp {


// K1: kernel size 1 (1x1 conv / ConvTranspose-as-GEMM) -- the tap/bounds arithmetic of the general path cost ~190
// instructions per K step against 8 MFMAs; the K1 path is one 32-bit add per load.
// EPI: 0 = generic fp32-staged epilogue; 1..4 = register epilogue case (conv_epilogue.hpp regepi_case)
// W8: the weights are OCP e4m3 bytes (ConvGemmParams::w8); compile-time for the reason given at conv3x3_halo_kernel, instantiated for the 64 x 64 tile only
template <int BK, int CO_TILE, int PX_TILE, int WCO, int WPX, bool SPLIT, int DEPTH, bool K1, int EPI = 0, bool W8 = false>
__global__ __launch_bounds__(256) void conv_gemm_kernel(const ConvGemmParams p) {
  static_assert(WCO * WPX == 4, "4 waves per workgroup");
  constexpr int ROWB = BK * 2 + 16;   // LDS row pitch in bytes
  constexpr int CH = BK / 8;          // 16-byte chunks per row
  constexpr int A_CHUNKS = CO_TILE * CH, B_CHUNKS = PX_TILE * CH;
  constexpr int A_ITERS = (A_CHUNKS + 255) / 256, B_ITERS = (B_CHUNKS + 255) / 256;
  constexpr int MT = CO_TILE / WCO / 32, NT = PX_TILE / WPX / 32;
  static_assert(MT >= 1 && NT >= 1, "wave tile must be at least 32x32");
  constexpr int A_BYTES = CO_TILE * ROWB, B_BYTES = PX_TILE * ROWB;
  constexpr int STAGE = (A_BYTES + B_BYTES) * (SPLIT ? 2 : 1);
  constexpr int OFF_AHI = 0, OFF_BHI = A_BYTES, OFF_ALO = A_BYTES + B_BYTES, OFF_BLO = 2 * A_BYTES + B_BYTES;
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wco = wave / WPX, wpx = wave % WPX;
  const int M = p.H * p.W;
  // XCD-aware workgroup -> tile map (see kernels_conv3x3.hip): every XCD gets a contiguous range of virtual ids.
  // Within a range the tiles that share the LARGER operand are adjacent in time: channel tile fastest when the pixel
  // operand is the big one (up-sampling GEMMs: the 4 quadrant tiles of a pixel tile re-read it from that XCD's L2
  // instead of the Infinity Cache), pixel tile fastest when the weights are (20x40 layers with 1280-wide K).
  int vid;
  {
    const int nwg = gridDim.x, q = nwg >> 3, r = nwg & 7, xcd = blockIdx.x & 7;
    vid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (blockIdx.x >> 3);
  }
  const int n_px_tiles = (M + PX_TILE - 1) / PX_TILE, n_co_tiles = p.CoutW / CO_TILE;
  int tile_px, tile_co, zsplit;
  if (p.CoutW > M) {  // weights larger than the pixel operand: pixel tile fastest
    tile_px = vid % n_px_tiles;
    const int rest = vid / n_px_tiles;
    tile_co = rest % n_co_tiles;
    zsplit = rest / n_co_tiles;
  } else {
    tile_co = vid % n_co_tiles;
    const int rest = vid / n_co_tiles;
    tile_px = rest % n_px_tiles;
    zsplit = rest / n_px_tiles;
  }
  const int m0 = tile_px * PX_TILE, co0 = tile_co * CO_TILE;
  const int Kw = p.Cin + p.Cin2;  // weight row length; Cin2 > 0 only for the fused ConvTranspose + skip-link GEMM (ks == 1)
  const int KC = Kw / BK;
  const int S = p.ks * p.ks * KC;
  const int s_begin = (int)(((long long)S * zsplit) / p.nsplit);
  const int s_end = (int)(((long long)S * (zsplit + 1)) / p.nsplit);
  const int half_k = p.ks >> 1;
  const int cstride = p.stride > 1 ? p.stride : 1, Hin = p.stride > 1 ? p.Hin : p.H, Win = p.stride > 1 ? p.Win : p.W;

  // ---- per-thread staging assignment
  size_t a_off[A_ITERS];
  int a_lds[A_ITERS];
#pragma unroll
  for (int i = 0; i < A_ITERS; ++i) {
    const int idx = tid + 256 * i, row = idx / CH, ch = idx % CH;
    a_off[i] = (size_t)(co0 + row) * Kw + ch * 8;
    a_lds[i] = row * ROWB + ch * 16;
  }
  int b_y[B_ITERS], b_x[B_ITERS], b_lds[B_ITERS], b_ch[B_ITERS];
  int a_off32[A_ITERS], b_off32[B_ITERS];  // K1 fast path: element offsets of the K=0 chunk (-1 = row outside the image)
  int b2_off32[B_ITERS];                   // same for the K extension tensor (quadrant pixel of the 2H x 2W image)
  const int quad = p.Cin2 > 0 ? co0 / p.Cstore : 0;
#pragma unroll
  for (int i = 0; i < A_ITERS; ++i) a_off32[i] = (int)a_off[i];
#pragma unroll
  for (int i = 0; i < B_ITERS; ++i) {
    const int idx = tid + 256 * i, row = idx / CH, ch = idx % CH;
    const int m = m0 + row;
    const bool ok = (idx < B_CHUNKS) && (m < M);
    const int y = m / p.W;
    b_y[i] = ok ? y : -(1 << 20);
    b_x[i] = m - y * p.W;
    b_ch[i] = ch * 8;
    b_lds[i] = row * ROWB + ch * 16;
    b_off32[i] = ok ? m * p.Cin + ch * 8 : -1;
    b2_off32[i] = ok ? ((2 * y + (quad >> 1)) * (2 * p.W) + 2 * b_x[i] + (quad & 1)) * p.Cin2 + ch * 8 : -1;
  }

  // Staging registers: a DEPTH-deep ring of K-step tiles, indexed by COMPILE-TIME slots (the K loop is unrolled by
  // DEPTH; macros, not lambdas: closures / runtime indices push these arrays and the accumulators into scratch).
  // Each global load is issued DEPTH-1 steps before its LDS store, so the ~1-2 us first-touch latency of tensors
  // written by the previous kernel (other XCDs' L2 -> MALL/HBM) is paid once per DEPTH steps, not every step.
  typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
  const u32x4 zero4 = {0u, 0u, 0u, 0u};
  u32x4 ra_hi[DEPTH][A_ITERS], rb_hi[DEPTH][B_ITERS], ra_lo[DEPTH][SPLIT ? A_ITERS : 1], rb_lo[DEPTH][SPLIT ? B_ITERS : 1];
#pragma unroll
  for (int d = 0; d < DEPTH; ++d) {
#pragma unroll
    for (int i = 0; i < A_ITERS; ++i) {
      ra_hi[d][i] = zero4;
      if (i < (SPLIT ? A_ITERS : 1)) ra_lo[d][i] = zero4;
    }
#pragma unroll
    for (int i = 0; i < B_ITERS; ++i) {
      rb_hi[d][i] = zero4;
      if (i < (SPLIT ? B_ITERS : 1)) rb_lo[d][i] = zero4;
    }
  }
  const int s_last = s_end - 1;

#define VP_LOAD(SLOT, SIDX)                                                                                   \
  {                                                                                                           \
    const int s_ = (SIDX) < s_last ? (SIDX) : s_last; /* clamped: loads stay unconditional */                  \
    if constexpr (K1) { /* 1x1 / ConvTranspose GEMM: no tap arithmetic, 32-bit offsets, one add per load */   \
      const int c0_ = s_ * BK;                                                                                \
      _Pragma("unroll") for (int i = 0; i < A_ITERS; ++i) if (A_CHUNKS % 256 == 0 || tid + 256 * i < A_CHUNKS) { \
        if constexpr (W8) { /* fp8 storage: 8 bytes, converted at the LDS store */                            \
          const vp_u32x2 r8_ = *reinterpret_cast<const vp_u32x2*>(p.w8 + (a_off32[i] + c0_));                       \
          ra_hi[SLOT][i] = u32x4{r8_[0], r8_[1], 0u, 0u};                                                       \
        } else {                                                                                              \
          ra_hi[SLOT][i] = *reinterpret_cast<const u32x4*>(p.w_hi + (a_off32[i] + c0_));                      \
          if constexpr (SPLIT) ra_lo[SLOT][i] = *reinterpret_cast<const u32x4*>(p.w_lo + (a_off32[i] + c0_)); \
        }                                                                                                     \
      }                                                                                                       \
      const bool sec_ = c0_ >= p.Cin; /* wave-uniform: this K step reads the extension tensor */              \
      _Pragma("unroll") for (int i = 0; i < B_ITERS; ++i) {                                                   \
        const int o32_ = sec_ ? b2_off32[i] : b_off32[i];                                                     \
        const bool ok_ = o32_ >= 0;                                                                           \
        const long long k_ = sec_ ? (long long)(c0_ - p.Cin) : (long long)c0_;                                \
        const long long g_ = (long long)(ok_ ? o32_ : 0) + k_;                                                \
        const u32x4 vh_ = *reinterpret_cast<const u32x4*>(p.in_hi + (g_ + (sec_ ? p.in2_delta_hi : 0ll)));   \
        rb_hi[SLOT][i] = ok_ ? vh_ : zero4;                                                                   \
        if constexpr (SPLIT) {                                                                                \
          const u32x4 vl_ = *reinterpret_cast<const u32x4*>(p.in_lo + (g_ + (sec_ ? p.in2_delta_lo : 0ll)));  \
          rb_lo[SLOT][i] = ok_ ? vl_ : zero4;                                                                 \
        }                                                                                                     \
      }                                                                                                       \
    } else {                                                                                                  \
    const int tap_ = s_ / KC;                                                                                 \
    const int c0_ = (s_ - tap_ * KC) * BK;                                                                    \
    const int ky_ = tap_ / p.ks;                                                                              \
    const int dy_ = ky_ - half_k, dx_ = (tap_ - ky_ * p.ks) - half_k;                                         \
    const size_t wbase_ = (size_t)tap_ * p.CoutW * p.Cin + c0_;                                               \
    _Pragma("unroll") for (int i = 0; i < A_ITERS; ++i) if (A_CHUNKS % 256 == 0 || tid + 256 * i < A_CHUNKS) { \
      if constexpr (W8) {                                                                                     \
        const vp_u32x2 r8_ = *reinterpret_cast<const vp_u32x2*>(p.w8 + wbase_ + a_off[i]);                          \
        ra_hi[SLOT][i] = u32x4{r8_[0], r8_[1], 0u, 0u};                                                         \
      } else {                                                                                                \
        ra_hi[SLOT][i] = *reinterpret_cast<const u32x4*>(p.w_hi + wbase_ + a_off[i]);                         \
        if constexpr (SPLIT) ra_lo[SLOT][i] = *reinterpret_cast<const u32x4*>(p.w_lo + wbase_ + a_off[i]);    \
      }                                                                                                       \
    }                                                                                                         \
    _Pragma("unroll") for (int i = 0; i < B_ITERS; ++i) {                                                     \
      const int yy_ = b_y[i] * cstride + dy_, xx_ = b_x[i] * cstride + dx_;                                   \
      const bool ok_ = ((unsigned)yy_ < (unsigned)Hin) && ((unsigned)xx_ < (unsigned)Win);                    \
      const size_t g_ = ok_ ? ((size_t)yy_ * Win + xx_) * p.Cin + c0_ + b_ch[i] : 0; /* 0 is always valid */  \
      const u32x4 vh_ = *reinterpret_cast<const u32x4*>(p.in_hi + g_);                                        \
      rb_hi[SLOT][i] = ok_ ? vh_ : zero4;                                                                     \
      if constexpr (SPLIT) {                                                                                  \
        const u32x4 vl_ = *reinterpret_cast<const u32x4*>(p.in_lo + g_);                                      \
        rb_lo[SLOT][i] = ok_ ? vl_ : zero4;                                                                   \
      }                                                                                                       \
    }                                                                                                         \
    }                                                                                                         \
  }
#define VP_STORE(SLOT, BUF)                                                                                  \
  {                                                                                                           \
    char* st_ = smem + (BUF) * STAGE;                                                                         \
    _Pragma("unroll") for (int i = 0; i < A_ITERS; ++i) if (A_CHUNKS % 256 == 0 || tid + 256 * i < A_CHUNKS) { \
      if constexpr (W8) {                                                                                     \
        *reinterpret_cast<u32x4*>(st_ + OFF_AHI + a_lds[i]) = e4m3x8_to_half8(ra_hi[SLOT][i][0], ra_hi[SLOT][i][1]); \
        if constexpr (SPLIT) *reinterpret_cast<u32x4*>(st_ + OFF_ALO + a_lds[i]) = zero4;                     \
      } else {                                                                                                \
        *reinterpret_cast<u32x4*>(st_ + OFF_AHI + a_lds[i]) = ra_hi[SLOT][i];                                 \
        if constexpr (SPLIT) *reinterpret_cast<u32x4*>(st_ + OFF_ALO + a_lds[i]) = ra_lo[SLOT][i];            \
      }                                                                                                       \
    }                                                                                                         \
    _Pragma("unroll") for (int i = 0; i < B_ITERS; ++i) if (B_CHUNKS % 256 == 0 || tid + 256 * i < B_CHUNKS) { \
      *reinterpret_cast<u32x4*>(st_ + OFF_BHI + b_lds[i]) = rb_hi[SLOT][i];                                   \
      if constexpr (SPLIT) *reinterpret_cast<u32x4*>(st_ + OFF_BLO + b_lds[i]) = rb_lo[SLOT][i];              \
    }                                                                                                         \
  }

  f32x16_t acc[MT][NT];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

  const int frag_off = (lane & 31) * ROWB + (lane >> 5) * 16;
  const int a_frag = OFF_AHI + (wco * 32) * ROWB + frag_off;  // wave owns channel tiles i*WCO + wco (contiguous per epilogue pass)
  const int b_frag = OFF_BHI + (wpx * NT * 32) * ROWB + frag_off;

  // prologue: steps 0..DEPTH-1 in flight; step 0 -> LDS buffer 0; slot 0 refills with step DEPTH
  if (s_begin < s_end) {
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) VP_LOAD(d, s_begin + d)
    VP_STORE(0, 0)
    VP_LOAD(0, s_begin + DEPTH)
  }
  __syncthreads();
  int buf = 0;
  for (int sb = s_begin; sb < s_end; sb += DEPTH) {
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) {
      const int s = sb + d;  // (s - s_begin) % DEPTH == d
      if (s < s_end) {
        const char* st = smem + buf * STAGE;
#pragma unroll
        for (int kk = 0; kk < BK / 16; ++kk) {
          h8_t a[MT], b[NT], alo[SPLIT ? MT : 1], blo[SPLIT ? NT : 1];
#pragma unroll
          for (int i = 0; i < MT; ++i) {
            a[i] = *reinterpret_cast<const h8_t*>(st + a_frag + i * WCO * 32 * ROWB + kk * 32);
            if constexpr (SPLIT) alo[i] = *reinterpret_cast<const h8_t*>(st + (OFF_ALO - OFF_AHI) + a_frag + i * WCO * 32 * ROWB + kk * 32);
          }
#pragma unroll
          for (int j = 0; j < NT; ++j) {
            b[j] = *reinterpret_cast<const h8_t*>(st + b_frag + j * 32 * ROWB + kk * 32);
            if constexpr (SPLIT) blo[j] = *reinterpret_cast<const h8_t*>(st + (OFF_BLO - OFF_BHI) + b_frag + j * 32 * ROWB + kk * 32);
          }
#pragma unroll
          for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int j = 0; j < NT; ++j) {
              if constexpr (SPLIT) {
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(alo[i], b[j], acc[i][j], 0, 0, 0);
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[i], blo[j], acc[i][j], 0, 0, 0);
              }
              acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[i], b[j], acc[i][j], 0, 0, 0);
            }
        }
        // step s+1 (loaded DEPTH-1 steps ago) -> the other LDS buffer; its slot refills with step s+1+DEPTH
        if (s + 1 < s_end) VP_STORE((d + 1) % DEPTH, buf ^ 1)
        VP_LOAD((d + 1) % DEPTH, s + 1 + DEPTH)
        __syncthreads();
        buf ^= 1;
      }
    }
  }
#undef VP_STORE
#undef VP_LOAD

  // ---- epilogue through LDS (conv_epilogue.hpp); the main loop's last barrier has retired every LDS read
  const PixLinear pix{m0, M};
  if constexpr (EPI == 1) {
    epilogue_regs_fp16<PX_TILE, CO_TILE, WCO, MT, NT, ACT_GELU_F16, STORE_NHWC, false>(p, smem, acc, co0, wco, wpx, pix);
  } else if constexpr (EPI == 2) {
    epilogue_regs_fp16<PX_TILE, CO_TILE, WCO, MT, NT, ACT_SILU_F16, STORE_NHWC, false>(p, smem, acc, co0, wco, wpx, pix);
  } else if constexpr (EPI == 3) {
    epilogue_regs_fp16<PX_TILE, CO_TILE, WCO, MT, NT, ACT_NONE, STORE_NHWC, false>(p, smem, acc, co0, wco, wpx, pix);
  } else if constexpr (EPI == 4) {
    epilogue_regs_fp16<PX_TILE, CO_TILE, WCO, MT, NT, ACT_NONE, STORE_SHUFFLE2, false>(p, smem, acc, co0, wco, wpx, pix);
  } else {
#pragma unroll
    for (int i = 0; i < MT; ++i) epilogue_pass<PX_TILE, WCO, NT>(p, smem, acc[i], co0 + i * WCO * 32, wco, wpx, pix, M, zsplit);
  }
}

// Sums the split-K partial slabs in a fixed order (deterministic) and runs the shared epilogue.
__global__ __launch_bounds__(256) void splitk_finish_kernel(const ConvGemmParams p) {
  const int M = p.H * p.W;
  const int groups = p.Ncols >> 3;
  const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
  if (t >= (long long)M * groups) return;
  const int m = (int)(t / groups), co = (int)(t % groups) * 8;
  float v[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  for (int z = 0; z < p.nsplit; ++z) {
    const float* src = p.partial + ((size_t)z * M + m) * p.CoutW + co;
    const f32x4_t q0 = *reinterpret_cast<const f32x4_t*>(src), q1 = *reinterpret_cast<const f32x4_t*>(src + 4);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      v[r] += q0[r];
      v[4 + r] += q1[r];
    }
  }
  const f32x4_t b0 = *reinterpret_cast<const f32x4_t*>(p.bias + co), b1 = *reinterpret_cast<const f32x4_t*>(p.bias + co + 4);
  const f32x4_t s0 = *reinterpret_cast<const f32x4_t*>(p.wscale + co), s1 = *reinterpret_cast<const f32x4_t*>(p.wscale + co + 4);
  epilogue_store8(p, M, m, co, v, b0, b1, s0, s1);
}

// ------------------------------------------------------------------------------------------------ launchers
hipError_t launch_splitk_finish(const ConvGemmParams& p, hipStream_t st) {
  const long long n = (long long)p.H * p.W * (p.Ncols >> 3);
  hipLaunchKernelGGL(splitk_finish_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, p);
  return hipGetLastError();
}

template <int BK, int CO, int PX, int WCO, int WPX, bool SPLIT>
static hipError_t launch_cfg(const ConvGemmParams& p, hipStream_t st) {
  constexpr int ROWB = BK * 2 + 16;
  constexpr int lds_main = 2 * (CO + PX) * ROWB * (SPLIT ? 2 : 1);
  constexpr int lds_a = lds_main > epilogue_stage_bytes<PX, WCO>() ? lds_main : epilogue_stage_bytes<PX, WCO>();
  constexpr int lds = lds_a > epilogue_fp16_stage_bytes<PX, CO>() ? lds_a : epilogue_fp16_stage_bytes<PX, CO>();
  // prefetch depth: as deep as the register budget allows (staging = DEPTH * (A+B chunks) * 16 B per lane)
  constexpr int chunks = ((CO + PX) * (BK / 8) + 255) / 256 * (SPLIT ? 2 : 1);
  constexpr int DEPTH = chunks <= 4 ? 4 : (chunks <= 8 ? 3 : 2);
  const bool k1 = p.ks == 1;
  // register epilogue (fp16 engines, K1 GEMMs only: that is where the epilogue dominates); see regepi_case
  const int epi = (k1 && !SPLIT && p.w8 == nullptr) ? regepi_case(p, CO, SPLIT) : 0;
  void (*k)(const ConvGemmParams) = k1 ? conv_gemm_kernel<BK, CO, PX, WCO, WPX, SPLIT, DEPTH, true, 0> : conv_gemm_kernel<BK, CO, PX, WCO, WPX, SPLIT, DEPTH, false, 0>;
  if (p.w8 != nullptr) {   // fp8 weight storage: the 64 x 64 tile only (the engine selects nothing else for such a layer)
    if constexpr (CO == 64 && PX == 64) k = k1 ? conv_gemm_kernel<BK, CO, PX, WCO, WPX, SPLIT, DEPTH, true, 0, true> : conv_gemm_kernel<BK, CO, PX, WCO, WPX, SPLIT, DEPTH, false, 0, true>;
    else return hipErrorInvalidValue;
  }
  if constexpr (!SPLIT) {
    if (epi == 1) k = conv_gemm_kernel<BK, CO, PX, WCO, WPX, false, DEPTH, true, 1>;
    if (epi == 2) k = conv_gemm_kernel<BK, CO, PX, WCO, WPX, false, DEPTH, true, 2>;
    if (epi == 3) k = conv_gemm_kernel<BK, CO, PX, WCO, WPX, false, DEPTH, true, 3>;
    if (epi == 4) k = conv_gemm_kernel<BK, CO, PX, WCO, WPX, false, DEPTH, true, 4>;
  }
  static LdsAttrOnce attr_once[2][6];
  if (hipError_t e = set_max_dynamic_lds(attr_once[k1][p.w8 ? 5 : epi], reinterpret_cast<const void*>(k), lds); e != hipSuccess) return e;
  const int M = p.H * p.W;
  dim3 grid(((M + PX - 1) / PX) * (p.CoutW / CO) * p.nsplit);  // decoded in the kernel (XCD-aware)
  hipLaunchKernelGGL(k, grid, dim3(256), lds, st, p);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return e;
  if (p.nsplit > 1) e = launch_splitk_finish(p, st);
  return e;
}

// tile ids: 0 = 128co x 128px, 1 = 64co x 128px, 2 = 64co x 64px, 3 = 32co x 128px
hipError_t launch_conv_gemm(const ConvGemmParams& p, int tile, int bk, bool split, hipStream_t st) {
#define VP_CASE(T, CO, PX, WCO, WPX)                                                          \
  if (tile == T) {                                                                            \
    if (bk == 64) return split ? launch_cfg<64, CO, PX, WCO, WPX, true>(p, st) : launch_cfg<64, CO, PX, WCO, WPX, false>(p, st); \
    return split ? launch_cfg<32, CO, PX, WCO, WPX, true>(p, st) : launch_cfg<32, CO, PX, WCO, WPX, false>(p, st);   \
  }
  VP_CASE(0, 128, 128, 2, 2)
  VP_CASE(1, 64, 128, 2, 2)
  VP_CASE(2, 64, 64, 2, 2)
  VP_CASE(3, 32, 128, 1, 4)
#undef VP_CASE
  return hipErrorInvalidValue;
}

int conv_tile_co(int tile) { return tile == 0 ? 128 : (tile == 3 ? 32 : 64); }
int conv_tile_px(int tile) { return tile == 2 ? 64 : 128; }

}  // namespace vp
