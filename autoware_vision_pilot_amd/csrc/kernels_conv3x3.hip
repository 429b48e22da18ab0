// 3x3 / stride 1 / pad 1 convolution, second generation: input halo tile resident in LDS.
//
// The decoder's 3x3 convs are ~96 % of the frame's FLOPs.  The generic implicit-GEMM kernel
// (kernels_conv.hip) re-stages the pixel operand from global memory for every tap, which makes it L1/LDS-write
// bound (~160 TFLOP/s measured, profiles/r01_*).  Here a workgroup owns a TH x TW patch of output pixels and a
// CO_TILE slice of output channels and, per 32-channel input chunk,
//   * stages the (TH+2) x (TW+2) x 32 input halo ONCE into LDS and reuses it for all 9 taps: the MFMA pixel
//     operand of tap (ky,kx) is the same LDS image read at a shifted per-lane address (im2col never exists);
//   * streams one [CO_TILE][32] weight tile per tap from a tap-major repacked weight tensor
//     [cin/32][9][CoutW][32], so every tile is one contiguous, fully coalesced 2-8 KiB block;
//   * overlaps the next tap's weight tile and one sixth of the next chunk's halo (global -> registers) with the
//     current tap's MFMAs, writes them to the other LDS buffers afterwards: one barrier per tap.
// Global->LDS traffic per FLOP drops ~6x vs the generic kernel; LDS rows keep the 80-byte pitch
// (conflict-free ds_read_b128).  Result layout, epilogue and split-K protocol are shared with kernels_conv.hip.
#include <cstdlib>
#include <type_traits>

#include "conv_epilogue.hpp"

namespace vp {

// ABL: ablation bits for tools/halo_ablate.hip only (1 = no global loads / LDS stores in the loop, 2 = no MFMA,
// 4 = no LDS fragment reads, 8 = no barrier, 16 / 32 = de-phase the workgroups sharing a CU by TG_ID parity / block index
// with `p.ks - 3` x s_sleep(127), the tool passes the count in ks -- DESIGN.md "Tried and dropped"); always 0 in the library.
// FASTEPI: single-pass register GELU + fp16-staged epilogue (conv_epilogue.hpp epilogue_regs_fp16); the launcher
// selects it when the layer is bias + ACT_GELU_F16 (VP_FP16 engines), no residual, NHWC, no split-K.
// W8: the weights are OCP e4m3 BYTES (ConvGemmParams::w8, VP_WEIGHTS_FP8 as storage): 8 bytes per piece instead of 16, converted to fp16 on their way to
// LDS.  A COMPILE-TIME flag: as a run-time branch in the tap macros it cost the fp16 engines' big layers 12-47 % (the compiler waits at the joins;
// measured, round 4) -- instantiated for the tiles an fp8 engine uses (64- and 32-channel 8x16 tiles), the library's other kernels are untouched.
template <int CO_TILE, int TH, int TW, int WCO, int WPX, bool SPLIT, int ABL = 0, bool FASTEPI = false, bool W8 = false>
__global__ __launch_bounds__(256) void conv3x3_halo_kernel(const ConvGemmParams p) {
  static_assert(WCO * WPX == 4, "4 waves");
  constexpr int ROWB = 80, CH = 4;  // 32 channels = 64 B + 16 B pad
  constexpr int PX = TH * TW, HWD = TW + 2, HHT = TH + 2, HPX = HHT * HWD;
  constexpr int HCHUNKS = HPX * CH, HP = (HCHUNKS + 255) / 256;
  static_assert(HP <= 7, "halo pieces are loaded at taps 0..HP-1 and stored two taps later");
  constexpr int WCHUNKS = CO_TILE * CH, WP = (WCHUNKS + 255) / 256;
  constexpr int MT = CO_TILE / WCO / 32, NT = PX / WPX / 32;
  static_assert(MT >= 1 && NT >= 1, "wave tile");
  constexpr int NPL = SPLIT ? 2 : 1;
  constexpr int WROW = 64;  // weight tile rows are unpadded; 16-byte chunks XOR-swizzled by (row>>2)&3: conflict-free ds_read_b128 AND ds_write_b128
  constexpr int HALO_BYTES = HPX * ROWB, W_BYTES = CO_TILE * WROW;
  static_assert(TW == 16, "lane->pixel map assumes 16-pixel-wide patches");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* const halo_base = smem;                             // [2][NPL][HALO_BYTES]
  char* const w_base = smem + 2 * NPL * HALO_BYTES;         // [2][NPL][W_BYTES]

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wco = wave / WPX, wpx = wave % WPX;
  const int tiles_x = (p.W + TW - 1) / TW;
  // XCD-aware workgroup -> tile map.  The dispatcher places workgroup b on XCD b % 8 (8 private 4 MiB L2s); the
  // bijective remap below hands every XCD one CONTIGUOUS range of virtual ids, decoded pixel tile fastest, then
  // output-channel tile, then split-K slice.  Workgroups that share a weight slice (same channel tile and K slice,
  // different pixel tiles) and neighbouring pixel tiles (shared halos) then hit in the same L2 instead of every XCD
  // streaming its own copy from the Infinity Cache (measured on the 20x40 neck layer: 159 MB of weight traffic for
  // 17.7 MB of weights, ~5.5 TB/s, the layer's bound).
  int vid;
  {
    const int nwg = gridDim.x, q = nwg >> 3, r = nwg & 7, xcd = blockIdx.x & 7;
    vid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (blockIdx.x >> 3);
  }
  const int n_px_tiles = tiles_x * ((p.H + TH - 1) / TH), n_co_tiles = p.CoutW / CO_TILE;
  // nsplit == 1: the output-channel tile is the fastest index -- the channel tiles of one pixel tile run side by side on one XCD and share its
  // input patch in L2 (round 4, measured on the pipelined kernels: kernels_conv3x3_x3.hip); split layers keep pixel tiles fastest inside a K slice
  const bool co_fast = p.nsplit == 1;
  const int tile_px = co_fast ? vid / n_co_tiles : vid % n_px_tiles, tile_rest = vid / n_px_tiles;
  const int tile_co = co_fast ? vid % n_co_tiles : tile_rest % n_co_tiles, zsplit = co_fast ? 0 : tile_rest / n_co_tiles;
  const int tyi = tile_px / tiles_x, txi = tile_px - tyi * tiles_x;
  const int y0 = tyi * TH, x0 = txi * TW;
  const int co0 = tile_co * CO_TILE;
  const int KC = p.Cin >> 5;
  const int c_begin = (int)(((long long)KC * zsplit) / p.nsplit);
  const int c_end = (int)(((long long)KC * (zsplit + 1)) / p.nsplit);
  const int M = p.H * p.W;

  // ---- staging assignment (compile-time indexed after unrolling)
  int h_goff[HP], h_lds[HP];
#pragma unroll
  for (int pc = 0; pc < HP; ++pc) {
    const int hidx = tid + 256 * pc;
    const int hp = hidx >> 2, ch = hidx & 3;
    const int hy = hp / HWD, hx = hp - hy * HWD;
    const int gy = y0 - 1 + hy, gx = x0 - 1 + hx;
    const bool in_tile = hidx < HCHUNKS;
    const bool ok = in_tile && (unsigned)gy < (unsigned)p.H && (unsigned)gx < (unsigned)p.W;
    h_goff[pc] = ok ? (gy * p.W + gx) * p.Cin + ch * 8 : -1;
    h_lds[pc] = in_tile ? hp * ROWB + ch * 16 : -1;
  }
  int w_goff[WP], w_lds[WP];
#pragma unroll
  for (int pc = 0; pc < WP; ++pc) {
    const int idx = tid + 256 * pc;
    const int row = idx >> 2, ch = idx & 3;
    const bool ok = idx < WCHUNKS;
    w_goff[pc] = ok ? (co0 + row) * 32 + ch * 8 : -1;
    w_lds[pc] = row * WROW + ((ch ^ ((row >> 2) & 3)) << 4);
  }
  const size_t w_step = (size_t)p.CoutW * 32;  // elements per (chunk, tap) weight tile row-set
  typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
  const u32x4 zero4 = {0u, 0u, 0u, 0u};

  // ---- fragment addressing
  int b_ofs[NT];
#pragma unroll
  for (int j = 0; j < NT; ++j) {
    const int q = (wpx * NT + j) * 32 + (lane & 31);
    int rowbit, px;
    lane_to_px16(q & 31, rowbit, px);  // conflict-free lane -> pixel map (conv_epilogue.hpp)
    b_ofs[j] = ((2 * (q >> 5) + rowbit) * HWD + px) * ROWB + (lane >> 5) * 16;
  }
  const int a_ofs = (wco * 32 + (lane & 31)) * WROW;  // wave owns channel tiles i*WCO + wco
  const int a_swz = ((lane & 31) >> 2) & 3;
  const int a_sw[2] = {(((lane >> 5)) ^ a_swz) << 4, ((2 + (lane >> 5)) ^ a_swz) << 4};

  f32x16_t acc[MT][NT];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

  // Staging registers: a 3-deep ring of per-tap weight tiles and a 3-deep ring of halo pieces, indexed by
  // COMPILE-TIME tap numbers (macros, not lambdas or runtime loops: closures / dynamic indices push these arrays and
  // the accumulators into scratch memory).  Ring depth 3 divides the 9 taps, so slot = tap % 3 everywhere.
  // Every load is issued 2-3 tap steps before its LDS store: HBM/MALL latency (~1-2 us for first-touch weights) is
  // hidden behind ~3 taps of MFMAs instead of one.
  u32x4 rw_hi[3][WP], rw_lo[3][SPLIT ? WP : 1], rh_hi[3], rh_lo[3];
#pragma unroll
  for (int r = 0; r < 3; ++r) {
    rh_hi[r] = zero4;
    rh_lo[r] = zero4;
#pragma unroll
    for (int pc = 0; pc < WP; ++pc) {
      rw_hi[r][pc] = zero4;
      if (pc < (SPLIT ? WP : 1)) rw_lo[r][pc] = zero4;
    }
  }
  const int s_last = c_end * 9 - 1;
  h8_t afix;  // ablation only
#pragma unroll
  for (int e = 0; e < 8; ++e) afix[e] = (half_t)(0.001f * (float)(lane + e));

#define VP_LOAD_W(SLOT, SIDX)                                                                         \
  {                                                                                                   \
    const int si_ = (SIDX) < s_last ? (SIDX) : s_last; /* clamped: loads stay unconditional */         \
    const size_t base_ = (size_t)si_ * w_step;                                                        \
    _Pragma("unroll") for (int pc = 0; pc < WP; ++pc) if (WCHUNKS % 256 == 0 || w_goff[pc] >= 0) {    \
      if constexpr (W8) { /* fp8 storage: 8 bytes instead of 16, converted at the LDS store */        \
        const vp_u32x2 r8_ = *reinterpret_cast<const vp_u32x2*>(p.w8 + base_ + w_goff[pc]);                 \
        rw_hi[SLOT][pc] = u32x4{r8_[0], r8_[1], 0u, 0u};                                                \
      } else {                                                                                        \
        rw_hi[SLOT][pc] = *reinterpret_cast<const u32x4*>(p.w_hi + base_ + w_goff[pc]);               \
        if constexpr (SPLIT) rw_lo[SLOT][pc] = *reinterpret_cast<const u32x4*>(p.w_lo + base_ + w_goff[pc]); \
      }                                                                                               \
    }                                                                                                 \
  }
#define VP_STORE_W(SLOT, BUF)                                                                         \
  {                                                                                                   \
    char* dst_ = w_base + (BUF) * NPL * W_BYTES;                                                      \
    _Pragma("unroll") for (int pc = 0; pc < WP; ++pc) if (WCHUNKS % 256 == 0 || w_goff[pc] >= 0) {    \
      if constexpr (W8) {                                                                             \
        *reinterpret_cast<u32x4*>(dst_ + w_lds[pc]) = e4m3x8_to_half8(rw_hi[SLOT][pc][0], rw_hi[SLOT][pc][1]); \
        if constexpr (SPLIT) *reinterpret_cast<u32x4*>(dst_ + W_BYTES + w_lds[pc]) = zero4;           \
      } else {                                                                                        \
        *reinterpret_cast<u32x4*>(dst_ + w_lds[pc]) = rw_hi[SLOT][pc];                                \
        if constexpr (SPLIT) *reinterpret_cast<u32x4*>(dst_ + W_BYTES + w_lds[pc]) = rw_lo[SLOT][pc]; \
      }                                                                                               \
    }                                                                                                 \
  }
  // out-of-image halo pixels: load from offset 0 (always valid) and zero the value, so the access stays a plain
  // global_load (a pointer select turned it into flat_load)
#define VP_LOAD_H(SLOT, PC, C)                                                                        \
  {                                                                                                   \
    const int g_ = h_goff[PC];                                                                        \
    const int o_ = (g_ >= 0 ? g_ : 0) + (C) * 32;                                                     \
    u32x4 v_ = *reinterpret_cast<const u32x4*>(p.in_hi + o_);                                         \
    rh_hi[SLOT] = g_ >= 0 ? v_ : zero4;                                                               \
    if constexpr (SPLIT) {                                                                            \
      u32x4 l_ = *reinterpret_cast<const u32x4*>(p.in_lo + o_);                                       \
      rh_lo[SLOT] = g_ >= 0 ? l_ : zero4;                                                             \
    }                                                                                                 \
  }
#define VP_STORE_H(SLOT, PC, BUF)                                                                     \
  if (h_lds[PC] >= 0) {                                                                               \
    char* dst_ = halo_base + (BUF) * NPL * HALO_BYTES;                                                \
    *reinterpret_cast<u32x4*>(dst_ + h_lds[PC]) = rh_hi[SLOT];                                        \
    if constexpr (SPLIT) *reinterpret_cast<u32x4*>(dst_ + HALO_BYTES + h_lds[PC]) = rh_lo[SLOT];      \
  }
#define VP_TAP(T)                                                                                     \
  {                                                                                                   \
    /* halo piece T of the NEXT chunk: issued now, stored two taps later */                           \
    if constexpr ((T) < HP && !(ABL & 1)) VP_LOAD_H((T) % 3, (T) < HP ? (T) : 0, next_chunk ? c + 1 : c) \
    constexpr int tap_ofs_ = (((T) / 3) * HWD + ((T) % 3)) * ROWB;                                    \
    const char* wbuf_ = w_base + wb * NPL * W_BYTES;                                                  \
    _Pragma("unroll") for (int kk = 0; kk < 2; ++kk) {                                                \
      h8_t a_[MT], b_[NT], alo_[SPLIT ? MT : 1], blo_[SPLIT ? NT : 1];                                \
      _Pragma("unroll") for (int i = 0; i < MT; ++i) {                                                \
        if constexpr (ABL & 4) { a_[i] = afix; if constexpr (SPLIT) alo_[i] = afix; continue; }       \
        a_[i] = *reinterpret_cast<const h8_t*>(wbuf_ + a_ofs + i * WCO * 32 * WROW + a_sw[kk]);        \
        if constexpr (SPLIT) alo_[i] = *reinterpret_cast<const h8_t*>(wbuf_ + W_BYTES + a_ofs + i * WCO * 32 * WROW + a_sw[kk]); \
      }                                                                                               \
      _Pragma("unroll") for (int j = 0; j < NT; ++j) {                                                \
        if constexpr (ABL & 4) { b_[j] = afix; if constexpr (SPLIT) blo_[j] = afix; continue; }       \
        b_[j] = *reinterpret_cast<const h8_t*>(hbuf + b_ofs[j] + tap_ofs_ + kk * 32);                 \
        if constexpr (SPLIT) blo_[j] = *reinterpret_cast<const h8_t*>(hbuf + HALO_BYTES + b_ofs[j] + tap_ofs_ + kk * 32); \
      }                                                                                               \
      _Pragma("unroll") for (int i = 0; i < MT; ++i) _Pragma("unroll") for (int j = 0; j < NT; ++j) { \
        if constexpr (ABL & 2) { asm volatile("" ::"v"(a_[i]), "v"(b_[j])); continue; }               \
        if constexpr (SPLIT) {                                                                        \
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(alo_[i], b_[j], acc[i][j], 0, 0, 0);     \
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a_[i], blo_[j], acc[i][j], 0, 0, 0);     \
        }                                                                                             \
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a_[i], b_[j], acc[i][j], 0, 0, 0);         \
      }                                                                                               \
    }                                                                                                 \
    /* weight tile of step s+1 (loaded 3 steps ago) -> the other LDS buffer; its slot then refills with step s+4 */ \
    if constexpr (!(ABL & 1)) {                                                                       \
      if (next_chunk || ((T) < 8)) VP_STORE_W(((T) + 1) % 3, wb ^ 1)                                  \
      VP_LOAD_W(((T) + 1) % 3, c * 9 + (T) + 4)                                                       \
      if constexpr ((T) >= 2 && (T) - 2 < HP) {                                                       \
        if (next_chunk) VP_STORE_H(((T) + 1) % 3 /* == (T - 2) % 3, never negative */, (T) >= 2 ? (T) - 2 : 0, hb ^ 1)                     \
      }                                                                                               \
    }                                                                                                 \
    if constexpr (!(ABL & 8)) __syncthreads();                                                        \
    wb ^= 1;                                                                                          \
  }

  if constexpr ((ABL & 48) != 0) {  // experiment: de-phase the workgroups that share a CU
    const unsigned tg = __builtin_amdgcn_s_getreg((3 << 11) | (16 << 6) | 4);  // HW_ID.TG_ID
    const bool late = (ABL & 16) ? (tg & 1) : ((blockIdx.x >> 8) & 1);
    if (late)
      for (int k = 0; k < p.ks - 3; ++k) __builtin_amdgcn_s_sleep(127);
  }
  // ---- prologue: halo(c_begin) -> LDS, weights(step 0) -> LDS, weights(steps 1..3) -> ring slots 1, 2, 0
  if (c_begin < c_end) {
#pragma unroll
    for (int pc = 0; pc < HP; ++pc) {
      VP_LOAD_H(0, pc, c_begin)
      VP_STORE_H(0, pc, 0)
    }
    VP_LOAD_W(0, c_begin * 9)
    VP_STORE_W(0, 0)
    VP_LOAD_W(1, c_begin * 9 + 1)
    VP_LOAD_W(2, c_begin * 9 + 2)
    VP_LOAD_W(0, c_begin * 9 + 3)
  }
  __syncthreads();

  int hb = 0, wb = 0;
  for (int c = c_begin; c < c_end; ++c) {
    const bool next_chunk = (c + 1 < c_end);
    const char* hbuf = halo_base + hb * NPL * HALO_BYTES;
    VP_TAP(0) VP_TAP(1) VP_TAP(2) VP_TAP(3) VP_TAP(4) VP_TAP(5) VP_TAP(6) VP_TAP(7) VP_TAP(8)
    hb ^= 1;
  }
#undef VP_TAP
#undef VP_STORE_H
#undef VP_LOAD_H
#undef VP_STORE_W
#undef VP_LOAD_W

  // ---- epilogue through LDS (conv_epilogue.hpp): accumulators -> stage[pixel][channel] -> coalesced 16-byte stores
  const PixPatch<TW> pix{y0, x0, p.H, p.W};
  if constexpr (FASTEPI) {
    epilogue_regs_fp16<PX, CO_TILE, WCO, MT, NT, ACT_GELU_F16, STORE_NHWC, true>(p, smem, acc, co0, wco, wpx, pix);
  } else {
#pragma unroll
    for (int i = 0; i < MT; ++i) epilogue_pass<PX, WCO, NT>(p, smem, acc[i], co0 + i * WCO * 32, wco, wpx, pix, M, zsplit);
  }
}

template <int CO, int TH, int TW, int WCO, int WPX, bool SPLIT>
static hipError_t launch_halo_cfg(const ConvGemmParams& p, hipStream_t st) {
  constexpr int lds_main = 2 * (SPLIT ? 2 : 1) * ((TH + 2) * (TW + 2) * 80 + CO * 64);
  constexpr int lds_a = lds_main > epilogue_stage_bytes<TH * TW, WCO>() ? lds_main : epilogue_stage_bytes<TH * TW, WCO>();
  constexpr int lds = lds_a > epilogue_fp16_stage_bytes<TH * TW, CO>() ? lds_a : epilogue_fp16_stage_bytes<TH * TW, CO>();
  static_assert(lds <= 160 * 1024, "LDS budget");
  const bool fast = !SPLIT && p.act == ACT_GELU_F16 && p.res_mode == RES_NONE && p.store_mode == STORE_NHWC && p.nsplit == 1 &&
                    p.out_lo == nullptr && p.w8 == nullptr;
  auto k = fast ? conv3x3_halo_kernel<CO, TH, TW, WCO, WPX, SPLIT, 0, !SPLIT> : conv3x3_halo_kernel<CO, TH, TW, WCO, WPX, SPLIT, 0, false>;
  if (p.w8 != nullptr) {   // fp8 weight storage: the 8x16-pixel tiles of 64 and 32 channels only (halo_w8_tile_ok; the engine selects nothing else)
    if constexpr (TH == 8 && CO <= 64) k = conv3x3_halo_kernel<CO, TH, TW, WCO, WPX, SPLIT, 0, false, true>;
    else return hipErrorInvalidValue;
  }
  static LdsAttrOnce attr_once[3];
  if (hipError_t e = set_max_dynamic_lds(attr_once[p.w8 ? 2 : fast], reinterpret_cast<const void*>(k), lds); e != hipSuccess) return e;
  dim3 grid(((p.H + TH - 1) / TH) * ((p.W + TW - 1) / TW) * (p.CoutW / CO) * p.nsplit);  // decoded in the kernel (XCD-aware)
  hipLaunchKernelGGL(k, grid, dim3(256), lds, st, p);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return e;
  if (p.nsplit > 1) e = launch_splitk_finish(p, st);
  return e;
}

// halo tile ids: 0 = 128co x (16x16)px, 1 = 128co x (8x16)px, 2 = 64co x (16x16)px, 3 = 64co x (8x16)px, 4 = 32co x (8x16)px,
//                5 = 32co x (16x16)px; fp16x3 only (kernels_conv3x3_x3.hip): 6 = 128co x (16x16)px with 8 waves,
//                7 = 128co x (8x16)px with 4 waves, two workgroups per CU
hipError_t launch_conv3x3_halo(const ConvGemmParams& p, int tile, bool split, hipStream_t st) {
#define VP_HCASE(T, CO, TH, TW, WCO, WPX) \
  if (tile == T) return split ? launch_halo_cfg<CO, TH, TW, WCO, WPX, true>(p, st) : launch_halo_cfg<CO, TH, TW, WCO, WPX, false>(p, st);
  if (tile >= 6 && tile <= 9) return launch_conv3x3_x3(p, tile, st);  // pipelined kernels (kernels_conv3x3_x3.hip): fp16x3, or fp16 on 64-channel chunks (its own weight layout)
  if (tile == 3 && !split) return launch_halo_cfg<64, 8, 16, 1, 4, false>(p, st);
  VP_HCASE(1, 128, 8, 16, 2, 2)
  VP_HCASE(3, 64, 8, 16, 2, 2)
  VP_HCASE(4, 32, 8, 16, 1, 4)
  VP_HCASE(5, 32, 16, 16, 1, 4)
#undef VP_HCASE
  if (tile == 0) return split ? hipErrorInvalidValue : launch_halo_cfg<128, 16, 16, 2, 2, false>(p, st);
  // 64-channel tiles: the four waves sit side by side along the pixels (each 64co x 64/32 px, MT = 2): one fragment read
  // per MFMA instead of 1.25 with the 2x2 arrangement (decode_layer_9 51.8 -> 49.8 us, decode_layer_5 47.4 -> 45.5 us)
  if (tile == 2) return split ? hipErrorInvalidValue : launch_halo_cfg<64, 16, 16, 1, 4, false>(p, st);
  return hipErrorInvalidValue;
}
int halo_tile_co(int tile) { return (tile <= 1 || tile == 6 || tile == 7) ? 128 : ((tile <= 3 || tile == 8 || tile == 9 || tile == 12) ? 64 : 32); }
int halo_tile_px(int tile) { return (tile == 0 || tile == 2 || tile == 5 || tile == 6 || tile == 9) ? 256 : 128; }
int halo_tile_th(int tile) { return (tile == 0 || tile == 2 || tile == 5 || tile == 6 || tile == 9) ? 16 : 8; }

}  // namespace vp
