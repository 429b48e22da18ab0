// 3x3 / stride 1 / pad 1 convolution of the PARITY mode (VP_FP16X3) on large maps: software-pipelined MFMA fragments,
// two waves per SIMD.
//
// In fp16x3 every tensor is a (hi, lo) fp16 pair and every product is three MFMAs, so the halo kernel's LDS plan
// (kernels_conv3x3.hip) doubles: its 128-channel tile needs 90 KiB, ONE 4-wave workgroup per CU, one wave per SIMD --
// and that lone wave alternates "ds_read fragments -> wait -> MFMA -> stage -> barrier", leaving the matrix pipe idle for
// every LDS round trip (measured 265-290 TFLOP/s algorithmic = 32 % of the fp16 MFMA peak on the big decoder layers,
// which are half of a SceneSeg + Scene3D frame).  The kernels here keep the data flow (halo tile resident in LDS, one weight
// tile per tap, XCD-aware tile map, identical K order => bit-identical accumulation) and change the execution shape:
//   * every wave owns 64 channels x 64 pixels (2 x 2 MFMA tiles, 12 MFMAs per 16-channel K sub-step) and TWO waves share
//     each SIMD's matrix pipe;
//   * fragments are prefetched ONE K SUB-STEP AHEAD (the 8 ds_read_b128 of sub-step k+1 are issued before the 12 MFMAs of
//     sub-step k), across tap boundaries too: a THIRD weight buffer in LDS makes tap t+1's weights resident before tap t
//     starts and the next tap's pixel operand is the same halo image at a shifted address.  The per-tap barrier sits
//     BETWEEN the two K sub-steps of a tap, where no prefetch is in flight (a barrier drains the LDS queue);
//   * weight tiles go global -> LDS by LDS-DMA (global_load_lds_dwordx4: 1 KiB per wave instruction, no VGPRs, no
//     ds_write): the host packs each (chunk, tap) tile in its LDS IMAGE order (XOR-swizzled 16-byte chunks), so a linear
//     copy lands conflict-free; tile s+2 is requested when step s starts and awaited (vmcnt) before step s's barrier.
//     With the register-staged version the 16 KiB of ds_write_b128 per tap (13 cycles each) plus their vmcnt waits cost
//     47 of 189 us on decode_layer_4 (tools/x3_ablate.hip, profiles/r02_x3_ablate.txt);
//   * halo pieces travel global -> registers (loaded three taps before their LDS store) -> LDS: border pixels are zeroed
//     in registers, and the 80-byte halo pitch is not a linear copy.  (The halo by LDS-DMA with a zero page for the border was
//     measured SLOWER, profiles/r02_x3_halo_dma_ab.txt, and so was this schedule on the VP_FP16 engines' single planes: DESIGN.md
//     "tried and dropped"; neither is in the library);
//   * register epilogue: bias + exact-erf GELU + (hi, lo) split on the accumulators, BOTH fp16 planes staged once in LDS
//     and written as 256 contiguous bytes per pixel and plane.
// Shapes (halo tile ids, kernels.hpp):
//   6 "x3w8": 512 threads, 16x16 pixels x 128 channels, halo double-buffered (the next chunk's halo is complete five taps
//             before it is needed, so the prefetch also crosses chunk boundaries).  152 832 B of LDS: one workgroup per CU,
//             the two waves of a SIMD belong to the same workgroup and reach prologue / epilogue together.
//   7 "x3w4": 256 threads, 8x16 pixels x 128 channels, halo SINGLE-buffered (the next chunk's halo waits in registers and
//             is written between two barriers at the chunk boundary).  77 952 B: TWO INDEPENDENT workgroups per CU, one
//             wave of each per SIMD -- they drift out of phase, so one workgroup's prologue / chunk hand-over / epilogue
//             (exact GELU + split: ~1000 VALU instructions per wave, 64 KiB of stores) runs under the other's MFMAs.
//   8 "x3w4c64": shape 7 on 64-channel tiles (53 KB, three workgroups per CU).
//   9 "x3w4c64t16" (round 5): 16x16 pixels x 64 channels, 4 waves SIDE BY SIDE along the pixels (WCO = 1): every wave owns all 64 channels x 64 pixels
//             = 2 x 2 MFMA tiles, 8 fragment reads per 12 MFMAs like shape 6 (shape 8's waves own 32 channels x 64 pixels: 6 reads per 6 MFMAs -- as
//             LDS-bound as the halo kernel's 64-channel tile, profiles/r04: decode_layer_5 / decode_layer_9 at 0.11 of peak); halo single-buffered
//             (six pieces per thread wait in registers), 76 416 B: two independent workgroups per CU.
#include <algorithm>
#include <cstdlib>
#include <type_traits>

#include "conv_epilogue.hpp"
#include "lds_dma.hpp"

namespace vp {

// ABL: ablation bits for tools/x3_ablate.hip only (1 = no global loads / LDS stores in the loop, 2 = no MFMA, 4 = no LDS
// fragment reads in the loop, 8 = no barrier in the loop, 16 = no epilogue arithmetic / stores, 32 = clock probe: workgroup 0
// writes {shader-clock ticks, 100 MHz wall ticks} of its K loop to p.partial[0..1] as raw 64-bit counters, 64 = no halo staging
// in the loop (weights still stream), 128 = no weight DMA in the loop (halo still streams), 256 = __syncthreads() instead of
// VP_LDS_BARRIER in the loop: every barrier drains vmcnt(0), lds_dma.hpp), 512 / 1024 / 2048 = ConvTranspose-fusion estimate (below); always 0 in the library.
// Measured with them (profiles/r02_x3_clock_probe.txt): either stream alone is free (255 k cycles of the 8-wave K loop on
// decode_layer_4 = 87 % matrix-pipe busy), both together cost 290 k (76 %): s_waitcnt vmcnt retires in issue order, so a wait
// for a weight tile (L2 hit) also waits for the older halo loads (HBM).  Splitting the roles between waves (waves 0-3 DMA,
// waves 4-7 halo, one of each per SIMD) was tried and is WORSE (348 k cycles, 64 %): the halo waves, with twice the pieces
// each, become the stragglers of every barrier.
// SPLITK: grid carries p.nsplit K slices per tile; a slice covers the input chunks [KC * z / nsplit, KC * (z + 1) / nsplit) and
// writes its fp32 accumulators to p.partial[z][pixel][CoutW]; splitk_finish_kernel (kernels_conv.hip) sums the slices in the
// fixed order z = 0..nsplit-1 and applies bias / activation / (hi, lo) split -- the small-map neck layers (20x40, 40x80).
// Stream-K over these tiles (persistent workgroups, the tiles' K loops cut into equal chunk-step ranges, fp32 slab hand-off to the tile's
// owner) was built, verified on the MI355X and measured in round 3: it removes the tile-quantisation tail (200 / 400 / 1600 tiles on 256 /
// 512 slots) and changes nothing in the layer times (profiles/r03_streamk_layers.tsv) -- the big layers are bound chip-wide, not per
// workgroup.  Kept out of the library: tools/dropped/kernels_conv3x3_x3_streamk.hip, DESIGN.md "tried and dropped".
// X1 (round 4): the VP_FP16 engines' form of the same schedule.  One fp16 plane per tensor and ONE MFMA per product would leave a third of the
// matrix work per barrier / staging step (measured slower than the halo kernel in round 2); instead a step covers SIXTY-FOUR input channels and
// the two "planes" of every LDS image are the two 32-channel halves of that chunk: plane 0 = channels [64c, 64c + 32), plane 1 = [64c + 32,
// 64c + 64) of the one fp16 tensor (weights packed likewise by the engine: w_hi / w_lo carry the two halves).  Same LDS plan, same DMA and
// halo traffic per step, two MFMAs per fragment pair instead of three -- two thirds of the parity kernel's matrix work per barrier.
template <int CO_TILE, int TH, int WCO, int WPX, bool HDB, int ACT, int ABL = 0, bool SPLITK = false, bool X1 = false>
__global__ __launch_bounds__(64 * WCO * WPX, (CO_TILE == 64 && TH == 8) ? 3 : 2) void conv3x3_x3_kernel(const ConvGemmParams p) {
  constexpr int NTH = 64 * WCO * WPX;
  constexpr int TW = 16, ROWB = 80, HWD = TW + 2, HPX = (TH + 2) * HWD, PX = TH * TW;
  constexpr int HALO_BYTES = HPX * ROWB, WROW = 64, W_BYTES = CO_TILE * WROW;
  constexpr int HCHUNKS = HPX * 4, HP = (HCHUNKS + NTH - 1) / NTH;
  constexpr int MT = CO_TILE / WCO / 32, NT = PX / WPX / 32;
  constexpr int NHB = HDB ? 2 : 1;
  constexpr int PL = 2;                // planes per tensor: (hi, lo)
  constexpr int HSTRIDE = HALO_BYTES;  // plane stride in LDS
  constexpr int LT = 0;                // HDB: tap at which the next chunk's halo pieces are loaded (stored at the start of tap 3)
  constexpr int RING = HP > 3 ? HP : 3;   // halo staging ring (registers): one slot per piece a thread moves (shape 9: six)
  static_assert(MT >= 1 && NT >= 1 && HP <= 6 && (HP <= 3 || !HDB), "tile shape");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* const halo_base = smem;                               // [NHB][2 planes][HALO_BYTES]
  char* const w_base = smem + NHB * PL * HSTRIDE;             // [3][PL planes][W_BYTES]

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wco = wave / WPX, wpx = wave % WPX;
  const int tiles_x = (p.W + TW - 1) / TW;
  const int n_px_tiles = tiles_x * ((p.H + TH - 1) / TH);
  const int n_co_tiles = p.CoutW / CO_TILE;
  constexpr int CSTEP = X1 ? 64 : 32;  // input channels per chunk
  const int KC_all = p.Cin / CSTEP;
  const half_t* const in_p1 = X1 ? p.in_hi + 32 : p.in_lo;  // second plane of the halo image
  int vid;  // XCD-aware workgroup -> tile map (see kernels_conv3x3.hip)
  {
    const int nwg = gridDim.x, q = nwg >> 3, r = nwg & 7, xcd = blockIdx.x & 7;
    vid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (blockIdx.x >> 3);
  }

  const int h_lds0 = (tid >> 2) * ROWB + (tid & 3) * 16;
  constexpr int NW = NTH / 64, WPIECES = W_BYTES / 1024 / NW;
  static_assert(W_BYTES % (1024 * NW) == 0, "weight tile splits into 1 KiB pieces per wave");
  const size_t w_step = (size_t)p.CoutW * 32;
  typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
  const u32x4 zero4 = {0u, 0u, 0u, 0u};
  // fragment addressing (same LDS image and lane maps as the halo kernel)
  int b_ofs0;  // pixel tile j of the wave sits two halo rows further: + j * 2 * HWD * ROWB
  {
    int rowbit, px;
    lane_to_px16(lane & 31, rowbit, px);
    b_ofs0 = ((2 * (wpx * NT) + rowbit) * HWD + px) * ROWB + (lane >> 5) * 16;
  }
  const int a_swz = ((lane & 31) >> 2) & 3;
  const int a_ofs0 = (wco * 32 + (lane & 31)) * WROW + (((lane >> 5) ^ a_swz) << 4);  // K sub-step 1: chunk index ^ 2 -> ^ 32 bytes

  int c_first = 0, KC = KC_all, zsplit = 0;
  // Without split-K the OUTPUT-CHANNEL tile is the fastest index (round 4): the two or four channel tiles of one pixel tile get neighbouring ids, i.e.
  // run side by side on one XCD, and the second one finds the input patch in that XCD's L2 (with the pixel tile fastest they ran a whole
  // round apart: measured 139.9 -> 135.6 us per launch over decode_layer_4 / 6 / 7, `profiles/r04_x3_cofast_ab.txt`; the weights of all
  // channel tiles -- 1.2-4.7 MB -- stay resident in the 4 MB L2 either way).  Same tiles, same K order: bit-identical results.
  const int tile_px = SPLITK ? vid % n_px_tiles : vid / n_co_tiles;
  const int tile_rest = vid / n_px_tiles;
  const int tile_co = SPLITK ? tile_rest % n_co_tiles : vid % n_co_tiles;
  if constexpr (SPLITK) {
    zsplit = tile_rest / n_co_tiles;
    c_first = (int)(((long long)KC_all * zsplit) / p.nsplit);
    KC = (int)(((long long)KC_all * (zsplit + 1)) / p.nsplit) - c_first;  // chunks of THIS slice; c below is slice-relative
  }
  const int tyi = tile_px / tiles_x, txi = tile_px - tyi * tiles_x;
  const int y0 = tyi * TH, x0 = txi * TW;
  const int co0 = tile_co * CO_TILE;

  // ---- staging assignment: thread t moves 16-byte piece t + NTH * pc of a tile (pieces of one thread sit NTH / 4 rows apart)
  int h_goff[HP];
#pragma unroll
  for (int pc = 0; pc < HP; ++pc) {
    const int hidx = tid + NTH * pc;
    const int hp = hidx >> 2, ch = hidx & 3;
    const int hy = hp / HWD, hx = hp - hy * HWD;
    const int gy = y0 - 1 + hy, gx = x0 - 1 + hx;
    const bool ok = hidx < HCHUNKS && (unsigned)gy < (unsigned)p.H && (unsigned)gx < (unsigned)p.W;
    h_goff[pc] = ok ? (gy * p.W + gx) * p.Cin + ch * 8 : -1;
    // ABL 512 (tools/x3_ablate.hip, the ConvTranspose -> 3x3 fusion estimate of VERDICT round 3 item 5): the halo comes from a QUARTER-RESOLUTION
    // tensor, as it would if the up-sampling ran inside this kernel (timing only: the values are meaningless)
    if constexpr ((ABL & 512) != 0) h_goff[pc] = ok ? ((gy >> 1) * (p.W >> 1) + (gx >> 1)) * p.Cin + ch * 8 : -1;
  }
  // weight tiles by LDS-DMA: a (chunk, tap) tile plane is CO_TILE x 64 B = W_BYTES contiguous bytes in global memory, already
  // in LDS image order; wave v copies the 1 KiB pieces v, v + NW, ... of both planes
  const size_t w_goff0 = (size_t)c_first * 9 * w_step + co0 * 32 + wave * 512 + lane * 8;  // elements: this lane's 16 bytes of the wave's first piece (of the segment's first tile)

  f32x16_t acc[MT][NT];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

  // fragment sets [K sub-step parity]: set 0 = channels 0..15 of the tap's 32, set 1 = channels 16..31
  h8_t fa[2][MT], fal[2][MT], fb[2][NT], fbl[2][NT];
  // ABL 4096 (prototype, tools/x3_ablate.hip): the WEIGHT fragments travel global -> registers directly, three steps deep, instead of through LDS
  // (no weight DMA, no weight ds_reads; the host packing is the LDS image, so a lane's fragment is 16 contiguous bytes at the same offset)
  constexpr bool DIRECT_A = (ABL & 4096) != 0;
  h8_t ra[DIRECT_A ? 3 : 1][2][MT], ral[DIRECT_A ? 3 : 1][2][MT];
  const size_t w_a0 = (size_t)c_first * 9 * w_step + co0 * 32;
  // halo staging ring (compile-time slots): piece pc in slot pc
  u32x4 rh_hi[RING], rh_lo[RING];
#pragma unroll
  for (int r = 0; r < RING; ++r) rh_hi[r] = rh_lo[r] = zero4;
  const int s_last = KC * 9 - 1;

  // weight tile SIDX (clamped to the last one: the tail requests are harmless re-reads) -> LDS buffer BUF, asynchronously
#define VP_DMA_W(BUF, SIDX)                                                                  \
  {                                                                                          \
    const int si_ = (SIDX) < s_last ? (SIDX) : s_last;                                       \
    const size_t base_ = (size_t)si_ * w_step + w_goff0;                                     \
    char* dst_ = w_base + (BUF) * PL * W_BYTES + wave * 1024;                                \
    _Pragma("unroll") for (int pc = 0; pc < WPIECES; ++pc) {                                 \
      VP_GLOBAL_LOAD_LDS16(p.w_hi + base_ + pc * NW * 512, dst_ + pc * NW * 1024);           \
      VP_GLOBAL_LOAD_LDS16(p.w_lo + base_ + pc * NW * 512, dst_ + W_BYTES + pc * NW * 1024); \
    }                                                                                        \
  }
#define VP_LOAD_A(RSLOT, SIDX)                                                               \
  {                                                                                          \
    const int si_ = (SIDX) < s_last ? (SIDX) : s_last;                                       \
    const size_t base_ = (size_t)si_ * w_step + w_a0;                                        \
    _Pragma("unroll") for (int ss_ = 0; ss_ < 2; ++ss_)                                      \
    _Pragma("unroll") for (int i = 0; i < MT; ++i) {                                         \
      const size_t o_ = base_ + (size_t)(((a_ofs0 ^ (ss_ * 32)) + i * WCO * 32 * WROW) >> 1); \
      ra[RSLOT][ss_][i] = *reinterpret_cast<const h8_t*>(p.w_hi + o_);                       \
      ral[RSLOT][ss_][i] = *reinterpret_cast<const h8_t*>(p.w_lo + o_);                      \
    }                                                                                        \
  }
#define VP_LOAD_H(SLOT, PC, C)                                                               \
  {                                                                                          \
    const int g_ = h_goff[PC];                                                               \
    const int o_ = (g_ >= 0 ? g_ : 0) + (c_first + (C)) * CSTEP;                             \
    const u32x4 v_ = *reinterpret_cast<const u32x4*>(p.in_hi + o_);                          \
    const u32x4 l_ = *reinterpret_cast<const u32x4*>(in_p1 + o_);                            \
    rh_hi[SLOT] = g_ >= 0 ? v_ : zero4;                                                      \
    rh_lo[SLOT] = g_ >= 0 ? l_ : zero4;                                                      \
  }
#define VP_STORE_H(SLOT, PC, BUF)                                                            \
  if (tid + NTH * (PC) < HCHUNKS) {                                                          \
    char* dst_ = halo_base + (BUF) * PL * HSTRIDE + h_lds0 + (PC) * (NTH / 4) * ROWB;        \
    *reinterpret_cast<u32x4*>(dst_) = rh_hi[SLOT];                                           \
    *reinterpret_cast<u32x4*>(dst_ + HSTRIDE) = rh_lo[SLOT];                                 \
  }
#define VP_READ_FRAGS(SET, WBUF, HBUF, TAPOFS)                                               \
  {                                                                                          \
    const char* wsrc_ = (WBUF) + (a_ofs0 ^ ((SET) * 32));                                    \
    if constexpr (!DIRECT_A) {                                                               \
    _Pragma("unroll") for (int i = 0; i < MT; ++i) {                                         \
      fa[SET][i] = *reinterpret_cast<const h8_t*>(wsrc_ + i * WCO * 32 * WROW);              \
      fal[SET][i] = *reinterpret_cast<const h8_t*>(wsrc_ + W_BYTES + i * WCO * 32 * WROW);   \
    }                                                                                        \
    }                                                                                        \
    _Pragma("unroll") for (int j = 0; j < NT; ++j) {                                         \
      fb[SET][j] = *reinterpret_cast<const h8_t*>((HBUF) + b_ofs0 + j * 2 * HWD * ROWB + (TAPOFS) + (SET) * 32); \
      fbl[SET][j] = *reinterpret_cast<const h8_t*>((HBUF) + HSTRIDE + b_ofs0 + j * 2 * HWD * ROWB + (TAPOFS) + (SET) * 32); \
    }                                                                                        \
  }
#define VP_MFMA(SET) VP_MFMA_RANGE(SET, 0, MT * NT)
  // accumulator tiles [Q0, Q1) of the wave (tile q = i * NT + j)
#define VP_MFMA_RANGE(SET, Q0, Q1)                                                           \
  _Pragma("unroll") for (int q_ = (Q0); q_ < (Q1); ++q_) {                                   \
    const int i = q_ / NT, j = q_ % NT;                                                      \
    if constexpr ((ABL & 2) != 0) { acc[i][j][0] += (float)fa[SET][i][0] + (float)fal[SET][i][1] + (float)fb[SET][j][2] + (float)fbl[SET][j][3]; continue; } \
    if constexpr (DIRECT_A) {                                                                \
      acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ral[aslot_][SET][i], fb[SET][j], acc[i][j], 0, 0, 0); \
      acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ra[aslot_][SET][i], fbl[SET][j], acc[i][j], 0, 0, 0); \
      acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ra[aslot_][SET][i], fb[SET][j], acc[i][j], 0, 0, 0);  \
      continue;                                                                              \
    }                                                                                        \
    if constexpr (X1) { /* the planes are K halves: a0 . b0 + a1 . b1 */                     \
      acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[SET][i], fb[SET][j], acc[i][j], 0, 0, 0);   \
      acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fal[SET][i], fbl[SET][j], acc[i][j], 0, 0, 0); \
      continue;                                                                              \
    }                                                                                        \
    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fal[SET][i], fb[SET][j], acc[i][j], 0, 0, 0); \
    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[SET][i], fbl[SET][j], acc[i][j], 0, 0, 0); \
    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[SET][i], fb[SET][j], acc[i][j], 0, 0, 0);  \
  }
  // One tap step.  On entry fragment set 0 of THIS step is in flight / in registers (read during the previous step).
  // HDB: the next chunk's halo pieces are loaded at taps 0..HP-1 and written to the OTHER halo buffer two taps later, the
  //      prefetch of tap 8 reads that buffer.  !HDB: the pieces stay in their ring slots until tap 8's barrier has passed,
  //      are written over the single halo image, and a second barrier opens the next chunk (no prefetch across it).
#define VP_TAP(T)                                                                            \
  {                                                                                          \
    constexpr int tap_ofs_ = (((T) / 3) * HWD + ((T) % 3)) * ROWB;                           \
    constexpr int aslot_ = DIRECT_A ? (T) % 3 : 0;   /* 9 taps per chunk: the step's ring slot is the tap's */ \
    constexpr int tn_ = ((T) + 1) % 9;                                                       \
    constexpr int tap_next_ = ((tn_ / 3) * HWD + (tn_ % 3)) * ROWB;                          \
    const char* wcur_ = w_base + ((T) % 3) * PL * W_BYTES;                                   \
    const char* wnext_ = w_base + (((T) + 1) % 3) * PL * W_BYTES;                            \
    const char* hnext_ = ((T) == 8 && HDB) ? hbuf_other : hbuf;                              \
    /* ---- K sub-step 0 (set 0 was fetched behind the previous barrier).  Half of its accumulator tiles go BEFORE set 1's */ \
    /* reads are issued: the wait in front of it then sees only reads that are a whole MFMA group old                    */ \
    /* weight tile of step s+2 -> the buffer step s-1 read last (its barrier has passed); must land before the NEXT step's barrier */ \
    /* (an L2 warm-up of tile s+5 -- every workgroup of an XCD asks for the same never-used tile at once -- was measured: -5 %) */ \
    /* HDB: the next chunk's halo pieces (ALL loaded at tap 0) go to the other halo image HERE, at the start of tap 3 and    */ \
    /* BEFORE this step's DMA is issued: the compiler guards the registers with s_waitcnt vmcnt(0) (LDS-DMA and plain loads   */ \
    /* in flight together make its counter model give up on partial counts), and at this point the only thing still in flight */ \
    /* is the weight tile requested one step ago, which this step's barrier needs anyway.  With the stores spread over taps    */ \
    /* 2..4 behind the DMA issue, each of those waits drained the just-requested tile (an L2 round trip) with the matrix pipe  */ \
    /* idle: 74 % busy in the K loop against 85 % with either stream alone (profiles/r02_x3_clock_probe.txt).                  */ \
    if constexpr (HDB && (T) == 3 && !(ABL & 1) && !(ABL & 64)) {                            \
      if (next_chunk) { _Pragma("unroll") for (int pc = 0; pc < HP; ++pc) VP_STORE_H(pc, pc, hb ^ 1) } \
    }                                                                                        \
    if constexpr (DIRECT_A) {                                                                \
      if (next_chunk || (T) < 7) VP_LOAD_A(((T) + 2) % 3, c * 9 + (T) + 2)                   \
    } else if constexpr (!(ABL & 1) && !(ABL & 128)) {                                       \
      if (next_chunk || (T) < 7) VP_DMA_W(((T) + 2) % 3, c * 9 + (T) + 2)                     \
    }                                                                                        \
    __builtin_amdgcn_sched_barrier(0);                                                       \
    VP_MFMA_RANGE(0, 0, MT * NT / 2)                                                         \
    __builtin_amdgcn_sched_barrier(0);                                                       \
    if constexpr (!(ABL & 4)) VP_READ_FRAGS(1, wcur_, hbuf, tap_ofs_)                        \
    __builtin_amdgcn_sched_barrier(0); /* keep the prefetch AHEAD of the MFMAs (the scheduler sinks it otherwise) */ \
    if constexpr (HDB && (T) == LT && !(ABL & 1) && !(ABL & 64)) {                           \
      _Pragma("unroll") for (int pc = 0; pc < HP; ++pc) VP_LOAD_H(pc, pc, next_chunk ? c + 1 : c) \
    }                                                                                        \
    if constexpr (!HDB && (T) < HP && !(ABL & 1) && !(ABL & 64)) {                           \
      if (next_chunk) VP_LOAD_H((T) % RING, (T) < HP ? (T) : 0, c + 1)                       \
    }                                                                                        \
    VP_MFMA_RANGE(0, MT * NT / 2, MT * NT)                                                   \
    /* round 4 (ISA): keep these MFMAs IN FRONT of the barrier's lgkmcnt(0) -- the scheduler moved five of the six behind it, so set 1's */ \
    /* reads were drained one MFMA after their issue                                                                                   */ \
    __builtin_amdgcn_sched_barrier(0);                                                       \
    if constexpr (!(ABL & 1) && !DIRECT_A) {                                                 \
      /* What THIS barrier must publish is the weight tile requested ONE STEP AGO (tile s+1: its first read follows this      */ \
      /* barrier); the tile requested in this step (s+2) is first read behind the NEXT barrier and stays in flight -- two taps */ \
      /* of lead for the L2 / Infinity-Cache round trip instead of one.  vmcnt counts in issue order, so "the previous step's  */ \
      /* DMA has landed" = at most {previous step's halo loads, this step's DMA, this step's halo loads} still outstanding.    */ \
      {                                                                                      \
        /* halo loads (2 per piece) issued in this step / in the previous step behind its DMA */ \
        constexpr int hon_ = (ABL & 64) ? 0 : 1;                                             \
        const int hthis_ = hon_ * (HDB ? ((T) == LT ? PL * HP : 0) : (((T) < HP && next_chunk) ? PL : 0)); \
        const int hprev_ = hon_ * (HDB ? ((T) == LT + 1 ? PL * HP : 0) : (((T) >= 1 && (T) <= HP && next_chunk) ? PL : 0)); \
        const int newer_ = ((next_chunk || (T) < 7) ? PL * WPIECES : 0) + hthis_ + hprev_;     \
        if (newer_ >= PL * WPIECES + PL * HP && HDB) { VP_WAIT_VMCNT(PL * WPIECES + PL * HP); } \
        else if (newer_ >= PL * WPIECES + 2 * PL) { VP_WAIT_VMCNT(PL * WPIECES + 2 * PL); }    \
        else if (newer_ >= PL * WPIECES + PL) { VP_WAIT_VMCNT(PL * WPIECES + PL); }            \
        else if (newer_ >= PL * WPIECES) { VP_WAIT_VMCNT(PL * WPIECES); }                      \
        else if (newer_ >= 2 * PL) { VP_WAIT_VMCNT(2 * PL); }                                \
        else if (newer_ >= PL) { VP_WAIT_VMCNT(PL); }                                        \
        else { VP_WAIT_VMCNT(0); }                                                           \
      }                                                                                      \
    }                                                                                        \
    /* THE BARRIER SITS BETWEEN THE TWO K SUB-STEPS: the only LDS operations outstanding here are set 1's reads (issued a    */ \
    /* whole MFMA group ago) and, on three taps of nine, two halo stores.  With the next step's prefetch issued BEFORE the    */ \
    /* barrier all eight waves drained 64 ds_read_b128 in lockstep at every step with the matrix pipe idle.                  */ \
    if constexpr ((ABL & 256) != 0) __syncthreads(); else if constexpr (!(ABL & 8)) VP_LDS_BARRIER();  \
    if constexpr (!HDB && (T) == 8) {                                                        \
      if (next_chunk && !(ABL & 1) && !(ABL & 64)) {                                         \
        _Pragma("unroll") for (int pc = 0; pc < HP; ++pc) VP_STORE_H(pc, pc, 0)              \
        if constexpr ((ABL & 256) != 0) __syncthreads(); else if constexpr (!(ABL & 8)) VP_LDS_BARRIER();  \
      }                                                                                      \
    }                                                                                        \
    /* ---- K sub-step 1: set 0 of the NEXT step is fetched while set 1 multiplies */       \
    if constexpr (!(ABL & 4)) VP_READ_FRAGS(0, wnext_, hnext_, tap_next_)                    \
    __builtin_amdgcn_sched_barrier(0);                                                       \
    VP_MFMA(1)                                                                               \
    /* ABL 1024 / 2048 (same estimate): one / two extra taps' worth of MFMAs per chunk = +11 % / +22 % matrix work, the price of the  */ \
    /* in-kernel up-sampling GEMM (+17 % on these layers)                                                                             */ \
    if constexpr (((ABL & 1024) != 0 && (T) == 4) || ((ABL & 2048) != 0 && (T) == 7)) { VP_MFMA(0) VP_MFMA(1) } \
    /* round 4 (seen in the ISA): without this fence the scheduler hoists the NEXT step's LDS-DMA issue to the head of this MFMA */ \
    /* group, right behind set 0's reads -- and an LDS-DMA may overwrite what an outstanding ds_read reads, so the compiler puts   */ \
    /* s_waitcnt lgkmcnt(0) in front of it: the prefetch was drained the moment it was issued.  Behind the 12 MFMAs it has landed. */ \
    __builtin_amdgcn_sched_barrier(0);                                                       \
  }

  // ---- prologue: halo(chunk 0) and weight tiles 0, 1 -> LDS
#pragma unroll
  for (int pc = 0; pc < HP; ++pc) {
    VP_LOAD_H(0, pc, 0)
    VP_STORE_H(0, pc, 0)
  }
  if constexpr (DIRECT_A) {
    VP_LOAD_A(0, 0)
    VP_LOAD_A(1, 1)
  } else {
    VP_DMA_W(0, 0)
    VP_DMA_W(1, 1)
  }
  VP_WAIT_VMCNT(0);
  __syncthreads();
  VP_READ_FRAGS(0, w_base, halo_base, 0)
  if constexpr ((ABL & 4) != 0) VP_READ_FRAGS(1, w_base, halo_base, 0)

  unsigned long long probe_c0 = 0, probe_w0 = 0;
  if constexpr ((ABL & 32) != 0) {
    probe_c0 = __builtin_readcyclecounter();   // s_memtime: shader clock
    probe_w0 = __builtin_amdgcn_s_memrealtime();  // constant 100 MHz
  }
  int hb = 0;
  for (int c = 0; c < KC; ++c) {
    const bool next_chunk = (c + 1 < KC);
    const char* hbuf = halo_base + (HDB ? hb : 0) * PL * HSTRIDE;
    const char* hbuf_other = halo_base + (HDB ? (hb ^ 1) : 0) * PL * HSTRIDE;
    VP_TAP(0) VP_TAP(1) VP_TAP(2) VP_TAP(3) VP_TAP(4) VP_TAP(5) VP_TAP(6) VP_TAP(7) VP_TAP(8)
    hb ^= 1;
  }
  if constexpr ((ABL & 32) != 0) {
    if (blockIdx.x == 0 && tid == 0) {
      unsigned long long* dst = reinterpret_cast<unsigned long long*>(p.partial);
      dst[0] = __builtin_readcyclecounter() - probe_c0;
      dst[1] = __builtin_amdgcn_s_memrealtime() - probe_w0;
    }
  }
#undef VP_TAP
#undef VP_MFMA
#undef VP_MFMA_RANGE
#undef VP_READ_FRAGS
#undef VP_STORE_H
#undef VP_LOAD_H
#undef VP_LOAD_A
#undef VP_DMA_W

  // ---- register epilogue: bias + activation + (hi, lo) split, both planes staged as [pixel][CO_TILE] fp16, 16-byte stores.
  constexpr int PITCH = CO_TILE * 2 + 16, STAGE_PLANE = PX * PITCH;
  static_assert(PL * STAGE_PLANE <= NHB * PL * HSTRIDE + 3 * PL * W_BYTES, "stage fits the main buffers");
  if constexpr ((ABL & 16) != 0) {  // ablation: keep the accumulators alive (a store no launch ever takes), skip arithmetic and stores
    if (p.H == -12345) {
#pragma unroll
      for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
          for (int r = 0; r < 16; ++r) p.out_f32[(i * NT + j) * 16 + r + tid * 64] = acc[i][j][r];
    }
    return;
  }
  if constexpr (SPLITK) {
    // fp32 partial sums straight from the accumulators: lanes l and l + 32 hold channels 8g + 0..3 / 8g + 4..7 of pixel l & 31,
    // together 32 contiguous bytes per register group
    const PixPatch<TW> pixs{y0, x0, p.H, p.W};
    const int M = p.H * p.W;
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      const int m = pixs((wpx * NT + j) * 32 + (lane & 31));
      if (m < 0) continue;
      float* row = p.partial + ((size_t)zsplit * M + m) * p.CoutW + co0 + 4 * (lane >> 5);
#pragma unroll
      for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const f32x4_t v = {acc[i][j][4 * g + 0], acc[i][j][4 * g + 1], acc[i][j][4 * g + 2], acc[i][j][4 * g + 3]};
          *reinterpret_cast<f32x4_t*>(row + (i * WCO + wco) * 32 + 8 * g) = v;
        }
    }
    return;
  }
  __syncthreads();  // every wave has finished its last K sub-step (and the dead prefetch behind the last barrier has landed)
  const PixPatch<TW> pix{y0, x0, p.H, p.W};
#pragma unroll
  for (int i = 0; i < MT; ++i) {
    const int cl = (i * WCO + wco) * 32 + 4 * (lane >> 5);
    f32x4_t b[4], sc[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      b[g] = *reinterpret_cast<const f32x4_t*>(p.bias + co0 + cl + 8 * g);
      sc[g] = *reinterpret_cast<const f32x4_t*>(p.wscale + co0 + cl + 8 * g);  // 2^-prescale of the weight rows: exact product
    }
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      char* row = smem + ((wpx * NT + j) * 32 + (lane & 31)) * PITCH + cl * 2;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        h4_t h, l;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float x = apply_act(fmaf(acc[i][j][4 * g + r], sc[g][r], b[g][r]), ACT);
          h[r] = (half_t)x;
          if constexpr (!X1) l[r] = (half_t)(x - (float)h[r]);
        }
        *reinterpret_cast<h4_t*>(row + g * 16) = h;
        if constexpr (!X1) *reinterpret_cast<h4_t*>(row + STAGE_PLANE + g * 16) = l;
      }
    }
  }
  __syncthreads();
  constexpr int CPR = CO_TILE / 8, RPI = NTH / CPR;
  static_assert(NTH % CPR == 0 && PX % RPI == 0, "row loop shape");
  const int c8 = tid % CPR, r0 = tid / CPR;
  const int co = co0 + c8 * 8;
  if (co >= p.Ncols) return;
#pragma unroll 4
  for (int r = r0; r < PX; r += RPI) {
    const int m = pix(r);
    if (m < 0) continue;
    const size_t o = (size_t)m * p.Cstore + co;
    *reinterpret_cast<h8_t*>(p.out_hi + o) = *reinterpret_cast<const h8_t*>(smem + r * PITCH + c8 * 16);
    if constexpr (!X1) *reinterpret_cast<h8_t*>(p.out_lo + o) = *reinterpret_cast<const h8_t*>(smem + STAGE_PLANE + r * PITCH + c8 * 16);
  }
}

// shape 8 ("x3w4c64"): 8x16 pixels x 64 channels, 4 waves of 32 channels x 64 pixels, 53 KB of LDS: THREE independent workgroups per CU.
// The layers whose 128-channel tiles are too few (64-channel outputs, small maps with split-K): same workgroup count and split
// factor as the halo kernel's 64-channel tile (halo tile 3), the pipelined schedule instead of its lone-wave one.
bool conv3x3_x3_supported(const ConvGemmParams& p, int shape) {
  const int co_tile = (shape == 8 || shape == 9) ? 64 : 128;
  const bool x1 = p.in_lo == nullptr;  // VP_FP16 engines: the two planes are the halves of a 64-channel chunk (template parameter X1)
  if (!(p.ks == 3 && p.stride <= 1 && p.in_hi && p.w_hi && p.w_lo && p.CoutW % co_tile == 0 && p.Cin % (x1 ? 64 : 32) == 0 && p.Cin2 == 0 && p.nsplit >= 1)) return false;
  const bool act_ok = x1 ? (p.act == ACT_GELU_F16 || p.act == ACT_NONE) : (p.act == ACT_GELU || p.act == ACT_NONE);
  const bool plain = p.out_hi && (x1 ? p.out_lo == nullptr : p.out_lo != nullptr) && p.store_mode == STORE_NHWC && p.res_mode == RES_NONE && p.post_act == ACT_NONE && act_ok;
  if (p.nsplit > 1) return p.partial != nullptr && p.nsplit <= p.Cin / (x1 ? 64 : 32);  // any epilogue: the finish kernel applies it
  return plain;
}

template <int CO, int TH, int WPX, bool HDB, bool X1, int WCO = 2>
static hipError_t launch_x3_cfg(const ConvGemmParams& p, hipStream_t st) {
  constexpr int lds = (HDB ? 2 : 1) * 2 * ((TH + 2) * 18 * 80) + 6 * (CO * 64);
  static_assert(lds <= 160 * 1024, "LDS budget");
  constexpr int GELU = X1 ? ACT_GELU_F16 : ACT_GELU;
  const bool gelu = p.act == GELU, sk = p.nsplit > 1;
  auto k = sk ? conv3x3_x3_kernel<CO, TH, WCO, WPX, HDB, ACT_NONE, 0, true, X1>
              : (gelu ? conv3x3_x3_kernel<CO, TH, WCO, WPX, HDB, GELU, 0, false, X1> : conv3x3_x3_kernel<CO, TH, WCO, WPX, HDB, ACT_NONE, 0, false, X1>);
  static LdsAttrOnce attr_once[3];
  if (hipError_t e = set_max_dynamic_lds(attr_once[sk ? 2 : gelu], reinterpret_cast<const void*>(k), lds); e != hipSuccess) return e;
  dim3 grid(((p.H + TH - 1) / TH) * ((p.W + 15) / 16) * (p.CoutW / CO) * p.nsplit);
  hipLaunchKernelGGL(k, grid, dim3(64 * WCO * WPX), lds, st, p);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return e;
  return sk ? launch_splitk_finish(p, st) : hipSuccess;
}

// shape 6: 16x16 pixels, 8 waves, one workgroup per CU; shape 7: 8x16 pixels, 4 waves, two independent workgroups per CU;
// shape 8: shape 7 on 64-channel tiles
hipError_t launch_conv3x3_x3(const ConvGemmParams& p, int shape, hipStream_t st) {
  if (!conv3x3_x3_supported(p, shape)) return hipErrorInvalidValue;
  if (p.in_lo == nullptr) {  // VP_FP16 engines
    if (shape == 6) return launch_x3_cfg<128, 16, 4, true, true>(p, st);
    if (shape == 7) return launch_x3_cfg<128, 8, 2, false, true>(p, st);
    if (shape == 8) return launch_x3_cfg<64, 8, 2, false, true>(p, st);
    if (shape == 9) return launch_x3_cfg<64, 16, 4, false, true, 1>(p, st);
    return hipErrorInvalidValue;
  }
  if (shape == 6) return launch_x3_cfg<128, 16, 4, true, false>(p, st);
  if (shape == 7) return launch_x3_cfg<128, 8, 2, false, false>(p, st);
  if (shape == 8) return launch_x3_cfg<64, 8, 2, false, false>(p, st);
  if (shape == 9) return launch_x3_cfg<64, 16, 4, false, false, 1>(p, st);
  return hipErrorInvalidValue;
}

}  // namespace vp
