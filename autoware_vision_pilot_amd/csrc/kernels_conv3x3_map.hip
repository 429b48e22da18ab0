// 3x3 / stride 1 / pad 1 convolution of the PARITY mode on the SMALL maps of the neck (20x40 and 40x80 pixels, 512-1280 input channels):
// one workgroup = one WEIGHT SLAB (32 output channels x a K slice) held against ALL pixels of a 20x40 region.
//
// decode_layer_0..3 are 61 GFLOP of the 367 of a network and a quarter of its decoder time (70 / 47 / 82 / 60 us, 0.07-0.11 of the
// matrix peak, round 2): on 800 / 3200 pixels the 8x16-pixel tiles of kernels_conv3x3.hip re-stream the whole weight tensor once per
// pixel tile -- 35 MB of (hi, lo) weights x 7 tiles on decode_layer_0, ~250 MB through the L2s for a 5 MB fp32 tensor -- and that
// traffic, not the matrix pipe, is their bound.  Turned round here:
//   * a workgroup (8 waves) owns a REGION of 20x40 = 800 pixels (25 MFMA pixel tiles; the 20x40 maps are one region, the 40x80 maps
//     four) x 32 output channels x a slice of the input channels, so every weight byte is fetched by R = 1 (4) workgroups instead of 7
//     (25) and the input by Cout / 32 of them; the fp32 accumulators of the whole region stay in registers (wave w: pixel tiles w,
//     w + 8, w + 16 (, 24): 48-64 registers);
//   * K advances in steps of 16 input channels (one MFMA K sub-step): the region's halo image, 22x42 pixels x 32 B x (hi, lo) = 59 KB,
//     and the step's nine 32x16 weight tiles, 18 KB, are DOUBLE-buffered in LDS (155 KB: one workgroup per CU) and filled by LDS-DMA
//     (global_load_lds_dwordx4) one step ahead, under the current step's 108 MFMAs per wave; pixels outside the map come from the zero
//     page; one LDS-only barrier per step;
//   * 32-byte rows: the two 16-byte slots of a row are XOR-swizzled by bit 3 of the row index, applied to the GLOBAL address a lane
//     supplies (the DMA itself is linear), so the fixed 16-lane groups of ds_read_b128 hit sixteen distinct bank slots; the weights
//     are packed on the host in that LDS image order, one contiguous 9 KB block per (channel tile, step, plane);
//   * a tap is an LDS address offset, as in the other 3x3 kernels; the swizzle bit of a fragment read is recomputed per tap (4 VALU
//     instructions per read);
//   * output: fp32 partial sums straight from the accumulators to p.partial[z][pixel][CoutW]; splitk_finish_kernel (kernels_conv.hip)
//     sums the K slices in z order and applies bias / activation / (hi, lo) split, as for every split layer.
// Same products, a different summation order than the tiled kernels (per K slice: channels in steps of 16, taps inside).
#include "conv_epilogue.hpp"
#include "lds_dma.hpp"

namespace vp {

// Geometry of one instantiation: RH x RW region, NWV waves.  <20, 40, 8> = the neck's 20x40 / 40x80 maps (halo tile id 11);
// <10, 20, 4> = the CONTEXT block's 10x20 maps (round 4, halo tile id 12: context_layer_4..6 ran on the halo kernel's 8x16 tiles with
// split-K at 0.002-0.03 of the matrix peak -- two pixel tiles re-streaming up to 23.6 MB of (hi, lo) weights each, 20 / 18 / 33 us + finish):
// 200 pixels = 7 MFMA pixel tiles (the last one ragged: 8 pixels), four waves carry 2 / 2 / 2 / 1 of them, 74 KB of LDS: two workgroups per CU.
template <int RH_, int RW_, int NWV_>
struct MapGeom {
  static constexpr int RH = RH_, RW = RW_, RPX = RH * RW, NWV = NWV_;   // region, waves
  static constexpr int HW = RW + 2, HH = RH + 2, HPX = HH * HW;         // its halo image
  static constexpr int NF = (RPX + 31) / 32;                            // pixel tiles of 32 (the last one may be ragged)
  static constexpr int HROWS = (HPX + 31) / 32 * 32;                    // rows per plane incl. the padding of the last DMA group
  static constexpr int H_PLANE = HROWS * 32, W_PLANE = 9 * 32 * 32;     // bytes: halo plane, weight plane of one step (9 216)
  static constexpr int H_BUF = 2 * H_PLANE, W_BUF = 2 * W_PLANE;        // (hi, lo)
  static constexpr int LDS = 2 * H_BUF + 2 * W_BUF;
  static constexpr int NHI = HROWS / 32, NWI = W_PLANE / 1024;          // DMA instructions per plane
  static constexpr int NINSTR = 2 * NHI + 2 * NWI;                      // per step
  static constexpr int IPW = (NINSTR + NWV - 1) / NWV;                  // per wave
  static constexpr int NFW_MAX = (NF + NWV - 1) / NWV;                  // pixel tiles of wave 0 (wave w: tiles w, w + NWV, ...)
  static_assert(LDS <= 160 * 1024, "LDS plan");
};
namespace mapk {
using Neck = MapGeom<20, 40, 8>;   // 924 halo pixels, 25 tiles, 155 648 B of LDS: one workgroup per CU
using Ctx = MapGeom<10, 20, 4>;    // 264 halo pixels, 7 tiles (the last: 8 pixels), 73 728 B: two workgroups per CU
static_assert(Neck::NF == 25 && Neck::LDS == 155648 && Neck::IPW == 10 && Ctx::NF == 7 && Ctx::IPW == 9, "geometry");
}  // namespace mapk

// NFW = pixel tiles of THIS wave: 4 for wave 0 (tiles 0, 8, 16, 24), 3 for the others.  The body is instantiated per count and the kernel
// branches ONCE on the (scalar) wave index: with `if (tile < 25)` tests inside the tap loop the compiler guarded every tap with
// s_waitcnt lgkmcnt(0) at the joins, i.e. the fragments requested one tap ahead were awaited BEFORE the current tap's MFMAs.
// X1 (round 4): the VP_FP16 engines' form -- one fp16 plane per tensor, a step covers THIRTY-TWO input channels and the two LDS planes are its two
// 16-channel halves (plane 0 = channels [32 s, 32 s + 16), plane 1 = [32 s + 16, 32 s + 32); weights packed likewise), two MFMAs per fragment pair
// (a0 . b0 + a1 . b1) instead of three: the same LDS plan, DMA plan and barrier per step (kernels_conv3x3_x3.hip does the same with its chunks).
template <class G, int NFW, int ABL, bool X1>
__device__ __forceinline__ void conv3x3_map_body(const ConvGemmParams& p, const int wave) {
  constexpr int RH = G::RH, RW = G::RW, RPX = G::RPX, NWV = G::NWV, HW = G::HW, HPX = G::HPX, H_PLANE = G::H_PLANE, W_PLANE = G::W_PLANE, H_BUF = G::H_BUF,
                W_BUF = G::W_BUF, NHI = G::NHI, NWI = G::NWI, NINSTR = G::NINSTR, IPW = G::IPW;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* const halo0 = smem;                 // [2 buffers][2 planes][H_PLANE]
  char* const wgt0 = smem + 2 * H_BUF;      // [2 buffers][2 planes][W_PLANE]

  const int tid = threadIdx.x, lane = tid & 63;
  const int regions_x = p.W / RW, n_regions = regions_x * (p.H / RH), n_co = p.CoutW >> 5;
  int vid;  // XCD-aware map: the workgroups that share a weight slab (the regions of one channel tile and K slice) are consecutive
  {
    const int nwg = gridDim.x, q = nwg >> 3, r = nwg & 7, xcd = blockIdx.x & 7;
    vid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (blockIdx.x >> 3);
  }
  const int region = vid % n_regions, rest = vid / n_regions;
  const int tile_co = rest % n_co, zsplit = rest / n_co;
  const int ry0 = (region / regions_x) * RH, rx0 = (region % regions_x) * RW;
  constexpr int CS = X1 ? 32 : 16;  // input channels per step
  const int KS_all = p.Cin / CS;
  const half_t* const in_p1 = X1 ? p.in_hi + 16 : p.in_lo;  // second plane of the halo image
  const int s_first = (int)(((long long)KS_all * zsplit) / p.nsplit);
  const int KS = (int)(((long long)KS_all * (zsplit + 1)) / p.nsplit) - s_first;

  // ---- DMA plan of this wave: instruction ii = wave + 8 i of a step.  [0, 2 NHI): halo, plane ii / NHI, rows 32 g .. 32 g + 31 (lane l:
  // row 32 g + (l >> 1), stored slot l & 1 = logical slot (l & 1) ^ bit 3 of the row); [2 NHI, NINSTR): weights, linear (host-packed image).
  // Everything that depends on the instruction's kind is resolved HERE, once: per instruction a source pointer and a per-step element stride
  // for this lane (0 for a lane that reads the zero page), a wave-uniform LDS offset and buffer stride -- the loop then issues IPW
  // branch-free DMA instructions (the last one under a single scalar test: waves 4..7 have nine).
  const half_t* d_src[IPW];
  int d_step[IPW], d_dst[IPW], d_buf[IPW];
  const int n_dma = (NINSTR - wave + NWV - 1) / NWV;  // instructions of this wave
#pragma unroll
  for (int i = 0; i < IPW; ++i) {
    const int ii = wave + NWV * i;
    if (ii < 2 * NHI) {
      const int pl = ii >= NHI ? 1 : 0, g = ii - pl * NHI;
      const int R = 32 * g + (lane >> 1), lslot = (lane & 1) ^ ((R >> 3) & 1);
      const int hy = R / HW, hx = R - hy * HW;
      const int gy = ry0 - 1 + hy, gx = rx0 - 1 + hx;
      const bool ok = R < HPX && (unsigned)gy < (unsigned)p.H && (unsigned)gx < (unsigned)p.W;
      d_src[i] = ok ? (pl ? in_p1 : p.in_hi) + ((size_t)(gy * p.W + gx) * p.Cin + lslot * 8 + s_first * CS) : p.zeros;
      d_step[i] = ok ? CS : 0;
      d_dst[i] = pl * H_PLANE + g * 1024;
      d_buf[i] = H_BUF;
    } else {
      const int wi = (ii < NINSTR ? ii : NINSTR - 1) - 2 * NHI, pl = wi >= NWI ? 1 : 0, pc = wi - pl * NWI;
      d_src[i] = (pl ? p.w_lo : p.w_hi) + (((size_t)tile_co * KS_all + s_first) * (W_PLANE / 2) + pc * 512 + lane * 8);
      d_step[i] = W_PLANE / 2;
      d_dst[i] = 2 * H_BUF + pl * W_PLANE + pc * 1024;
      d_buf[i] = W_BUF;
    }
  }
#define VP_MAP_DMA(BUF, STEP)                                                                                         \
  {                                                                                                                   \
    _Pragma("unroll") for (int i = 0; i < IPW; ++i) {                                                                 \
      if (i < IPW - 1 || n_dma == IPW) VP_GLOBAL_LOAD_LDS16(d_src[i] + (size_t)(STEP) * d_step[i], smem + d_dst[i] + (BUF) * d_buf[i]); \
    }                                                                                                                 \
  }

  // ---- fragment addressing.  Pixel tile f = wave + 8 j holds region pixels 32 f .. 32 f + 31; lane's pixel -> halo row of tap (0, 0)
  int b_row[NFW];
#pragma unroll
  for (int j = 0; j < NFW; ++j) {
    const int pix = min((wave + NWV * j) * 32 + (lane & 31), RPX - 1);   // ragged last tile: its surplus lanes repeat the last pixel (never stored)
    const int y = pix / RW, x = pix - y * RW;
    b_row[j] = y * HW + x;
  }
  const int ks = lane >> 5;                                                          // which 8 of the step's 16 channels this lane feeds
  const int a_ofs = (lane & 31) * 32 + ((ks ^ (((lane & 31) >> 3) & 1)) << 4);      // weight row tap * 32 + (lane & 31): bit 3 of the row = bit 3 of the lane's channel

  f32x16_t acc[NFW];
#pragma unroll
  for (int j = 0; j < NFW; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[j][r] = 0.0f;

  VP_MAP_DMA(0, 0)
  for (int s = 0; s < KS; ++s) {
    const int buf = s & 1;
    if constexpr (!(ABL & 8)) {
      VP_WAIT_VMCNT(0);   // this wave's pieces of step s have landed ...
      VP_LDS_BARRIER();   // ... everybody's have, and everybody is done reading the other buffer (step s - 1)
    }
    if constexpr (!(ABL & 1)) {
      if (s + 1 < KS) VP_MAP_DMA(buf ^ 1, s + 1)
    }
    const char* hb = halo0 + buf * H_BUF;
    const char* wb = wgt0 + buf * W_BUF;
    // Fragments are read ONE TAP AHEAD into a second register set (issued before the current tap's MFMAs, sched_barrier keeps them
    // there): with the reads right in front of their MFMAs the two waves of a SIMD ran "read -> wait -> multiply" back to back and LDS
    // latency added to the matrix time instead of hiding under it (tools/map_ablate.hip: fragment reads alone 26 us of decode_layer_2's
    // 74, MFMAs alone 32, together 58).
    h8_t fa_hi[2], fa_lo[2], fb_hi[2][NFW], fb_lo[2][NFW];
#define VP_MAP_READ(SET, T)                                                                                          \
  {                                                                                                                  \
    constexpr int tofs_ = ((T) / 3) * HW + ((T) % 3);                                                                \
    fa_hi[SET] = *reinterpret_cast<const h8_t*>(wb + (T) * 1024 + a_ofs);                                            \
    fa_lo[SET] = *reinterpret_cast<const h8_t*>(wb + W_PLANE + (T) * 1024 + a_ofs);                                  \
    _Pragma("unroll") for (int j = 0; j < NFW; ++j) {                                                                \
      const int R_ = (ABL & 4) ? b_row[j] : b_row[j] + tofs_;                                                        \
      const int o_ = R_ * 32 + ((ks ^ ((R_ >> 3) & 1)) << 4);                                                        \
      fb_hi[SET][j] = *reinterpret_cast<const h8_t*>(hb + o_);                                                       \
      fb_lo[SET][j] = *reinterpret_cast<const h8_t*>(hb + H_PLANE + o_);                                             \
    }                                                                                                                \
  }
#define VP_MAP_MFMA(SET)                                                                                             \
  _Pragma("unroll") for (int j = 0; j < NFW; ++j) {                                                                  \
    if constexpr ((ABL & 2) != 0) {                                                                                  \
      acc[j][0] += (float)fa_lo[SET][0] + (float)fa_hi[SET][1] + (float)fb_hi[SET][j][2] + (float)fb_lo[SET][j][3];  \
    } else if constexpr (X1) { /* the planes are the step's two K halves */                                          \
      acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa_hi[SET], fb_hi[SET][j], acc[j], 0, 0, 0);                   \
      acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa_lo[SET], fb_lo[SET][j], acc[j], 0, 0, 0);                   \
    } else {                                                                                                         \
      acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa_lo[SET], fb_hi[SET][j], acc[j], 0, 0, 0);                   \
      acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa_hi[SET], fb_lo[SET][j], acc[j], 0, 0, 0);                   \
      acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa_hi[SET], fb_hi[SET][j], acc[j], 0, 0, 0);                   \
    }                                                                                                                \
  }
#define VP_MAP_TAP(T)                                                                                                \
  {                                                                                                                  \
    if constexpr ((T) < 8) {                                                                                         \
      VP_MAP_READ(((T) + 1) & 1, (T) + 1)                                                                            \
      /* this tap's fragments (read one tap ago) must have landed; the 2 + 2 NFW reads just issued stay in flight.  Left to itself the  */ \
      /* compiler puts s_waitcnt lgkmcnt(0) here on every other tap: the prefetch was drained before the MFMAs it should hide under     */ \
      if constexpr (!(ABL & 16)) VP_WAIT_LGKMCNT(2 + 2 * NFW);                                                       \
    }                                                                                                                \
    __builtin_amdgcn_sched_barrier(0);                                                                               \
    VP_MAP_MFMA((T) & 1)                                                                                             \
    __builtin_amdgcn_sched_barrier(0);                                                                               \
  }
    VP_MAP_READ(0, 0)
    VP_MAP_TAP(0) VP_MAP_TAP(1) VP_MAP_TAP(2) VP_MAP_TAP(3) VP_MAP_TAP(4) VP_MAP_TAP(5) VP_MAP_TAP(6) VP_MAP_TAP(7) VP_MAP_TAP(8)
#undef VP_MAP_TAP
#undef VP_MAP_MFMA
#undef VP_MAP_READ
  }
#undef VP_MAP_DMA

  // ---- fp32 partial sums straight from the accumulators: lane holds channels 8 g + 4 (lane >> 5) + r of its pixel
  const int M = p.H * p.W, co0 = tile_co * 32;
#pragma unroll
  for (int j = 0; j < NFW; ++j) {
    const int pix = (wave + NWV * j) * 32 + (lane & 31);
    if (RPX % 32 != 0 && pix >= RPX) continue;
    const int y = pix / RW, x = pix - y * RW;
    const int m = (ry0 + y) * p.W + rx0 + x;
    float* row = p.partial + ((size_t)zsplit * M + m) * p.CoutW + co0 + 4 * (lane >> 5);
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const f32x4_t v = {acc[j][4 * g + 0], acc[j][4 * g + 1], acc[j][4 * g + 2], acc[j][4 * g + 3]};
      *reinterpret_cast<f32x4_t*>(row + 8 * g) = v;
    }
  }
}

// ABL: ablation bits for tools/map_ablate.hip only (1 = no DMA inside the loop, 2 = no MFMA, 4 = tap-invariant fragment addresses, 8 = no barrier / vmcnt
// wait in the loop); 0 in the library.
template <int ABL = 0, bool X1 = false>
__global__ __launch_bounds__(512, 2) void conv3x3_map_kernel(const ConvGemmParams p) {
  using G = mapk::Neck;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));  // wave-uniform by construction: keep it scalar
  // wave 0 carries pixel tiles 0, 8, 16, 24; waves 1..7 three each
  if (wave == 0) conv3x3_map_body<G, 4, ABL, X1>(p, wave);
  else conv3x3_map_body<G, 3, ABL, X1>(p, wave);
}
// the context block's 10x20 maps: four waves, pixel tiles {0, 4}, {1, 5}, {2, 6}, {3}
template <int ABL = 0, bool X1 = false>
__global__ __launch_bounds__(256, 2) void conv3x3_map_ctx_kernel(const ConvGemmParams p) {
  using G = mapk::Ctx;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  if (wave < 3) conv3x3_map_body<G, 2, ABL, X1>(p, wave);
  else conv3x3_map_body<G, 1, ABL, X1>(p, wave);
}

// weight element (output channel co, input channel ci, tap t) -> index into the packed tensor: [co / 32][ci / 16][plane block of 9 x 32 rows x
// 32 B in LDS image order]; cin_pad = padded input channels
size_t conv3x3_map_pack_index(int co, int ci, int t, int cin_pad) {
  const int col = co & 31, k16 = ci & 15;
  const int slot = (k16 >> 3) ^ ((col >> 3) & 1);
  return (((size_t)(co >> 5) * (cin_pad >> 4) + (ci >> 4)) * (9 * 32) + (size_t)t * 32 + col) * 16 + slot * 8 + (k16 & 7);
}
// the fp16 engines' form (X1): steps of 32 input channels, the element's plane = (ci >> 4) & 1 (the caller's two plane arrays), index inside it:
size_t conv3x3_map_pack_index_k32(int co, int ci, int t, int cin_pad) {
  const int col = co & 31, k16 = ci & 15;
  const int slot = (k16 >> 3) ^ ((col >> 3) & 1);
  return (((size_t)(co >> 5) * (cin_pad >> 5) + (ci >> 5)) * (9 * 32) + (size_t)t * 32 + col) * 16 + slot * 8 + (k16 & 7);
}

// geometry a map takes: 1 = 20x40 regions (neck), 2 = 10x20 regions (context block; only where 20x40 does not tile), 0 = neither
int conv3x3_map_geometry(int H, int W) {
  if (H >= 20 && W >= 40 && H % 20 == 0 && W % 40 == 0) return 1;
  if (H >= 10 && W >= 20 && H % 10 == 0 && W % 20 == 0) return 2;
  return 0;
}
bool conv3x3_map_shape_ok(int H, int W, int cin_pad, int coutw) {
  return conv3x3_map_geometry(H, W) != 0 && cin_pad % 16 == 0 && cin_pad >= 32 && coutw % 32 == 0;
}

// ------------------------------------------------------------------------------------------------------------------------------------------
// Round 5, halo tile 12 ("map2"): the same kernel on a 64-CHANNEL weight slab -- every wave carries TWO 32-channel M tiles.
//
// Why (profiles/r04_map_ablate_ab.txt): with one M tile per wave the loop is LDS-fragment-read-bound -- per tap a wave reads 2 + 2 NFW fragments
// for 3 NFW MFMAs (0.89 ds_read_b128 per MFMA at NFW = 3; the kernel without its MFMAs still takes 41 of 48 us on decode_layer_0).  With two M tiles
// every pixel fragment feeds two MFMA triples: 4 + 2 NFW reads for 6 NFW MFMAs (0.56).  What it costs, and how it is paid:
//   * LDS: a step's weights double (9 x 64 rows x 32 B x (hi, lo) = 36 KB) and no longer fit twice beside the double-buffered halo image
//     (2 x 59 KB): they are SINGLE-buffered in two tap groups -- A = taps 0..4, B = taps 5..8 -- refilled by LDS-DMA under the other group's
//     MFMAs, with two LDS-only barriers per step: #1 at the step's head publishes halo[s] + A[s] and retires B[s-1] (then B[s] and halo[s+1]
//     are requested), #2 at the head of tap 4 -- behind the last issue of a group-A read, in front of the first group-B read -- publishes B[s]
//     and retires A[s] (then A[s+1] is requested).  Every request has four taps (> 2000 cycles per wave) to land.  155 648 B, as tile 11.
//   * registers: 6 accumulator tiles per wave (+ one for waves 0 / 1, below) leave no room for two full fragment sets, so the prefetch ROLLS:
//     while pixel tile j multiplies, tile j + 1's two fragments (or the next tap's first) are in flight into the other of two small sets, and the
//     next tap's four weight fragments are requested during the tap's second tile: 12 fragment registers sets of 4 instead of 24.
//   * balance: 25 pixel tiles over 8 waves left wave 0 with four tiles against three (a quarter of every step's time at the barrier for the
//     other seven).  Here tile 24 is SPLIT BY M TILE between waves 0 and 1 (XTRA = 0 / 1: three MFMAs per tap more each): 13 / 13 / 12 / 12
//     MFMA units per SIMD instead of 14 / 12 / 12 / 12.
//   * workgroups: a slab is twice the work, so the same K split gives HALF the workgroups (120-128 per layer): one neck layer no longer fills the
//     chip by itself -- the two networks of the metric configuration (forked heads) or the other cameras in flight do -- and the fp32 slabs the
//     finish kernel re-reads do not grow.  Summation order per output: as tile 11 (channels in steps of 16, taps inside) -- bit-identical to it for
//     equal K slices.
// Parity mode only (fp16 engines keep tile 11's X1 form), 20x40 regions only.
template <int NFW, int XTRA>
__device__ __forceinline__ void conv3x3_map2_body(const ConvGemmParams& p, const int wave) {
  using G = mapk::Neck;
  constexpr int RH = G::RH, RW = G::RW, RPX = G::RPX, NWV = G::NWV, HW = G::HW, HPX = G::HPX, H_PLANE = G::H_PLANE, H_BUF = G::H_BUF, NHI = G::NHI;
  constexpr int W_PLANE = 9 * 64 * 32;          // bytes of one weight plane of a step (18 432): [tap][64 rows][32 B]
  constexpr int NWP = W_PLANE / 1024;           // 1 KiB DMA pieces per plane (18): taps 0..4 = pieces 0..9 (group A), taps 5..8 = 10..17 (group B)
  constexpr int NPA = 10, NPB = NWP - NPA;
  static_assert(2 * H_BUF + 2 * W_PLANE <= 160 * 1024 && NFW == 3 && G::NF == 25, "LDS / tile plan");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* const halo0 = smem;                 // [2 buffers][2 planes][H_PLANE]
  char* const wgt0 = smem + 2 * H_BUF;      // [2 planes][W_PLANE], single-buffered

  const int tid = threadIdx.x, lane = tid & 63;
  const int regions_x = p.W / RW, n_regions = regions_x * (p.H / RH), n_co = p.CoutW >> 6;
  int vid;
  {
    const int nwg = gridDim.x, q = nwg >> 3, r = nwg & 7, xcd = blockIdx.x & 7;
    vid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (blockIdx.x >> 3);
  }
  const int region = vid % n_regions, rest = vid / n_regions;
  const int tile_co = rest % n_co, zsplit = rest / n_co;
  const int ry0 = (region / regions_x) * RH, rx0 = (region % regions_x) * RW;
  const int KS_all = p.Cin / 16;
  const int s_first = (int)(((long long)KS_all * zsplit) / p.nsplit);
  const int KS = (int)(((long long)KS_all * (zsplit + 1)) / p.nsplit) - s_first;

  // ---- DMA plan.  Halo: instruction ii = wave + 8 i of [0, 2 NHI) (58: waves 0, 1 issue eight, the others seven), exactly tile 11's.
  constexpr int IPH = (2 * NHI + NWV - 1) / NWV;
  const half_t* h_src[IPH];
  int h_step[IPH], h_dst[IPH];
  const int n_h = (2 * NHI - wave + NWV - 1) / NWV;
#pragma unroll
  for (int i = 0; i < IPH; ++i) {
    const int ii = min(wave + NWV * i, 2 * NHI - 1);
    const int pl = ii >= NHI ? 1 : 0, g = ii - pl * NHI;
    const int R = 32 * g + (lane >> 1), lslot = (lane & 1) ^ ((R >> 3) & 1);
    const int hy = R / HW, hx = R - hy * HW;
    const int gy = ry0 - 1 + hy, gx = rx0 - 1 + hx;
    const bool ok = R < HPX && (unsigned)gy < (unsigned)p.H && (unsigned)gx < (unsigned)p.W;
    h_src[i] = ok ? (pl ? p.in_lo : p.in_hi) + ((size_t)(gy * p.W + gx) * p.Cin + lslot * 8 + s_first * 16) : p.zeros;
    h_step[i] = ok ? 16 : 0;
    h_dst[i] = pl * H_PLANE + g * 1024;
  }
  // Weights: linear copies of the host-packed LDS image.  Group A = 2 x 10 pieces: jj = ((wave + 4) & 7) + 8 i < 20 (the waves with seven halo
  // pieces take three); group B = 2 x 8 pieces: two per wave.
  const size_t w_lane = ((size_t)tile_co * KS_all + s_first) * (W_PLANE / 2) + lane * 8;   // elements; + step * (W_PLANE / 2) + piece * 512
#define VP_MAP2_DMA_H(BUF, STEP)                                                                                     \
  {                                                                                                                  \
    _Pragma("unroll") for (int i = 0; i < IPH; ++i) {                                                                \
      if (i < IPH - 1 || n_h == IPH) VP_GLOBAL_LOAD_LDS16(h_src[i] + (size_t)(STEP) * h_step[i], halo0 + (BUF) * H_BUF + h_dst[i]); \
    }                                                                                                                \
  }
#define VP_MAP2_DMA_WPIECE(JJ, NP, P0, STEP)                                                                         \
  {                                                                                                                  \
    const int jj_ = (JJ), pl_ = jj_ >= (NP) ? 1 : 0, pc_ = (P0) + jj_ - pl_ * (NP);                                   \
    VP_GLOBAL_LOAD_LDS16((pl_ ? p.w_lo : p.w_hi) + (w_lane + (size_t)(STEP) * (W_PLANE / 2) + pc_ * 512), wgt0 + pl_ * W_PLANE + pc_ * 1024); \
  }
#define VP_MAP2_DMA_A(STEP)                                                                                          \
  {                                                                                                                  \
    const int j0_ = (wave + 4) & 7;                                                                                  \
    VP_MAP2_DMA_WPIECE(j0_, NPA, 0, STEP)                                                                            \
    VP_MAP2_DMA_WPIECE(j0_ + 8, NPA, 0, STEP)                                                                        \
    if (j0_ + 16 < 2 * NPA) VP_MAP2_DMA_WPIECE(j0_ + 16, NPA, 0, STEP)                                               \
  }
#define VP_MAP2_DMA_B(STEP)                                                                                          \
  {                                                                                                                  \
    VP_MAP2_DMA_WPIECE(wave, NPB, NPA, STEP)                                                                         \
    VP_MAP2_DMA_WPIECE(wave + 8, NPB, NPA, STEP)                                                                     \
  }

  // ---- fragment addressing: pixel tiles wave, wave + 8, wave + 16 (+ tile 24 for XTRA >= 0)
  constexpr int NB = NFW + (XTRA >= 0 ? 1 : 0);
  int b_row[NB];
#pragma unroll
  for (int j = 0; j < NB; ++j) {
    const int pix = min((j < NFW ? wave + NWV * j : 24) * 32 + (lane & 31), RPX - 1);
    const int y = pix / RW, x = pix - y * RW;
    b_row[j] = y * HW + x;
  }
  const int ks = lane >> 5;
  const int a_ofs = (lane & 31) * 32 + ((ks ^ (((lane & 31) >> 3) & 1)) << 4);   // M tile i: + i * 1024; tap t: + t * 2048; lo plane: + W_PLANE

  f32x16_t acc[2][NFW], accx;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    accx[r] = 0.0f;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < NFW; ++j) acc[i][j][r] = 0.0f;
  }

  VP_MAP2_DMA_H(0, 0)
  VP_MAP2_DMA_A(0)
  for (int s = 0; s < KS; ++s) {
    const int buf = s & 1;
    VP_WAIT_VMCNT(0);   // halo[s] and A[s] of this wave have landed ...
    VP_LDS_BARRIER();   // ... everybody's have; everybody is done with B[s - 1] and halo[s - 1]
    VP_MAP2_DMA_B(s)
    if (s + 1 < KS) VP_MAP2_DMA_H(buf ^ 1, s + 1)
    const char* hb = halo0 + buf * H_BUF;
    // rolling prefetch: A fragments in two sets by tap parity, B fragments in two sets by slot parity (slot = NB * tap + j)
    h8_t fa_hi[2][2], fa_lo[2][2], fb_hi[2], fb_lo[2];
#define VP_MAP2_READ_A(T)                                                                                            \
  {                                                                                                                  \
    _Pragma("unroll") for (int i = 0; i < 2; ++i) {                                                                  \
      fa_hi[(T) & 1][i] = *reinterpret_cast<const h8_t*>(wgt0 + (T) * 2048 + i * 1024 + a_ofs);                      \
      fa_lo[(T) & 1][i] = *reinterpret_cast<const h8_t*>(wgt0 + W_PLANE + (T) * 2048 + i * 1024 + a_ofs);           \
    }                                                                                                                \
  }
#define VP_MAP2_READ_B(SLOT)                                                                                         \
  {                                                                                                                  \
    constexpr int t_ = (SLOT) / NB, j_ = (SLOT) % NB;                                                                \
    constexpr int tofs_ = (t_ / 3) * HW + (t_ % 3);                                                                  \
    const int R_ = b_row[j_] + tofs_;                                                                                \
    const int o_ = R_ * 32 + ((ks ^ ((R_ >> 3) & 1)) << 4);                                                          \
    fb_hi[(SLOT) & 1] = *reinterpret_cast<const h8_t*>(hb + o_);                                                     \
    fb_lo[(SLOT) & 1] = *reinterpret_cast<const h8_t*>(hb + H_PLANE + o_);                                           \
  }
#define VP_MAP2_MFMA3(ACC, T, I, SLOT)                                                                               \
  {                                                                                                                  \
    ACC = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa_lo[(T) & 1][I], fb_hi[(SLOT) & 1], ACC, 0, 0, 0);                \
    ACC = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa_hi[(T) & 1][I], fb_lo[(SLOT) & 1], ACC, 0, 0, 0);                \
    ACC = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa_hi[(T) & 1][I], fb_hi[(SLOT) & 1], ACC, 0, 0, 0);                \
  }
    // one slot = one pixel tile of one tap.  Reads issued in a slot: the NEXT slot's pixel fragments (2), and in slot j == 1 of taps 0..7 the
    // next tap's four weight fragments (in front of them) -- except tap 4's, whose next tap reads group B: barrier #2 sits at the head of tap 4
    // and the read of A(5) comes right behind it.  Before the MFMAs: everything but what this slot issued must have landed.
#define VP_MAP2_SLOT(T, J)                                                                                           \
  {                                                                                                                  \
    constexpr int slot_ = (T) * NB + (J);                                                                            \
    constexpr bool rdA_ = (J) == 1 && (T) < 8;                                                                       \
    constexpr bool rdB_ = slot_ + 1 < 9 * NB;                                                                        \
    if constexpr ((J) == 0 && (T) == 4) {                                                                            \
      VP_WAIT_VMCNT(0);     /* B[s] (and halo[s + 1]) of this wave have landed ... */                                 \
      VP_LDS_BARRIER();     /* ... everybody's; every read of group A (taps 0..4, the last issued in tap 3) is done   */ \
      if (s + 1 < KS) VP_MAP2_DMA_A(s + 1)                                                                           \
    }                                                                                                                \
    if constexpr (rdA_) VP_MAP2_READ_A((T) + 1)                                                                      \
    if constexpr (rdB_) VP_MAP2_READ_B(slot_ + 1)                                                                    \
    VP_WAIT_LGKMCNT((rdA_ ? 4 : 0) + (rdB_ ? 2 : 0));                                                                \
    __builtin_amdgcn_sched_barrier(0);                                                                               \
    if constexpr ((J) < NFW) {                                                                                       \
      VP_MAP2_MFMA3(acc[0][(J) < NFW ? (J) : 0], T, 0, slot_)                                                        \
      VP_MAP2_MFMA3(acc[1][(J) < NFW ? (J) : 0], T, 1, slot_)                                                        \
    } else {                                                                                                         \
      VP_MAP2_MFMA3(accx, T, (XTRA >= 0 ? XTRA : 0), slot_)                                                          \
    }                                                                                                                \
    __builtin_amdgcn_sched_barrier(0);                                                                               \
  }
#define VP_MAP2_TAP(T)                                                                                               \
  {                                                                                                                  \
    VP_MAP2_SLOT(T, 0) VP_MAP2_SLOT(T, 1) VP_MAP2_SLOT(T, 2)                                                         \
    if constexpr (NB > 3) VP_MAP2_SLOT(T, 3)                                                                         \
  }
    VP_MAP2_READ_A(0)
    VP_MAP2_READ_B(0)
    VP_MAP2_TAP(0) VP_MAP2_TAP(1) VP_MAP2_TAP(2) VP_MAP2_TAP(3) VP_MAP2_TAP(4) VP_MAP2_TAP(5) VP_MAP2_TAP(6) VP_MAP2_TAP(7) VP_MAP2_TAP(8)
#undef VP_MAP2_TAP
#undef VP_MAP2_SLOT
#undef VP_MAP2_MFMA3
#undef VP_MAP2_READ_B
#undef VP_MAP2_READ_A
  }
#undef VP_MAP2_DMA_B
#undef VP_MAP2_DMA_A
#undef VP_MAP2_DMA_WPIECE
#undef VP_MAP2_DMA_H

  // ---- fp32 partial sums straight from the accumulators (tile 11's layout: p.partial[z][pixel][CoutW])
  const int M = p.H * p.W, co0 = tile_co * 64;
#pragma unroll
  for (int j = 0; j < NB; ++j) {
    const int pix = (j < NFW ? wave + NWV * j : 24) * 32 + (lane & 31);
    const int y = pix / RW, x = pix - y * RW;
    const int m = (ry0 + y) * p.W + rx0 + x;
    float* row = p.partial + ((size_t)zsplit * M + m) * p.CoutW + co0 + 4 * (lane >> 5);
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      if (j >= NFW && i != XTRA) continue;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const f32x16_t& a = j < NFW ? acc[i][j < NFW ? j : 0] : accx;
        const f32x4_t v = {a[4 * g + 0], a[4 * g + 1], a[4 * g + 2], a[4 * g + 3]};
        *reinterpret_cast<f32x4_t*>(row + i * 32 + 8 * g) = v;
      }
    }
  }
}

__global__ __launch_bounds__(512, 2) void conv3x3_map2_kernel(const ConvGemmParams p) {
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  // pixel tile 24 is split by M tile between waves 0 and 1
  if (wave == 0) conv3x3_map2_body<3, 0>(p, wave);
  else if (wave == 1) conv3x3_map2_body<3, 1>(p, wave);
  else conv3x3_map2_body<3, -1>(p, wave);
}

// packed weights of tile 12: [co / 64][ci / 16][plane block of 9 taps x 64 rows x 32 B in LDS image order]
size_t conv3x3_map2_pack_index(int co, int ci, int t, int cin_pad) {
  const int col = co & 63, k16 = ci & 15;
  const int slot = (k16 >> 3) ^ ((col >> 3) & 1);
  return (((size_t)(co >> 6) * (cin_pad >> 4) + (ci >> 4)) * (9 * 64) + (size_t)t * 64 + col) * 16 + slot * 8 + (k16 & 7);
}
bool conv3x3_map2_shape_ok(int H, int W, int cin_pad, int coutw) {
  return conv3x3_map_geometry(H, W) == 1 && cin_pad % 16 == 0 && cin_pad >= 32 && coutw % 64 == 0;
}
bool conv3x3_map2_supported(const ConvGemmParams& p) {
  return p.ks == 3 && p.stride <= 1 && p.in_hi && p.in_lo && p.w_hi && p.w_lo && p.Cin2 == 0 && p.partial != nullptr && p.zeros != nullptr && p.nsplit >= 1 &&
         p.nsplit <= p.Cin / 16 && conv3x3_map2_shape_ok(p.H, p.W, p.Cin, p.CoutW);
}
hipError_t launch_conv3x3_map2(const ConvGemmParams& p, hipStream_t st) {
  if (!conv3x3_map2_supported(p)) return hipErrorInvalidValue;
  using G = mapk::Neck;
  constexpr int lds = 2 * G::H_BUF + 2 * 9 * 64 * 32;
  static LdsAttrOnce once;
  if (hipError_t e = set_max_dynamic_lds(once, reinterpret_cast<const void*>(conv3x3_map2_kernel), lds); e != hipSuccess) return e;
  const int n_regions = (p.H / G::RH) * (p.W / G::RW);
  hipLaunchKernelGGL(conv3x3_map2_kernel, dim3(n_regions * (p.CoutW / 64) * p.nsplit), dim3(512), lds, st, p);
  if (hipError_t e = hipGetLastError(); e != hipSuccess) return e;
  return launch_splitk_finish(p, st);
}

bool conv3x3_map_supported(const ConvGemmParams& p) {
  const bool x1 = p.in_lo == nullptr;   // VP_FP16 engines: steps of 32 channels, the step's halves in the two planes
  return p.ks == 3 && p.stride <= 1 && p.in_hi && p.w_hi && p.w_lo && p.Cin2 == 0 && p.partial != nullptr && p.zeros != nullptr && p.nsplit >= 1 &&
         p.Cin % (x1 ? 32 : 16) == 0 && p.nsplit <= p.Cin / (x1 ? 32 : 16) && conv3x3_map_shape_ok(p.H, p.W, p.Cin, p.CoutW);
}

template <bool X1>
static hipError_t launch_map_cfg(const ConvGemmParams& p, hipStream_t st) {
  if (conv3x3_map_geometry(p.H, p.W) == 1) {
    using G = mapk::Neck;
    static LdsAttrOnce once;
    if (hipError_t e = set_max_dynamic_lds(once, reinterpret_cast<const void*>(conv3x3_map_kernel<0, X1>), G::LDS); e != hipSuccess) return e;
    const int n_regions = (p.H / G::RH) * (p.W / G::RW);
    hipLaunchKernelGGL((conv3x3_map_kernel<0, X1>), dim3(n_regions * (p.CoutW / 32) * p.nsplit), dim3(64 * G::NWV), G::LDS, st, p);
  } else {
    using G = mapk::Ctx;
    static LdsAttrOnce once;
    if (hipError_t e = set_max_dynamic_lds(once, reinterpret_cast<const void*>(conv3x3_map_ctx_kernel<0, X1>), G::LDS); e != hipSuccess) return e;
    const int n_regions = (p.H / G::RH) * (p.W / G::RW);
    hipLaunchKernelGGL((conv3x3_map_ctx_kernel<0, X1>), dim3(n_regions * (p.CoutW / 32) * p.nsplit), dim3(64 * G::NWV), G::LDS, st, p);
  }
  return hipSuccess;
}

hipError_t launch_conv3x3_map(const ConvGemmParams& p, hipStream_t st) {
  if (!conv3x3_map_supported(p)) return hipErrorInvalidValue;
  if (hipError_t e = p.in_lo ? launch_map_cfg<false>(p, st) : launch_map_cfg<true>(p, st); e != hipSuccess) return e;
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return e;
  return launch_splitk_finish(p, st);  // also for nsplit == 1: bias / activation / (hi, lo) split live there
}

}  // namespace vp
