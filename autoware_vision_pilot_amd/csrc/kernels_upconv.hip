// Composed up-sampling stage of the PARITY mode (round 6): ConvTranspose2d(k2, s2) [+ Conv1x1(skip)] -> Conv3x3 (+ GELU) in ONE launch.
//
// The reference's decoders apply `upsample_layer_k` (+ `skip_link_layer_k`) and then `decode_layer_2k` with no nonlinearity in between
// (scene_neck.py:29-35,41-46,52-57; scene_seg_head.py:24-29,35-38; scene_3d_head.py:26-31,38-41), so the pair is one linear map.  Written from the
// LOW-resolution tensor x (H x W) it is, for the output phase (py, px) = (Y & 1, X & 1) of the high-resolution pixel (Y, X) = (2y + py, 2x + px):
//     out[co, Y, X] = act( bias[class(Y), class(X)][co]
//                        + sum_{a, b in {0, 1}} sum_ci  x[ci, y + py - 1 + a, x + px - 1 + b] * Wx[py, px][a, b][co, ci]     (2x2 conv per phase)
//                        + sum_{ty, tx in {0, 1, 2}} sum_cs skip[cs, Y + ty - 1, X + tx - 1] * Ws[ty, tx][co, cs] )           (3x3 conv of the skip)
// with Wx = sum over the intermediate channels of W3 * WT, Ws = W3 * Wskip (engine_upconv.cpp composes them at load, fp64 accumulation) and a bias that
// depends only on which of the nine high-resolution taps lie inside the map (zero padding of the high-resolution tensor maps onto zero padding of
// x: row 2H is low-resolution row H).  0.40-0.51x the multiply-adds of the two launches it replaces, the up-sampled tensor never exists.
//
// Execution shape = kernels_conv3x3_x3.hip's (same LDS plan, fragment prefetch one K sub-step ahead, weight tiles by LDS-DMA three deep, the barrier
// between the two K sub-steps of a step, register epilogue), with the K loop turned into a list of STEPS:
//   * a workgroup owns one PHASE of a low-resolution patch (16x16 or 8x16 pixels = 256 / 128 output pixels of that phase) x 128 output channels;
//     the four phases and the channel tiles of a patch get neighbouring workgroup ids (one XCD: the second finds the patch in that L2);
//   * K = list of 32-channel chunks (kernels.hpp upconv_chunk): Cin / 32 chunks of x with 4 taps each, then the skip tensor seen as four
//     half-resolution images (one per pixel class (qy, qx): a STRIDED view, nothing is copied), Cs / 32 chunks per class with 4 / 2 / 2 / 1 taps.
//     Per chunk the (TH + 2) x 18 halo image (origin: low-resolution pixel (y0 - 1, x0 - 1)) is staged once in LDS; a tap is an LDS address offset
//     ((py + a) * 18 + px + b pixels); pixels the phase never reads are not fetched;
//   * the halo of chunk c + 1 is written to LDS at the chunk's LAST step (HDB: other halo image, at the step's head; else between two barriers behind
//     the step's barrier) and the registers are refilled with chunk c + 2 at once: a halo has as many steps to arrive as its predecessor has taps;
//   * epilogue: fmaf(acc, 2^-prescale[phase][co], bias[row class][column class][co]), exact GELU, (hi, lo) split, pixel-shuffle store: every pixel's 128
//     channels are 256 contiguous bytes per plane; split-K slices (the small maps of the neck) leave fp32 partials for upconv_finish_kernel.
#include <algorithm>
#include <cstdlib>

#include "conv_epilogue.hpp"
#include "lds_dma.hpp"

namespace vp {

namespace {

// local pixel q of a phase patch -> linear pixel of the HIGH-resolution map, or -1 outside; lane -> pixel map of lane_to_px16 (conv_epilogue.hpp)
struct PixPhase {
  int y0, x0, H, W, py, px;   // low-resolution patch origin and map size
  __device__ __forceinline__ int operator()(int q) const {
    int rowbit, c;
    lane_to_px16(q & 31, rowbit, c);
    const int y = y0 + 2 * (q >> 5) + rowbit, x = x0 + c;
    return (y < H && x < W) ? (2 * y + py) * (2 * W) + 2 * x + px : -1;
  }
};
// bias class of high-resolution pixel m (row-major, width W2, height H2): (first row ? 0 : last row ? 2 : 1) * 3 + the same for the column
__device__ __forceinline__ int bias_class(int m, int H2, int W2) {
  const int Y = m / W2, X = m - Y * W2;
  return (Y == 0 ? 0 : (Y == H2 - 1 ? 2 : 1)) * 3 + (X == 0 ? 0 : (X == W2 - 1 ? 2 : 1));
}

}  // namespace

// vmcnt wait with a run-time (wave-uniform) count: the immediate forms for the counts the K loop produces
#define VP_WAIT_VMCNT_RT(N)                       \
  do {                                            \
    const int n_ = (N);                           \
    if (n_ >= 16) { VP_WAIT_VMCNT(16); }          \
    else if (n_ >= 14) { VP_WAIT_VMCNT(14); }     \
    else if (n_ >= 12) { VP_WAIT_VMCNT(12); }     \
    else if (n_ >= 10) { VP_WAIT_VMCNT(10); }     \
    else if (n_ >= 8) { VP_WAIT_VMCNT(8); }       \
    else if (n_ >= 6) { VP_WAIT_VMCNT(6); }       \
    else if (n_ >= 4) { VP_WAIT_VMCNT(4); }       \
    else if (n_ >= 2) { VP_WAIT_VMCNT(2); }       \
    else { VP_WAIT_VMCNT(0); }                    \
  } while (0)

// X1 (round 6): the VP_FP16 engines' form, as kernels_conv3x3_x3.hip's: a chunk covers SIXTY-FOUR channels and the two "planes" of every LDS image are its
// 32-channel halves (plane 0 = channels [64c, 64c + 32), plane 1 = [64c + 32, 64c + 64) of the ONE fp16 tensor; w_hi / w_lo carry the halves of the composed
// weights likewise), two MFMAs per fragment pair, one output plane.  The chunk list is upconv_chunk's over HALVED channel counts (Cin / 2, Cs rounded up to
// 64 and halved; ch0 doubled); the last chunk of a skip tensor of 32 or 96 channels (the 20x40, 160x320 and 320x640 stages) fills plane 0 only -- plane 1
// of that chunk is zeroed at the halo store and its weights are zero (half of those steps' MFMAs multiply zeros: up to 18 % of a stage's steps, accepted).
// ABL (tools/upconv_ablate.hip only; 0 in the library): 1 = no halo loads / stores after the prologue, 2 = no MFMA, 16 = no epilogue (the accumulators stay
// alive through a store no launch takes), 32 = epilogue arithmetic without the global stores, 128 = no weight DMA after the prologue
template <int CO_TILE, int TH, int WCO, int WPX, bool HDB, int ACT, bool SPLITK, bool X1 = false, int ABL = 0>
__global__ __launch_bounds__(64 * WCO * WPX, 2) void upconv_x3_kernel(const UpconvParams p) {
  constexpr int NTH = 64 * WCO * WPX;
  constexpr int TW = 16, ROWB = 80, HWD = TW + 2, HPX = (TH + 2) * HWD, PX = TH * TW;
  constexpr int HALO_BYTES = HPX * ROWB, WROW = 64, W_BYTES = CO_TILE * WROW;
  constexpr int HCHUNKS = HPX * 4, HP = (HCHUNKS + NTH - 1) / NTH;
  constexpr int MT = CO_TILE / WCO / 32, NT = PX / WPX / 32;
  constexpr int NHB = HDB ? 2 : 1;
  constexpr int PL = 2;                // planes per tensor: (hi, lo)
  constexpr int HSTRIDE = HALO_BYTES;  // plane stride in LDS
  static_assert(MT >= 1 && NT >= 1 && HP <= 3, "tile shape");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* const halo_base = smem;                               // [NHB][2 planes][HALO_BYTES]
  char* const w_base = smem + NHB * PL * HSTRIDE;             // [3][PL planes][W_BYTES]

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wco = wave / WPX, wpx = wave % WPX;
  const int tiles_x = (p.W + TW - 1) / TW;
  const int n_px_tiles = tiles_x * ((p.H + TH - 1) / TH);
  const int n_co_tiles = p.CoutW / CO_TILE;
  int vid;  // XCD-aware workgroup -> tile map (see kernels_conv3x3.hip)
  {
    const int nwg = gridDim.x, q = nwg >> 3, r = nwg & 7, xcd = blockIdx.x & 7;
    vid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (blockIdx.x >> 3);
  }
  // channel tile fastest, then the phase: the 4 * n_co_tiles workgroups of one patch are neighbours (without split-K; with it the pixel tile is fastest)
  int tile_px, tile_co, phase, zsplit = 0;
  if constexpr (SPLITK) {
    tile_px = vid % n_px_tiles;
    int rest = vid / n_px_tiles;
    tile_co = rest % n_co_tiles;
    rest /= n_co_tiles;
    phase = rest & 3;
    zsplit = rest >> 2;
  } else {
    tile_co = vid % n_co_tiles;
    const int rest = vid / n_co_tiles;
    phase = rest & 3;
    tile_px = rest >> 2;
  }
  const int py = phase >> 1, px = phase & 1;
  const int tyi = tile_px / tiles_x, txi = tile_px - tyi * tiles_x;
  const int y0 = tyi * TH, x0 = txi * TW;
  const int co0 = tile_co * CO_TILE;
  // channel counts the chunk list is built on: X1 chunks are 64 channels wide = two 32-channel halves
  const int CinV = X1 ? p.Cin >> 1 : p.Cin, CsV = X1 ? ((p.Cs + 63) >> 6) << 5 : p.Cs;
  constexpr int CHM = X1 ? 2 : 1;   // real channel offset of a chunk = CHM * its descriptor's ch0
  const half_t* const in_p1 = X1 ? p.in_hi + 32 : p.in_lo;   // second plane of the halo image
  const half_t* const sk_p1 = X1 ? p.sk_hi + 32 : p.sk_lo;
  // X1: the LAST 64-channel chunk of a skip tensor whose (padded) channel count is an odd multiple of 32 has no second half: plane 1 of that chunk is
  // fetched from the first half's address (never past the tensor) and zeroed at the halo store
#define VP_SK_DEAD1(D) (X1 && (D).skip && CHM * (D).ch0 + 32 >= p.Cs)
  const int n_chunks = upconv_chunks(CinV, CsV);
  int cA = 0, cB = n_chunks;
  if constexpr (SPLITK) {
    cA = (int)(((long long)n_chunks * zsplit) / p.nsplit);
    cB = (int)(((long long)n_chunks * (zsplit + 1)) / p.nsplit);
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
    int rowbit, c;
    lane_to_px16(lane & 31, rowbit, c);
    b_ofs0 = ((2 * (wpx * NT) + rowbit) * HWD + c) * ROWB + (lane >> 5) * 16;
  }
  const int a_swz = ((lane & 31) >> 2) & 3;
  const int a_ofs0 = (wco * 32 + (lane & 31)) * WROW + (((lane >> 5) ^ a_swz) << 4);  // K sub-step 1: chunk index ^ 2 -> ^ 32 bytes

  // ---- staging assignment: thread t moves 16-byte piece t + NTH * pc of a halo image (pieces of one thread sit NTH / 4 pixels apart).
  // Element offsets of the piece's pixel in the two sources (-1: outside the map, beyond the image, or a pixel this phase never reads)
  int h_gx[HP], h_gs[HP];
#pragma unroll
  for (int pc = 0; pc < HP; ++pc) {
    const int hidx = tid + NTH * pc;
    const int hp = hidx >> 2, ch = hidx & 3;
    const int hy = hp / HWD, hx = hp - hy * HWD;
    const int gy = y0 - 1 + hy, gx = x0 - 1 + hx;
    const bool ok = hidx < HCHUNKS && (unsigned)gy < (unsigned)p.H && (unsigned)gx < (unsigned)p.W && hy >= py && hy <= py + TH && hx >= px && hx <= px + TW;
    h_gx[pc] = ok ? (gy * p.W + gx) * p.Cin + ch * 8 : -1;
    h_gs[pc] = ok ? (4 * gy * p.W + 2 * gx) * p.Cs + ch * 8 : -1;   // skip pixel (2 gy, 2 gx) of the 2H x 2W map; the class adds (qy * 2W + qx) * Cs
  }
  // weight tiles by LDS-DMA: a step's tile plane is CO_TILE x 64 B = W_BYTES contiguous bytes in global memory, already in LDS image order; wave v
  // copies the 1 KiB pieces v, v + NW, ... of both planes.  Steps of one phase are consecutive tiles.
  const UpconvChunk dA = upconv_chunk(cA, py, px, CinV, CsV);
  const int S = (cB < n_chunks ? upconv_chunk(cB, py, px, CinV, CsV).step0 : upconv_steps(CinV, CsV)) - dA.step0;   // steps of THIS slice
  const size_t w_goff0 = ((size_t)phase * upconv_steps(CinV, CsV) + dA.step0) * w_step + co0 * 32 + wave * 512 + lane * 8;

  f32x16_t acc[MT][NT];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

  // fragment sets [K sub-step parity]: set 0 = channels 0..15 of the step's 32, set 1 = channels 16..31
  h8_t fa[2][MT], fal[2][MT], fb[2][NT], fbl[2][NT];
  u32x4 rh_hi[HP], rh_lo[HP];   // halo staging registers: piece pc of the chunk after next
#pragma unroll
  for (int r = 0; r < HP; ++r) rh_hi[r] = rh_lo[r] = zero4;

  // weight tile of slice step SIDX -> LDS buffer at byte offset WOFS (from w_base), asynchronously
#define VP_DMA_W(WOFS, SIDX)                                                                 \
  if (!(ABL & 128) || (SIDX) < 2) {                                                          \
    const size_t base_ = (size_t)(SIDX) * w_step + w_goff0;                                  \
    char* dst_ = w_base + (WOFS) + wave * 1024;                                              \
    _Pragma("unroll") for (int pc = 0; pc < WPIECES; ++pc) {                                 \
      VP_GLOBAL_LOAD_LDS16(p.w_hi + base_ + pc * NW * 512, dst_ + pc * NW * 1024);           \
      VP_GLOBAL_LOAD_LDS16(p.w_lo + base_ + pc * NW * 512, dst_ + W_BYTES + pc * NW * 1024); \
    }                                                                                        \
  }
  // halo pieces of chunk descriptor D -> registers.  Pieces outside the map (or never read by this phase) fetch element 0 and are ZEROED AT THE STORE:
  // a select here, inside the conditional block that issues the loads, is evaluated in that block -- the compiler put s_waitcnt vmcnt(0) right behind
  // the loads (seen in the ISA: a whole memory round trip exposed at every chunk's last step)
#define VP_LOAD_H(D)                                                                         \
  if (!(ABL & 1) || abl_prologue) {                                                          \
    const half_t* sh_ = (D).skip ? p.sk_hi : p.in_hi;                                        \
    const half_t* sl_ = (D).skip ? (VP_SK_DEAD1(D) ? p.sk_hi : sk_p1) : in_p1;               \
    const int add_ = (D).skip ? ((D).qy * 2 * p.W + (D).qx) * p.Cs + CHM * (D).ch0 : CHM * (D).ch0; \
    _Pragma("unroll") for (int pc = 0; pc < HP; ++pc) {                                      \
      const int g_ = (D).skip ? h_gs[pc] : h_gx[pc];                                         \
      const int o_ = (g_ >= 0 ? g_ + add_ : 0);                                              \
      rh_hi[pc] = *reinterpret_cast<const u32x4*>(sh_ + o_);                                 \
      rh_lo[pc] = *reinterpret_cast<const u32x4*>(sl_ + o_);                                 \
    }                                                                                        \
  }
  // DEAD1: plane 1 of the chunk in the registers does not exist (X1: a chunk of a 32-channel skip tensor): zeros
#define VP_STORE_H(BUF, DEAD1)                                                               \
  if (!(ABL & 1) || abl_prologue)                                                            \
  _Pragma("unroll") for (int pc = 0; pc < HP; ++pc) {                                        \
    if (tid + NTH * pc < HCHUNKS) {                                                          \
      char* dst_ = halo_base + (BUF) * PL * HSTRIDE + h_lds0 + pc * (NTH / 4) * ROWB;        \
      *reinterpret_cast<u32x4*>(dst_) = h_gx[pc] >= 0 ? rh_hi[pc] : zero4;                   \
      *reinterpret_cast<u32x4*>(dst_ + HSTRIDE) = (h_gx[pc] >= 0 && !(DEAD1)) ? rh_lo[pc] : zero4; \
    }                                                                                        \
  }
#define VP_READ_FRAGS(SET, WPTR, HPTR, TAPOFS)                                               \
  {                                                                                          \
    const char* wsrc_ = (WPTR) + (a_ofs0 ^ ((SET) * 32));                                    \
    _Pragma("unroll") for (int i = 0; i < MT; ++i) {                                         \
      fa[SET][i] = *reinterpret_cast<const h8_t*>(wsrc_ + i * WCO * 32 * WROW);              \
      fal[SET][i] = *reinterpret_cast<const h8_t*>(wsrc_ + W_BYTES + i * WCO * 32 * WROW);   \
    }                                                                                        \
    const char* hsrc_ = (HPTR) + b_ofs0 + (TAPOFS) + (SET) * 32;                             \
    _Pragma("unroll") for (int j = 0; j < NT; ++j) {                                         \
      fb[SET][j] = *reinterpret_cast<const h8_t*>(hsrc_ + j * 2 * HWD * ROWB);               \
      fbl[SET][j] = *reinterpret_cast<const h8_t*>(hsrc_ + HSTRIDE + j * 2 * HWD * ROWB);    \
    }                                                                                        \
  }
  // accumulator tiles [Q0, Q1) of the wave (tile q = i * NT + j): hi*lo + lo*hi + hi*hi
#define VP_MFMA_RANGE(SET, Q0, Q1)                                                           \
  _Pragma("unroll") for (int q_ = (Q0); q_ < (Q1); ++q_) {                                   \
    const int i = q_ / NT, j = q_ % NT;                                                      \
    if constexpr ((ABL & 2) != 0) { acc[i][j][0] += (float)fa[SET][i][0] + (float)fal[SET][i][1] + (float)fb[SET][j][2] + (float)fbl[SET][j][3]; continue; } \
    if constexpr (X1) { /* the planes are K halves: a0 . b0 + a1 . b1 */                     \
      acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[SET][i], fb[SET][j], acc[i][j], 0, 0, 0);   \
      acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fal[SET][i], fbl[SET][j], acc[i][j], 0, 0, 0); \
      continue;                                                                              \
    }                                                                                        \
    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fal[SET][i], fb[SET][j], acc[i][j], 0, 0, 0); \
    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[SET][i], fbl[SET][j], acc[i][j], 0, 0, 0); \
    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[SET][i], fb[SET][j], acc[i][j], 0, 0, 0);  \
  }
#define VP_TAP_OFS(D, K) ((((py + (D).a0 + ((D).nb == 2 ? ((K) >> 1) : (K))) * HWD) + px + (D).b0 + ((D).nb == 2 ? ((K) & 1) : 0)) * ROWB)

  // ---- prologue: halo(chunk cA) and weight tiles 0, 1 -> LDS; halo(chunk cA + 1) -> registers
  bool abl_prologue = true;   // (ablation hooks: the prologue always stages)
  VP_LOAD_H(dA)
  VP_STORE_H(0, VP_SK_DEAD1(dA))
  VP_DMA_W(0, 0)
  if (S > 1) VP_DMA_W(PL * W_BYTES, 1)
  VP_WAIT_VMCNT(0);
  __syncthreads();
  UpconvChunk d = dA;                                            // current chunk
  UpconvChunk dn = dA;                                           // next chunk (valid while c + 1 < cB)
  if (cA + 1 < cB) {
    dn = upconv_chunk(cA + 1, py, px, CinV, CsV);
    VP_LOAD_H(dn)
  }
  VP_READ_FRAGS(0, w_base, halo_base, VP_TAP_OFS(d, 0))
  abl_prologue = false;

  int c = cA, t = 0, s = 0, hb = 0;
  int w_cur = 0, w_nxt = PL * W_BYTES, w_fre = 2 * PL * W_BYTES;   // byte offsets of the three weight buffers: read now / next step / being filled
  // One K step.  KIND 0: not the chunk's last tap.  KIND 1 / 2 / 3: the chunk's LAST tap, where the halo of chunk c + 1 (in the staging registers since
  // the previous chunk's last step) goes to LDS and (KIND 1) the registers are refilled with chunk c + 2; 2: no chunk c + 2; 3: no chunk c + 1.
  // The kinds are separate straight-line instances (the chunk loop below is peeled by them) because the compiler's wait-count model is not path
  // sensitive: with "store if last tap" and "refill if last tap" as two conditionals of ONE loop body it assumed the refill's destination registers
  // could still be the targets of the previous refill's loads and put s_waitcnt vmcnt(0) in front of it -- draining the weight tile requested a few
  // instructions earlier, an L2 round trip with the matrix pipe idle at every chunk's last step (seen in the ISA).
  // PREVLD: the step in front of this chunk's first step refilled the staging registers (its loads sit BEHIND that step's DMA in the vmcnt order).
#define VP_STEP(KIND, PREVLD)                                                                \
  {                                                                                          \
    const int tap_cur_ = VP_TAP_OFS(d, t);                                                   \
    const int tap_nxt_ = (KIND) == 0 ? VP_TAP_OFS(d, t + 1) : ((KIND) == 3 ? tap_cur_ : VP_TAP_OFS(dn, 0)); \
    const char* hbuf = halo_base + (HDB ? hb : 0) * PL * HSTRIDE;                            \
    const char* hbuf_next = (HDB && (KIND) != 0) ? halo_base + (hb ^ 1) * PL * HSTRIDE : hbuf; \
    /* HDB: the next chunk's halo goes to the OTHER halo image at the head of this chunk's last step, BEFORE this step's DMA is issued (the compiler */ \
    /* guards the registers with s_waitcnt vmcnt(0): at this point the only other thing in flight is the weight tile requested one step ago, which */ \
    /* this step's barrier needs anyway -- kernels_conv3x3_x3.hip, tap 3)                                                                          */ \
    if constexpr (HDB && ((KIND) == 1 || (KIND) == 2)) { VP_STORE_H(hb ^ 1, VP_SK_DEAD1(dn)) }   \
    /* weight tile of step s + 2 -> the buffer step s - 1 read last (its barrier has passed); must land before the NEXT step's barrier */ \
    if (s + 2 < S) VP_DMA_W(w_fre, s + 2)                                                    \
    __builtin_amdgcn_sched_barrier(0);                                                       \
    VP_MFMA_RANGE(0, 0, MT * NT / 2)                                                         \
    __builtin_amdgcn_sched_barrier(0);                                                       \
    VP_READ_FRAGS(1, w_base + w_cur, hbuf, tap_cur_)                                         \
    __builtin_amdgcn_sched_barrier(0);  /* keep the prefetch AHEAD of the MFMAs (the scheduler sinks it otherwise) */ \
    UpconvChunk dnn = dn;                                                                    \
    if constexpr (HDB && (KIND) == 1) {                                                      \
      dnn = upconv_chunk(c + 2, py, px, CinV, CsV);                                        \
      VP_LOAD_H(dnn)                                                                         \
    }                                                                                        \
    VP_MFMA_RANGE(0, MT * NT / 2, MT * NT)                                                   \
    __builtin_amdgcn_sched_barrier(0);                                                       \
    /* What THIS barrier must publish is the weight tile requested ONE STEP AGO (tile s + 1); vmcnt retires in issue order, so it has landed once at */ \
    /* most {the halo loads issued behind it in the previous step, this step's DMA, this step's halo loads} are outstanding                          */ \
    {                                                                                        \
      const int dma_n = s + 2 < S ? PL * WPIECES : 0;                                        \
      const int ld_prev = ((PREVLD) && t == 0) ? PL * HP : 0;                                \
      const int ld_this = (HDB && (KIND) == 1) ? PL * HP : 0;                                \
      VP_WAIT_VMCNT_RT(dma_n + ld_prev + ld_this);                                           \
    }                                                                                        \
    /* THE BARRIER SITS BETWEEN THE TWO K SUB-STEPS (kernels_conv3x3_x3.hip): the only LDS operations outstanding here are set 1's reads */ \
    VP_LDS_BARRIER();                                                                        \
    if constexpr (!HDB && ((KIND) == 1 || (KIND) == 2)) {                                    \
      /* single halo image: every wave has completed its last read of chunk c (the barrier's lgkmcnt(0)) -- the next chunk's pieces go over it, a */ \
      /* second barrier opens it, the registers are refilled with chunk c + 2                                                                    */ \
      VP_STORE_H(0, VP_SK_DEAD1(dn))                                                         \
      VP_LDS_BARRIER();                                                                      \
      if constexpr ((KIND) == 1) {                                                           \
        dnn = upconv_chunk(c + 2, py, px, CinV, CsV);                                      \
        VP_LOAD_H(dnn)                                                                       \
      }                                                                                      \
    }                                                                                        \
    /* ---- K sub-step 1: set 0 of the NEXT step is fetched while set 1 multiplies */       \
    VP_READ_FRAGS(0, w_base + w_nxt, hbuf_next, tap_nxt_)                                    \
    __builtin_amdgcn_sched_barrier(0);                                                       \
    VP_MFMA_RANGE(1, 0, MT * NT)                                                             \
    __builtin_amdgcn_sched_barrier(0);                                                       \
    /* ---- advance the step state (wave-uniform scalars) */                                \
    {                                                                                        \
      const int w_old = w_cur;                                                               \
      w_cur = w_nxt;                                                                         \
      w_nxt = w_fre;                                                                         \
      w_fre = w_old;                                                                         \
    }                                                                                        \
    ++s;                                                                                     \
    if constexpr ((KIND) == 0) {                                                             \
      ++t;                                                                                   \
    } else {                                                                                 \
      ++c;                                                                                   \
      t = 0;                                                                                 \
      d = dn;                                                                                \
      dn = dnn;                                                                              \
      hb ^= 1;                                                                               \
    }                                                                                        \
  }
  // chunks with two successors: store + refill at the last tap (the prologue refilled the registers for the first of them)
  while (c + 2 < cB) {
    while (t + 1 < d.nt) VP_STEP(0, true)
    VP_STEP(1, true)
  }
  if (c + 1 < cB) {   // the last but one chunk: store, nothing left to refill with
    while (t + 1 < d.nt) VP_STEP(0, true)
    VP_STEP(2, true)
  }
  while (t + 1 < d.nt) VP_STEP(0, false)   // the last chunk (the step in front of it refilled nothing)
  VP_STEP(3, false)
#undef VP_STEP
#undef VP_SK_DEAD1
#undef VP_TAP_OFS
#undef VP_MFMA_RANGE
#undef VP_READ_FRAGS
#undef VP_STORE_H
#undef VP_LOAD_H
#undef VP_DMA_W

  const PixPhase pix{y0, x0, p.H, p.W, py, px};
  if constexpr ((ABL & 16) != 0) {   // ablation: keep the accumulators alive, skip the epilogue
    if (p.H == -12345) {
#pragma unroll
      for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
          for (int r = 0; r < 16; ++r) p.partial[(i * NT + j) * 16 + r + tid * 64] = acc[i][j][r];
    }
    return;
  }
  if constexpr (SPLITK) {
    // fp32 partial sums straight from the accumulators: lanes l and l + 32 hold channels 8g + 0..3 / 8g + 4..7 of pixel l & 31
    const int M2 = 4 * p.H * p.W;
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      const int m = pix((wpx * NT + j) * 32 + (lane & 31));
      if (m < 0) continue;
      float* row = p.partial + ((size_t)zsplit * M2 + m) * p.CoutW + co0 + 4 * (lane >> 5);
#pragma unroll
      for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const f32x4_t v = {acc[i][j][4 * g + 0], acc[i][j][4 * g + 1], acc[i][j][4 * g + 2], acc[i][j][4 * g + 3]};
          *reinterpret_cast<f32x4_t*>(row + (i * WCO + wco) * 32 + 8 * g) = v;
        }
    }
    return;
  } else {
    // ---- register epilogue: bias (by border class) + activation + (hi, lo) split, both planes staged as [pixel][CO_TILE] fp16, 16-byte stores
    constexpr int PITCH = CO_TILE * 2 + 16, STAGE_PLANE = PX * PITCH;
    static_assert(PL * STAGE_PLANE <= NHB * PL * HSTRIDE + 3 * PL * W_BYTES, "stage fits the main buffers");
    // (X1: one output plane -- the lo half of the stage and out_lo stay untouched)
    // bias (by the pixel's border class) and prescale vectors of this lane's accumulators: ALL requested before the barrier, so their round trips
    // overlap each other and the barrier instead of one load -> wait -> GELU chain per register group
    const float* const wsc = p.wscale + (size_t)phase * p.CoutW + co0;
    f32x4_t bb[NT][MT][4], sc[MT][4];
    int qpix[NT];
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      qpix[j] = (wpx * NT + j) * 32 + (lane & 31);
      const int m = pix(qpix[j]);
      const float* const bsrc = p.bias + (size_t)(m >= 0 ? bias_class(m, 2 * p.H, 2 * p.W) : 4) * p.CoutW + co0;
#pragma unroll
      for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int g = 0; g < 4; ++g) bb[j][i][g] = *reinterpret_cast<const f32x4_t*>(bsrc + (i * WCO + wco) * 32 + 4 * (lane >> 5) + 8 * g);
    }
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
      for (int g = 0; g < 4; ++g) sc[i][g] = *reinterpret_cast<const f32x4_t*>(wsc + (i * WCO + wco) * 32 + 4 * (lane >> 5) + 8 * g);   // 2^-prescale: exact product
    __syncthreads();  // every wave has finished its last K sub-step (and the dead prefetch behind the last barrier has landed)
#pragma unroll
    for (int j = 0; j < NT; ++j) {
#pragma unroll
      for (int i = 0; i < MT; ++i) {
        const int cl = (i * WCO + wco) * 32 + 4 * (lane >> 5);
        char* row = smem + qpix[j] * PITCH + cl * 2;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          h4_t h, l;
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float x = apply_act(fmaf(acc[i][j][4 * g + r], sc[i][g][r], bb[j][i][g][r]), ACT);
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
      if constexpr ((ABL & 32) != 0) { if (p.H != -12345) continue; }   // ablation: epilogue arithmetic and staging, no global stores
      const size_t o = (size_t)m * p.Cstore + co;
      *reinterpret_cast<h8_t*>(p.out_hi + o) = *reinterpret_cast<const h8_t*>(smem + r * PITCH + c8 * 16);
      if constexpr (!X1) *reinterpret_cast<h8_t*>(p.out_lo + o) = *reinterpret_cast<const h8_t*>(smem + STAGE_PLANE + r * PITCH + c8 * 16);
    }
  }
}

// Sums the K slices of a split launch in the fixed order z = 0 .. nsplit - 1 (deterministic) and applies the stage's epilogue: phase prescale,
// border-class bias, activation, (hi, lo) split; a thread = 8 channels of one high-resolution pixel.
__global__ __launch_bounds__(256) void upconv_finish_kernel(const UpconvParams p) {
  const int H2 = 2 * p.H, W2 = 2 * p.W, M2 = H2 * W2;
  const int groups = p.Ncols >> 3;
  const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
  if (t >= (long long)M2 * groups) return;
  const int m = (int)(t / groups), co = (int)(t % groups) * 8;
  float v[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  for (int z = 0; z < p.nsplit; ++z) {
    const float* src = p.partial + ((size_t)z * M2 + m) * p.CoutW + co;
    const f32x4_t q0 = *reinterpret_cast<const f32x4_t*>(src), q1 = *reinterpret_cast<const f32x4_t*>(src + 4);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      v[r] += q0[r];
      v[4 + r] += q1[r];
    }
  }
  const int Y = m / W2, X = m - Y * W2;
  const int phase = (Y & 1) * 2 + (X & 1);
  const float* b = p.bias + (size_t)bias_class(m, H2, W2) * p.CoutW + co;
  const float* s = p.wscale + (size_t)phase * p.CoutW + co;
  h8_t hi, lo;
#pragma unroll
  for (int r = 0; r < 8; ++r) {
    const float x = apply_act(fmaf(v[r], s[r], b[r]), p.act);
    hi[r] = (half_t)x;
    lo[r] = (half_t)(x - (float)hi[r]);
  }
  const size_t o = (size_t)m * p.Cstore + co;
  *reinterpret_cast<h8_t*>(p.out_hi + o) = hi;
  if (p.out_lo) *reinterpret_cast<h8_t*>(p.out_lo + o) = lo;   // (VP_FP16 engines: one plane)
}

bool upconv_supported(const UpconvParams& p, int shape) {
  if (shape != 6 && shape != 7) return false;
  const bool x1 = p.in_lo == nullptr;   // VP_FP16 engines: one plane per tensor, 64-channel chunks (template parameter X1)
  if (!(p.in_hi && p.w_hi && p.w_lo && p.bias && p.wscale && p.out_hi && (x1 ? p.out_lo == nullptr : p.out_lo != nullptr))) return false;
  if (p.H < 1 || p.W < 1 || p.Cin < (x1 ? 64 : 32) || p.Cin % (x1 ? 64 : 32) != 0 || p.Cs % 32 != 0 || p.Cs < 0) return false;
  if (p.Cs > 0 && !(p.sk_hi && (x1 ? p.sk_lo == nullptr : p.sk_lo != nullptr))) return false;
  if (p.CoutW % 128 != 0 || p.Ncols % 8 != 0 || p.Ncols > p.CoutW || p.Cstore < p.Ncols) return false;
  if (!(p.act == (x1 ? ACT_GELU_F16 : ACT_GELU) || p.act == ACT_NONE)) return false;
  if (p.nsplit < 1 || p.nsplit > (x1 ? upconv_chunks(p.Cin >> 1, ((p.Cs + 63) >> 6) << 5) : upconv_chunks(p.Cin, p.Cs))) return false;
  if (p.nsplit > 1 && !p.partial) return false;
  // element offsets are 32-bit in the kernel
  if ((long long)p.H * p.W * p.Cin >= (1ll << 31) || 4ll * p.H * p.W * std::max(p.Cs, 1) >= (1ll << 31) || 4ll * p.H * p.W * p.Cstore >= (1ll << 31)) return false;
  return true;
}

template <int TH, int WPX, bool HDB, bool X1>
static hipError_t launch_upconv_cfg(const UpconvParams& p, hipStream_t st) {
  constexpr int CO = 128;
  constexpr int lds = (HDB ? 2 : 1) * 2 * ((TH + 2) * 18 * 80) + 6 * (CO * 64);
  static_assert(lds <= 160 * 1024, "LDS budget");
  constexpr int GELU = X1 ? ACT_GELU_F16 : ACT_GELU;
  const bool gelu = p.act == GELU, sk = p.nsplit > 1;
  auto k = sk ? upconv_x3_kernel<CO, TH, 2, WPX, HDB, ACT_NONE, true, X1>
              : (gelu ? upconv_x3_kernel<CO, TH, 2, WPX, HDB, GELU, false, X1> : upconv_x3_kernel<CO, TH, 2, WPX, HDB, ACT_NONE, false, X1>);
  static LdsAttrOnce attr_once[3];
  if (hipError_t e = set_max_dynamic_lds(attr_once[sk ? 2 : gelu], reinterpret_cast<const void*>(k), lds); e != hipSuccess) return e;
  dim3 grid(((p.H + TH - 1) / TH) * ((p.W + 15) / 16) * 4 * (p.CoutW / CO) * p.nsplit);
  hipLaunchKernelGGL(k, grid, dim3(64 * 2 * WPX), lds, st, p);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess || !sk) return e;
  const long long n = 4ll * p.H * p.W * (p.Ncols >> 3);
  hipLaunchKernelGGL(upconv_finish_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, p);
  return hipGetLastError();
}

hipError_t launch_upconv(const UpconvParams& p, int shape, hipStream_t st) {
  if (!upconv_supported(p, shape)) return hipErrorInvalidValue;
  if (p.in_lo == nullptr) {   // VP_FP16 engines
    if (shape == 6) return launch_upconv_cfg<16, 4, true, true>(p, st);
    return launch_upconv_cfg<8, 2, false, true>(p, st);
  }
  if (shape == 6) return launch_upconv_cfg<16, 4, true, false>(p, st);
  return launch_upconv_cfg<8, 2, false, false>(p, st);
}

// ------------------------------------------------------------------------------------------------ weight composition (load time)
// C[m][n] = sum over the group's (A, B) pairs of sum_k A[m][k] * B[n][k]: fp32 operands, every product exact in fp64, fp64 accumulation.
// 64 x 64 tile per workgroup, 16 x 16 threads with a 4 x 4 register tile, K in steps of 16 through LDS.  Load-time work (124 GFLOP fp64 for a scene
// network, ~10 ms); not a hot-path kernel.
__global__ __launch_bounds__(256) void compose_gemm_kernel(const ComposeGemmParams* groups, int tiles_n) {
  __shared__ float As[16][64 + 4], Bs[16][64 + 4];
  const ComposeGemmParams& g = groups[blockIdx.y];
  const int tm = blockIdx.x / tiles_n, tn = blockIdx.x - tm * tiles_n;
  const int tid = threadIdx.x, tx = tid & 15, ty = tid >> 4;
  const int lrow = tid >> 2, lk = (tid & 3) * 4;   // staging: row of the tile, first of 4 consecutive k
  double acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = 0.0;
  for (int pr = 0; pr < g.pairs; ++pr) {
    const float* A = g.a[pr];
    const float* B = g.b[pr];
    for (int k0 = 0; k0 < g.K; k0 += 16) {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int k = k0 + lk + e;
        const int ma = tm * 64 + lrow, nb = tn * 64 + lrow;
        As[lk + e][lrow] = (ma < g.M && k < g.K) ? A[(size_t)ma * g.K + k] : 0.0f;
        Bs[lk + e][lrow] = (nb < g.N && k < g.K) ? B[(size_t)nb * g.K + k] : 0.0f;
      }
      __syncthreads();
#pragma unroll
      for (int k = 0; k < 16; ++k) {
        double a[4], b[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          a[i] = (double)As[k][ty * 4 + i];
          b[i] = (double)Bs[k][tx * 4 + i];
        }
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < 4; ++j) acc[i][j] = fma(a[i], b[j], acc[i][j]);
      }
      __syncthreads();
    }
  }
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int m = tm * 64 + ty * 4 + i, n = tn * 64 + tx * 4 + j;
      if (m < g.M && n < g.N) g.c[(size_t)m * g.N + n] = acc[i][j];
    }
}

hipError_t launch_compose_gemm(const ComposeGemmParams* groups_dev, int n_groups, int M, int N, hipStream_t st) {
  const int tiles_m = (M + 63) / 64, tiles_n = (N + 63) / 64;
  hipLaunchKernelGGL(compose_gemm_kernel, dim3(tiles_m * tiles_n, n_groups), dim3(256), 0, st, groups_dev, tiles_n);
  return hipGetLastError();
}

}  // namespace vp
