// ConvTranspose2d(k2,s2) + fused 1x1 skip link on the SMALL maps (neck stages: 10x20, 20x40, 40x80 input pixels, K = 544..1376,
// N = 4 x 512..1280 weight rows): a software-pipelined GEMM fed entirely by LDS-DMA.
//
// The implicit-GEMM kernel (kernels_conv.hip: 4 waves, register-staged double buffer, __syncthreads per K step, split-K on the
// smallest map) ran these three layers at 65-164 TFLOP/s algorithmic in the parity mode (42 us each for 2.8-7 GFLOP): a lone wave
// per SIMD alternating "stage -> barrier -> fragments -> MFMA", partial sums of the split through HBM.  Here, as in the 3x3
// kernels of kernels_conv3x3_x3.hip:
//   * 8 waves (two per SIMD), workgroup tile 256 weight rows x 128 pixels, every wave 64 x 64 (12 MFMAs per 16-channel sub-step
//     on 8 fragment reads);
//   * both operands travel global -> LDS by LDS-DMA, three K steps (32 channels each) deep: tile s+2 is requested when step s
//     starts; 64-byte rows, the 16-byte slots XOR-swizzled by (row >> 2) & 3 -- the swizzle is applied to the GLOBAL address each
//     lane supplies, so activations (which no host code can pre-arrange) land conflict-free too;
//   * one LDS-only barrier per K step (VP_LDS_BARRIER + explicit vmcnt: the tile requested last step stays in flight);
//   * split-K only where the weights are the stream (10x20 / 20x40 pixels: 40 / 84 tiles, 28 / 10 MB of weights from HBM);
//   * the weight rows are permuted on the DMA's global side so that a lane's accumulators are 32 consecutive output channels:
//     wave-private fp32 patch, 8-channel pieces, eight lanes = 128 contiguous bytes of an output pixel (as kernels_convt_rs.hip);
//   * K extension: steps past Cin read the skip tensor at the output pixel of the workgroup's quadrant (the 256-row tile lies
//     inside one quadrant).
#include <cstdlib>

#include "conv_epilogue.hpp"
#include "lds_dma.hpp"

namespace vp {

// ABL: ablation bits for tools/gemm_dma_ablate.hip only (1 = no DMA in the K loop, 2 = no MFMA, 4 = no LDS fragment reads in the loop,
// 8 = no barrier / vmcnt wait in the loop, 16 = no epilogue); always 0 in the library.
template <bool SPLIT, int ABL = 0>
__global__ __launch_bounds__(512, 2) void gemm_dma_kernel(const ConvGemmParams p) {
  constexpr int CO_T = 256, PX_T = 128, BK = 32, ROWB = 64;
  constexpr int PLANES = SPLIT ? 2 : 1;
  constexpr int A_BYTES = CO_T * ROWB, B_BYTES = PX_T * ROWB;      // one plane of one stage
  constexpr int STAGE = PLANES * (A_BYTES + B_BYTES);              // [A hi][A lo][B hi][B lo]
  constexpr int OFF_B = PLANES * A_BYTES;
  constexpr int D = PLANES * 3;                                    // DMA instructions per wave and stage: 2 A row sets + 1 B row set, per plane
  constexpr int MT = 2, NT = 2;
  constexpr int PP = 64 * 4 + 16, PATCH = 32 * PP;                 // wave-private fp32 patch [32 px][64 co] (+16: odd slot pitch)
  static_assert(8 * PATCH <= 3 * STAGE, "patches reuse the stage ring");
  extern __shared__ __attribute__((aligned(16))) char smem[];      // [3][STAGE]

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wco = wave >> 1, wpx = wave & 1;                       // 4 x 2 waves, 64 rows x 64 pixels each
  const int M = p.H * p.W;
  const int n_px_tiles = (M + PX_T - 1) / PX_T;
  // the workgroups that share a weight tile (its pixel tiles) get consecutive ids = the same XCD: the weights are the big operand
  int vid;
  {
    const int nwg = gridDim.x, q = nwg >> 3, r = nwg & 7, xcd = blockIdx.x & 7;
    vid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (blockIdx.x >> 3);
  }
  const int n_co_tiles = p.Ncols / CO_T;
  const int tile_px = vid % n_px_tiles, tile_co = (vid / n_px_tiles) % n_co_tiles, zsplit = vid / (n_px_tiles * n_co_tiles);
  const int m0 = tile_px * PX_T, co0 = tile_co * CO_T;
  // K steps of 32 channels; the first KS1 (of the whole K) read the ConvTranspose input.  Split-K: slice z covers the steps
  // [KS_all z / nsplit, KS_all (z + 1) / nsplit) and leaves its fp32 sums in p.partial[z] (summed in z order, bias added and
  // stored by splitk_finish_kernel, kernels_conv.hip) -- the weight-bound layers on 10x20 / 20x40 pixels, where a CU pulls
  // weights from HBM at only ~50 GB/s (LDS-deep prefetch / latency) and 40-84 tiles leave most of the chip's bandwidth unused
  const int Kw = p.Cin + p.Cin2, KS_all = Kw / BK, KS1 = p.Cin / BK;
  const int s_first = (int)(((long long)KS_all * zsplit) / p.nsplit);
  const int KS = (int)(((long long)KS_all * (zsplit + 1)) / p.nsplit) - s_first;  // steps of THIS slice; s below is slice-relative
  const int quad = co0 / p.Cstore, qdy = quad >> 1, qdx = quad & 1;

  // ---- DMA plan.  A lane moves the 16-byte slot (lane & 3) of row 16 j + (lane >> 2); the slot holds the logical chunk
  // slot ^ swz(row).  A rows: j = wave and wave + 8; B rows: j = wave.
  const int slot = lane & 3;
  // WEIGHTS: the host packs every (256-row tile, K step) as one contiguous 16 KB block per plane in its LDS IMAGE order
  // (gemm_dma_pack_index below: rows permuted, slots swizzled), so a DMA instruction is a linear 1 KB copy of eight full cache
  // lines (64-byte row pieces of the plain [rows][K] matrix were 16 half lines per instruction: the L2 -> LDS path is what bounds
  // this kernel).  This lane's 16 bytes of the wave's two 1 KB pieces (j = wave, wave + 8) of the tile of K step 0:
  size_t a_src[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) a_src[i] = (size_t)tile_co * KS_all * (CO_T * BK) + (size_t)(wave + 8 * i) * 512 + lane * 8;
  long long b_src1, b_src2;  // element offsets of this lane's chunk of its pixel row: ConvTranspose input / skip tensor
  {
    const int r = 16 * wave + (lane >> 2);
    const int m = min(m0 + r, M - 1);                              // rows past the image re-read the last pixel; never stored
    const int y = m / p.W, x = m - y * p.W;
    const int ch = (slot ^ ((r >> 2) & 3)) << 3;
    b_src1 = (long long)m * p.Cin + ch;
    b_src2 = ((long long)(2 * y + qdy) * (2 * p.W) + (2 * x + qdx)) * p.Cin2 + ch;
  }
#define VP_DMA_STAGE(S, BUF)                                                                                       \
  {                                                                                                                \
    char* st_ = smem + (BUF) * STAGE + wave * 1024;                                                                \
    const int k0_ = (s_first + (S)) * BK;                                                                          \
    _Pragma("unroll") for (int i = 0; i < 2; ++i) {                                                                \
      VP_GLOBAL_LOAD_LDS16(p.w_hi + a_src[i] + (size_t)(s_first + (S)) * (CO_T * BK), st_ + i * 8192);             \
      if constexpr (SPLIT) VP_GLOBAL_LOAD_LDS16(p.w_lo + a_src[i] + (size_t)(s_first + (S)) * (CO_T * BK), st_ + A_BYTES + i * 8192); \
    }                                                                                                              \
    if (s_first + (S) < KS1) {                                                                                     \
      VP_GLOBAL_LOAD_LDS16(p.in_hi + b_src1 + k0_, st_ + OFF_B);                                                   \
      if constexpr (SPLIT) VP_GLOBAL_LOAD_LDS16(p.in_lo + b_src1 + k0_, st_ + OFF_B + B_BYTES);                    \
    } else {                                                                                                       \
      VP_GLOBAL_LOAD_LDS16(p.in_hi + p.in2_delta_hi + b_src2 + (k0_ - p.Cin), st_ + OFF_B);                        \
      if constexpr (SPLIT) VP_GLOBAL_LOAD_LDS16(p.in_lo + p.in2_delta_lo + b_src2 + (k0_ - p.Cin), st_ + OFF_B + B_BYTES); \
    }                                                                                                              \
  }
  VP_DMA_STAGE(0, 0)
  if (KS > 1) VP_DMA_STAGE(1, 1)

  // fragment addressing (the LDS image of kernels_conv3x3_x3.hip's weight tiles): row * 64 + ((2 sub + kh) ^ swz(row)) * 16
  const int frow = lane & 31, fswz = (frow >> 2) & 3;
  const int a_ofs = (wco * 64 + frow) * ROWB + (((lane >> 5) ^ fswz) << 4);   // + i * 32 rows; K sub-step 1: ^ 32 bytes
  const int b_ofs = OFF_B + (wpx * 64 + frow) * ROWB + (((lane >> 5) ^ fswz) << 4);

  f32x16_t acc[MT][NT];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

  // fragment sets [K sub-step parity]: set 0 = channels 0..15 of the step's 32, set 1 = channels 16..31.  Schedule of a step (the
  // one of kernels_conv3x3_x3.hip): set 0 was fetched behind the PREVIOUS step's barrier; half of its MFMAs, then set 1's reads,
  // the other half, then "tile s+1 has landed" + the barrier (between the two K sub-steps, where no prefetch is in flight), then
  // set 0 of the NEXT step is fetched while set 1 multiplies.  Left to itself the compiler reuses one fragment set and waits
  // for every batch of ds_read_b128 with the matrix pipe idle (measured: 2.1 us per step against 0.75 us of MFMA issue).
  h8_t fa[2][MT], fal[2][SPLIT ? MT : 1], fb[2][NT], fbl[2][SPLIT ? NT : 1];
#define VP_READ_FRAGS(SET, ST)                                                                                     \
  {                                                                                                                \
    _Pragma("unroll") for (int i = 0; i < MT; ++i) {                                                               \
      fa[SET][i] = *reinterpret_cast<const h8_t*>((ST) + ((a_ofs + i * 32 * ROWB) ^ ((SET) * 32)));                \
      if constexpr (SPLIT) fal[SET][i] = *reinterpret_cast<const h8_t*>((ST) + A_BYTES + ((a_ofs + i * 32 * ROWB) ^ ((SET) * 32))); \
    }                                                                                                              \
    _Pragma("unroll") for (int j = 0; j < NT; ++j) {                                                               \
      fb[SET][j] = *reinterpret_cast<const h8_t*>((ST) + ((b_ofs + j * 32 * ROWB) ^ ((SET) * 32)));                \
      if constexpr (SPLIT) fbl[SET][j] = *reinterpret_cast<const h8_t*>((ST) + B_BYTES + ((b_ofs + j * 32 * ROWB) ^ ((SET) * 32))); \
    }                                                                                                              \
  }
#define VP_MFMA_RANGE(SET, Q0, Q1)                                                                                 \
  _Pragma("unroll") for (int q_ = (Q0); q_ < (Q1); ++q_) {                                                         \
    const int i = q_ / NT, j = q_ % NT;                                                                            \
    if constexpr (SPLIT) {                                                                                         \
      acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fal[SET][i], fb[SET][j], acc[i][j], 0, 0, 0);             \
      acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[SET][i], fbl[SET][j], acc[i][j], 0, 0, 0);             \
    }                                                                                                              \
    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[SET][i], fb[SET][j], acc[i][j], 0, 0, 0);                \
  }
  VP_WAIT_VMCNT(D);  // tile 0 has landed (tile 1 may be in flight; with a single K step this waits for less than asked, harmlessly)
  if (KS == 1) { VP_WAIT_VMCNT(0); }
  VP_LDS_BARRIER();
  VP_READ_FRAGS(0, smem)
  for (int s = 0; s < KS; ++s) {
    const char* st = smem + (s % 3) * STAGE;
    const char* st_next = smem + ((s + 1) % 3) * STAGE;
    // tile s+2 -> the buffer tile s-1 was read from (everyone passed the previous step's barrier with those reads complete)
    if constexpr (!(ABL & 1)) {
      if (s + 2 < KS) VP_DMA_STAGE(s + 2, (s + 2) % 3)
    }
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (!(ABL & 2)) VP_MFMA_RANGE(0, 0, MT * NT / 2)
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (!(ABL & 4)) VP_READ_FRAGS(1, st)
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (!(ABL & 2)) VP_MFMA_RANGE(0, MT * NT / 2, MT * NT)
    // tile s+1 (requested one step ago) must be visible behind the barrier; the tile requested in this step stays in flight
    if constexpr (!(ABL & 8)) {
      if (s + 2 < KS && !(ABL & 1)) { VP_WAIT_VMCNT(D); } else { VP_WAIT_VMCNT(0); }
      VP_LDS_BARRIER();
    }
    if constexpr (!(ABL & 4)) {
      if (s + 1 < KS) VP_READ_FRAGS(0, st_next)
    }
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (!(ABL & 2)) VP_MFMA_RANGE(1, 0, MT * NT)
    if constexpr ((ABL & 2) != 0) {  // keep the fragments alive
      acc[0][0][0] += (float)fa[0][0][0] + (float)fb[0][0][0] + (float)fa[1][0][0] + (float)fb[1][0][0];
    }
  }
#undef VP_MFMA_RANGE
#undef VP_READ_FRAGS
#undef VP_DMA_STAGE
  VP_LDS_BARRIER();  // every wave has read its last fragments: the ring becomes the epilogue patches
  if constexpr ((ABL & 16) != 0) {  // ablation: keep the accumulators alive (a store no launch ever takes), skip the epilogue
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

  // ---- wave-private epilogue, one 32-pixel tile at a time
  char* const mypatch = smem + wave * PATCH;
  char* const pw = mypatch + (lane & 31) * PP + (32 * (lane >> 5)) * 4;   // + 64 T + 16 g
  const int pc = lane & 7, r0 = lane >> 3;                                 // 8-channel piece pc of patch rows r0, r0 + 8, ...
  const char* const pr = mypatch + r0 * PP + pc * 32;
  const int co = co0 + wco * 64 + pc * 8;
  if (p.nsplit > 1) {  // fp32 partial sums: eight lanes = 256 contiguous bytes of one pixel's row
#pragma unroll
    for (int j = 0; j < NT; ++j) {
#pragma unroll
      for (int T = 0; T < MT; ++T)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const f32x4_t v = {acc[T][j][4 * g + 0], acc[T][j][4 * g + 1], acc[T][j][4 * g + 2], acc[T][j][4 * g + 3]};
          *reinterpret_cast<f32x4_t*>(pw + 64 * T + 16 * g) = v;
        }
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int pass = 0; pass < 4; ++pass) {
        const int m = m0 + wpx * 64 + j * 32 + r0 + pass * 8;
        if (m < M) {
          float* dst = p.partial + ((size_t)zsplit * M + m) * p.CoutW + co;
          *reinterpret_cast<f32x4_t*>(dst) = *reinterpret_cast<const f32x4_t*>(pr + pass * 8 * PP);
          *reinterpret_cast<f32x4_t*>(dst + 4) = *reinterpret_cast<const f32x4_t*>(pr + pass * 8 * PP + 16);
        }
      }
      __builtin_amdgcn_wave_barrier();
    }
    return;
  }
  // the bias is added in the accumulator layout, BEFORE the row loop: a register loaded from global memory and used inside the
  // (divergent: ragged last tile) store loop made the compiler guard every pass with s_waitcnt vmcnt(0), i.e. wait for the
  // previous pass's stores
  {
    const float* bsrc = p.bias + co0 + wco * 64 + 32 * (lane >> 5);
    const float* ssrc = p.wscale + co0 + wco * 64 + 32 * (lane >> 5);  // 2^-prescale of the weight rows (exact product)
#pragma unroll
    for (int T = 0; T < MT; ++T)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const f32x4_t bv = *reinterpret_cast<const f32x4_t*>(bsrc + 16 * T + 4 * g);
        const f32x4_t sv = *reinterpret_cast<const f32x4_t*>(ssrc + 16 * T + 4 * g);
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
          for (int r = 0; r < 4; ++r) acc[T][j][4 * g + r] = fmaf(acc[T][j][4 * g + r], sv[r], bv[r]);
      }
  }
  const f32x4_t b0 = {0.f, 0.f, 0.f, 0.f}, b1 = b0, one4 = {1.f, 1.f, 1.f, 1.f};
  // pixel-shuffle address: quadrant / channel of this lane's piece are fixed, (y, x) of its first row divided ONCE, then advanced
  // by 8 pixels per pass (two integer divisions per 16-byte store were ~110 VALU instructions each)
  const int cq = co - quad * p.Cstore, W2 = 2 * p.W;
  int em = m0 + wpx * 64 + r0, ey = em / p.W, ex = em - ey * p.W;
#pragma unroll
  for (int j = 0; j < NT; ++j) {
#pragma unroll
    for (int T = 0; T < MT; ++T)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const f32x4_t v = {acc[T][j][4 * g + 0], acc[T][j][4 * g + 1], acc[T][j][4 * g + 2], acc[T][j][4 * g + 3]};
        *reinterpret_cast<f32x4_t*>(pw + 64 * T + 16 * g) = v;
      }
    // same wave writes and reads the patch: LDS operations of a wave complete in order (the wave barrier emits no instruction;
    // it states the dependency for the compiler and for the CPU emulation, tests/emul)
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int pass = 0; pass < 4; ++pass) {
      const f32x4_t s0 = *reinterpret_cast<const f32x4_t*>(pr + pass * 8 * PP);
      const f32x4_t s1 = *reinterpret_cast<const f32x4_t*>(pr + pass * 8 * PP + 16);
      float v[8] = {s0[0], s0[1], s0[2], s0[3], s1[0], s1[1], s1[2], s1[3]};
      if (em < M) epilogue_store8<STORE_SHUFFLE2, RES_NONE, ACT_NONE>(p, M, em, co, v, b0, b1, one4, one4, ((long long)(2 * ey + qdy) * W2 + (2 * ex + qdx)) * p.Cstore + cq);
      em += 8;
      ex += 8;
      while (ex >= p.W) {
        ex -= p.W;
        ++ey;
      }
    }
    __builtin_amdgcn_wave_barrier();
  }
}

// Where element (row n, column k) of the [ncols][kw] weight matrix goes in the packed layout: tile (n / 256, k / 32) is a block of
// 256 x 32 elements in LDS image order -- LDS row R = 64 w + 32 T + rho (rho = 8 g + 4 h + i2: MFMA row) holds weight row
// 64 w + 32 h + 16 T + 4 g + i2 of the tile (accumulator register 4 g + i2 of tile T in lane (pixel, h) is then channel
// 32 h + 16 T + 4 g + i2 of the wave's 64), and the row's four 16-byte slots are XOR-swizzled by (R >> 2) & 3.
size_t gemm_dma_pack_index(int n, int k, int kw) {
  const int tile_co = n >> 8, r = n & 255, w = r >> 6, c = r & 63;       // c = 32 h + 16 T + 4 g + i2
  const int h = c >> 5, T = (c >> 4) & 1, g = (c >> 2) & 3, i2 = c & 3;
  const int R = 64 * w + 32 * T + 8 * g + 4 * h + i2;
  const int s = k >> 5, chunk = (k & 31) >> 3, e = k & 7;
  return ((size_t)tile_co * (kw / 32) + s) * (256 * 32) + (size_t)R * 32 + ((chunk ^ ((R >> 2) & 3)) << 3) + e;
}

bool gemm_dma_shape_ok(int M, int ncols, int cin_pad, int cin2_pad, int cstore) {
  return M >= 128 && ncols % 256 == 0 && cstore % 256 == 0 && cin_pad % 32 == 0 && cin2_pad % 32 == 0 && cin_pad + cin2_pad >= 256;
}

bool gemm_dma_supported(const ConvGemmParams& p, bool split) {
  return split == (p.in_lo != nullptr) && split == (p.out_lo != nullptr) && p.ks == 1 && p.stride <= 1 && p.store_mode == STORE_SHUFFLE2 &&
         p.act == ACT_NONE && p.res_mode == RES_NONE && p.post_act == ACT_NONE && p.nsplit >= 1 && p.nsplit <= (p.Cin + p.Cin2) / 32 &&
         (p.nsplit == 1 || p.partial != nullptr) && p.CoutW == p.Ncols && p.out_hi != nullptr &&
         gemm_dma_shape_ok(p.H * p.W, p.Ncols, p.Cin, p.Cin2, p.Cstore);
}

namespace {
template <bool SPLIT>
hipError_t launch_gemm_dma_cfg(const ConvGemmParams& p, hipStream_t st) {
  constexpr int lds = 3 * (SPLIT ? 2 : 1) * (256 + 128) * 64;
  static_assert(lds <= 160 * 1024, "LDS budget");
  auto k = gemm_dma_kernel<SPLIT>;
  static LdsAttrOnce attr_once;
  if (hipError_t e = set_max_dynamic_lds(attr_once, reinterpret_cast<const void*>(k), lds); e != hipSuccess) return e;
  const int M = p.H * p.W;
  hipLaunchKernelGGL(k, dim3(((M + 127) / 128) * (p.Ncols / 256) * p.nsplit), dim3(512), lds, st, p);
  hipError_t e = hipGetLastError();
  if (e == hipSuccess && p.nsplit > 1) e = launch_splitk_finish(p, st);
  return e;
}
}  // namespace

hipError_t launch_gemm_dma(const ConvGemmParams& p, hipStream_t st) {
  const bool split = p.in_lo != nullptr;
  if (!gemm_dma_supported(p, split)) return hipErrorInvalidValue;
  return split ? launch_gemm_dma_cfg<true>(p, st) : launch_gemm_dma_cfg<false>(p, st);
}

}  // namespace vp
