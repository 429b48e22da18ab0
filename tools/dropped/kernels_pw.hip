// Pointwise (1x1) convolution for the SMALL GEMMs of the encoder: LDS-free operand path.  EXPERIMENT, OPT-IN (VP_PW=1 or
// tile 4 of vp_op_conv2d): parity-green but SLOWER than the LDS-pipelined kernel on MI355X -- 662 vs 474 us (fp16x3) and
// 390 vs 330 us (fp16) over the encoder's 32 GEMMs: a fragment load touches 64 different rows of 16 bytes each, which the
// memory pipeline serialises, and the long-K projections leave a few dozen waves walking K alone.  Kept with its tests as the
// measured answer to "skip LDS for tiny GEMMs" (tried and dropped).
//
// EfficientNet-B0's 32 expand / project / top 1x1 convolutions are 0.03-0.17 GFLOP each (K = 32..1152 channels, 200..51 200
// pixels).  Through the generic implicit-GEMM kernel (kernels_conv.hip: DEPTH-deep register ring -> double-buffered LDS,
// one barrier per K step, LDS-staged block epilogue) each costs 7-15 us -- microseconds of fixed pipeline cost around a K
// loop of one to a few dozen steps; together a third of the single-stream frame latency for 0.8 % of its FLOPs.
// Here a wave loads its MFMA fragments STRAIGHT FROM GLOBAL MEMORY: the A fragment of v_mfma_f32_32x32x16_f16 is
// "lane & 31 = row, lane >> 5 = which 8 of the 16 k" = 16 contiguous bytes of a [rows][K] matrix per lane, and both
// operands are stored that way (weights [CoutW][Cin], activations NHWC = [pixels][Cin]).  No LDS staging, no barrier in
// the K loop, no workgroup-wide epilogue: the next 32-channel block's fragments are in flight while the current block
// multiplies, and every wave transposes its own 64co x 32px result through a wave-private LDS patch into 128-byte (per
// plane) row pieces.  Operand re-reads (weights by every pixel tile, pixels by every channel tile) are L1 / L2 hits at
// these sizes.  Bias + activation + residual + (hi, lo) split reuse epilogue_store8 (conv_epilogue.hpp).
#include "conv_epilogue.hpp"

namespace vp {

template <bool SPLIT, int ACT, int RES>
__global__ __launch_bounds__(256) void pw_gemm_kernel(const ConvGemmParams p) {
  constexpr int MT = 2;                       // wave tile: 64 output channels x 32 pixels
  constexpr int PITCH = 64 * 4 + 16;          // fp32 patch row (one pixel, 64 channels) + pad
  extern __shared__ __attribute__((aligned(16))) char smem[];  // [4 waves][32][PITCH]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int M = p.H * p.W, K = p.Cin;
  const int n_co_tiles = p.CoutW >> 6;
  const int tile_co = blockIdx.x % n_co_tiles, tile_px = blockIdx.x / n_co_tiles;  // channel tile fastest: the pixel rows stay in L2
  const int co0 = tile_co * 64, m0 = tile_px * 128 + wave * 32;
  if (m0 >= M) return;  // whole wave beyond the image (waves are independent: no barrier anywhere)

  typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
  const int kofs = (lane >> 5) * 8;
  const int mrow = m0 + (lane & 31) < M ? m0 + (lane & 31) : M - 1;  // rows past the end re-read the last pixel; their results are never stored
  const half_t* a_hi = p.w_hi + (size_t)(co0 + (lane & 31)) * K + kofs;
  const half_t* b_hi = p.in_hi + (size_t)mrow * K + kofs;
  const half_t* a_lo = SPLIT ? p.w_lo + (size_t)(co0 + (lane & 31)) * K + kofs : nullptr;
  const half_t* b_lo = SPLIT ? p.in_lo + (size_t)mrow * K + kofs : nullptr;
  const size_t a_tile = (size_t)32 * K;  // second channel tile of the wave

  f32x16_t acc[MT];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.0f;

  // fragments of one 32-channel block: [kk][i] for A, [kk] for B
  h8_t fa[2][2][MT], fal[2][2][MT], fb[2][2], fbl[2][2];
#define VP_PW_LOAD(SET, K0)                                                                       \
  _Pragma("unroll") for (int kk = 0; kk < 2; ++kk) {                                             \
    _Pragma("unroll") for (int i = 0; i < MT; ++i) {                                             \
      fa[SET][kk][i] = *reinterpret_cast<const h8_t*>(a_hi + i * a_tile + (K0) + kk * 16);        \
      if constexpr (SPLIT) fal[SET][kk][i] = *reinterpret_cast<const h8_t*>(a_lo + i * a_tile + (K0) + kk * 16); \
    }                                                                                            \
    fb[SET][kk] = *reinterpret_cast<const h8_t*>(b_hi + (K0) + kk * 16);                          \
    if constexpr (SPLIT) fbl[SET][kk] = *reinterpret_cast<const h8_t*>(b_lo + (K0) + kk * 16);    \
  }
#define VP_PW_MFMA(SET)                                                                           \
  _Pragma("unroll") for (int kk = 0; kk < 2; ++kk) _Pragma("unroll") for (int i = 0; i < MT; ++i) { \
    if constexpr (SPLIT) {                                                                        \
      acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fal[SET][kk][i], fb[SET][kk], acc[i], 0, 0, 0); \
      acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[SET][kk][i], fbl[SET][kk], acc[i], 0, 0, 0); \
    }                                                                                             \
    acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[SET][kk][i], fb[SET][kk], acc[i], 0, 0, 0); \
  }
  VP_PW_LOAD(0, 0)
  int k0 = 0;
  for (; k0 + 64 <= K; k0 += 64) {  // two blocks per trip: compile-time fragment sets
    VP_PW_LOAD(1, k0 + 32)
    VP_PW_MFMA(0)
    if (k0 + 64 < K) VP_PW_LOAD(0, k0 + 64)
    VP_PW_MFMA(1)
  }
  if (k0 < K) VP_PW_MFMA(0)  // odd number of 32-channel blocks: the last one was loaded into set 0
#undef VP_PW_LOAD
#undef VP_PW_MFMA

  // ---- wave-private transposing epilogue: accumulators (lane & 31 = pixel, register 4g + r = channel 8g + 4 (lane >> 5) + r)
  // -> fp32 patch [32 px][64 co] -> 8 channels of one pixel per lane -> bias / activation / residual / split / 16-byte stores
  char* const patch = smem + wave * 32 * PITCH;
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const f32x4_t v = {acc[i][4 * g + 0], acc[i][4 * g + 1], acc[i][4 * g + 2], acc[i][4 * g + 3]};
      *reinterpret_cast<f32x4_t*>(patch + (lane & 31) * PITCH + (i * 32 + 8 * g + 4 * (lane >> 5)) * 4) = v;
    }
  // same wave wrote and reads the patch: LDS operations of a wave complete in order (the wave barrier emits no instruction;
  // it states the dependency for the compiler and for the CPU emulation, tests/emul)
  __builtin_amdgcn_wave_barrier();
  const int c8 = lane & 7, co = co0 + c8 * 8;
  if (co >= p.Ncols) return;
  const f32x4_t b0 = *reinterpret_cast<const f32x4_t*>(p.bias + co), b1 = *reinterpret_cast<const f32x4_t*>(p.bias + co + 4);
#pragma unroll
  for (int pass = 0; pass < 4; ++pass) {
    const int r = pass * 8 + (lane >> 3), m = m0 + r;
    if (m >= M) continue;
    const f32x4_t s0 = *reinterpret_cast<const f32x4_t*>(patch + r * PITCH + c8 * 32);
    const f32x4_t s1 = *reinterpret_cast<const f32x4_t*>(patch + r * PITCH + c8 * 32 + 16);
    float v[8] = {s0[0], s0[1], s0[2], s0[3], s1[0], s1[1], s1[2], s1[3]};
    epilogue_store8<STORE_NHWC, RES, ACT>(p, M, m, co, v, b0, b1);
  }
}

bool pw_gemm_supported(const ConvGemmParams& p) {
  return p.ks == 1 && p.stride <= 1 && p.Cin2 == 0 && p.nsplit == 1 && p.store_mode == STORE_NHWC && p.post_act == ACT_NONE &&
         (p.res_mode == RES_NONE || p.res_mode == RES_ADD) && (p.act == ACT_NONE || p.act == ACT_SILU || p.act == ACT_SILU_F16) &&
         p.Cin % 32 == 0 && p.CoutW % 64 == 0 && p.out_hi != nullptr;
}

template <bool SPLIT>
static hipError_t launch_pw_split(const ConvGemmParams& p, hipStream_t st) {
  constexpr int lds = 4 * 32 * (64 * 4 + 16);
  void (*k)(const ConvGemmParams) = nullptr;
  const bool res = p.res_mode == RES_ADD;
  if (p.act == ACT_NONE) k = res ? pw_gemm_kernel<SPLIT, ACT_NONE, RES_ADD> : pw_gemm_kernel<SPLIT, ACT_NONE, RES_NONE>;
  else if (p.act == ACT_SILU) k = res ? pw_gemm_kernel<SPLIT, ACT_SILU, RES_ADD> : pw_gemm_kernel<SPLIT, ACT_SILU, RES_NONE>;
  else k = res ? pw_gemm_kernel<SPLIT, ACT_SILU_F16, RES_ADD> : pw_gemm_kernel<SPLIT, ACT_SILU_F16, RES_NONE>;
  const int M = p.H * p.W;
  dim3 grid(((M + 127) / 128) * (p.CoutW / 64));
  hipLaunchKernelGGL(k, grid, dim3(256), lds, st, p);
  return hipGetLastError();
}

hipError_t launch_pw_gemm(const ConvGemmParams& p, bool split, hipStream_t st) {
  if (!pw_gemm_supported(p) || split != (p.in_lo != nullptr)) return hipErrorInvalidValue;
  return split ? launch_pw_split<true>(p, st) : launch_pw_split<false>(p, st);
}

}  // namespace vp
