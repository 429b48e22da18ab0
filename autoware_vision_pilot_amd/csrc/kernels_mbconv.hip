// MBConv front half in ONE launch: expand 1x1 (+BN +SiLU) -> depthwise k x k / stride s (+BN +SiLU) -> squeeze-excite pool sums
// (torchvision efficientnet_b0().features[1..7], MBConv.block[0..1] + the average pool of block[2]; Models/model_components/backbone.py:9-22).
//
// The encoder is 0.4 % of a frame's FLOPs and a third of one camera's latency: a chain of ~56 dependent launches of 5-20 us inside the
// replayed graph (profiles/r03_trace_sceneseg_fp16x3_single_stream.tsv), and with several cameras in flight it still costs ~250 us of
// every 2.5 ms frame (profiles/r03_encoder_hiding_bound.txt).  The expanded tensor (6 x the block's input channels) is the largest
// tensor of every block, written by one launch and read back by the next.  Here it never leaves the CU:
//   1. a workgroup owns an output patch (8x16 pixels at stride 1, 4x16 at stride 2) x 32 expanded channels; the input patch under it
//      -- (TH-1) s + k rows x (TW-1) s + k columns: the depthwise halo -- is expanded by an MFMA GEMM (halo pixels x Cin -> 32
//      channels, 32-channel input chunks staged global -> LDS, 3 MFMAs per product as everywhere in the parity mode): the halo is
//      RECOMPUTED by neighbouring workgroups (1.4x / 1.9x of the expand FLOPs at k = 3 / 5, nothing at the encoder's scale);
//   2. bias + SiLU in registers, pixels outside the image forced to ZERO (the depthwise pads its INPUT, i.e. the expanded tensor, with
//      zeros -- not with expand(0)), the fp32 tile goes to LDS over the dead staging buffers;
//   3. the depthwise taps read that tile, bias + SiLU, (hi, lo) store, per-channel pool sums as 2^24 fixed-point int64 (LDS atomics, then
//      one global atomic per channel and workgroup into a replica row: integer adds, bit-deterministic) -- the same arithmetic, tap order
//      and pool protocol as dwconv_pool_kernel (kernels_backbone.hip), whose launch this replaces together with the expand GEMM's.
// The depthwise sees the expand output in fp32 instead of its (hi, lo) rounding: a hair closer to the oracle, not bit-identical to the
// two-launch path.  fp16x3 engines, one frame per pass (the batched encoder and the fp16 engines keep the two launches).
#include "act_io.hpp"
#include "conv_epilogue.hpp"
#include "se_phases.hpp"

namespace vp {

// SiLU of the fused halves: x * rcp(1 + exp2(-x log2 e)) on v_exp_f32 / v_rcp_f32 (about 1 ulp each: |error| <= 3e-7 |silu(x)|, the class of the
// GELU in the convolution epilogues).  silu_f's libm expf + IEEE division are ~30 instructions per value: 3-9 us of EVERY front launch
// (46 values per thread; tools/mbf_check.hip, profiles/r03_mbf_check.txt).
__device__ __forceinline__ float silu_mb(float x) { return x * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(x * -1.4426950408889634f)); }

template <int K, int S>
struct MbTile {
  static constexpr int TH = S == 1 ? 8 : 4, TW = 16;
  static constexpr int IH = (TH - 1) * S + K, IW = (TW - 1) * S + K, HPX = IH * IW;
  static constexpr int NF = (HPX + 31) / 32;          // 32-pixel MFMA column tiles of the halo
  static constexpr int XPITCH = 80;                   // bytes per staged pixel row (32 channels fp16 + 16: conflict-free ds_read_b128)
  static constexpr int X_BYTES = NF * 32 * XPITCH;    // one plane of the input chunk
  static constexpr int W_BYTES = 32 * XPITCH;         // one plane of the weight chunk (32 expanded channels x 32 input channels)
  static constexpr int EPITCH = 32 * 4 + 16;          // bytes per expanded pixel row (32 channels fp32 + 16)
  static constexpr int E_BYTES = HPX * EPITCH;
  static constexpr int STAGE = 2 * X_BYTES + 2 * W_BYTES;
  static constexpr int MAIN = STAGE > E_BYTES ? STAGE : E_BYTES;   // the expanded tile overlays the staging buffers
  static constexpr int LDS = MAIN + K * K * 32 * 4 + 32 * 8 + 64 * 32 * 4;   // + depthwise filter slice + pool accumulators + squeeze-FC slice
};

// NW = waves of the workgroup: 4 where the grid is several workgroups per CU (the 80x160 and 160x320 maps), 8 on the smaller maps -- there a
// CU holds ONE workgroup, and with one wave per SIMD the chain LDS stage -> barrier -> fragment reads -> dependent MFMAs -> barrier of a K chunk
// ran strictly one after the other (1.05 us per chunk, profiles/r03_mbf_check.txt); two waves per SIMD interleave their halves of it.
// ABL (tools/mbf_check.hip only): 1 = no expand loop, 2 = no depthwise taps, 4 = no pool atomics (LDS and global), 8 = no SiLU,
// 16 = expand loop without its global loads after chunk 0, 32 = expand loop without LDS staging / fragment reads / MFMAs (loads only)
// X3: the parity mode's (hi, lo) planes and three MFMAs per product; false (round 4: the VP_FP16 engines, until then on the two-launch path):
// one plane in, one MFMA per product, one plane out -- the lo staging buffers stay unused, the layout is the same.
template <int K, int S, int NW, int ABL = 0, bool X3 = true>
__global__ __launch_bounds__(64 * NW) void mbconv_front_kernel(const MbFrontParams p) {
  using T = MbTile<K, S>;
  constexpr int NT = 64 * NW;
  constexpr int TH = T::TH, TW = T::TW, IW = T::IW, HPX = T::HPX, NF = T::NF, NFW = (NF + NW - 1) / NW, XP = T::XPITCH, EP = T::EPITCH;
  constexpr int pad = (K - 1) / 2;
  extern __shared__ unsigned char dw_smem[];
  unsigned char* const xs_hi = dw_smem;
  unsigned char* const xs_lo = xs_hi + T::X_BYTES;
  unsigned char* const ws_hi = xs_lo + T::X_BYTES;
  unsigned char* const ws_lo = ws_hi + T::W_BYTES;
  unsigned char* const es = dw_smem;                                             // [HPX][EP] fp32, after the GEMM
  float* const wl = reinterpret_cast<float*>(dw_smem + T::MAIN);                 // [K*K][32]
  unsigned long long* const red64 = reinterpret_cast<unsigned long long*>(dw_smem + T::MAIN + K * K * 32 * 4);  // [32]
  float* const w1s = reinterpret_cast<float*>(dw_smem + T::MAIN + K * K * 32 * 4 + 32 * 8);                       // [sq][32] squeeze FC, this workgroup's 32 channels

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int OW = p.out.W, OH = p.out.H;
  const int tiles_x = (OW + TW - 1) / TW;
  const int tyi = blockIdx.x / tiles_x, txi = blockIdx.x - tyi * tiles_x;
  const int oy0 = tyi * TH, ox0 = txi * TW;
  const int iy0 = oy0 * S - pad, ix0 = ox0 * S - pad;   // image coordinates of halo pixel (0, 0)
  const int c0 = blockIdx.y * 32;                       // first expanded channel of this workgroup
  const int Cin = p.in.C, Cexp = p.out.C;
  typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
  const u32x4 zero4 = {0u, 0u, 0u, 0u};

  // depthwise filter slice + pool accumulators (read after several barriers)
  for (int i = tid; i < K * K * 8; i += NT) {
    const int tp = i >> 3, j = i & 7;
    *reinterpret_cast<f32x4_t*>(wl + tp * 32 + j * 4) = *reinterpret_cast<const f32x4_t*>(p.w_dw + (size_t)tp * Cexp + c0 + j * 4);
  }
  if (tid < 32) red64[tid] = 0ull;
  if (p.w1)  // fetched now, used after the last barrier: the tail of the kernel is then LDS reads + one atomic per squeeze unit
    for (int i = tid; i < p.sq * 8; i += NT)
      *reinterpret_cast<f32x4_t*>(w1s + (i >> 3) * 32 + (i & 7) * 4) = *reinterpret_cast<const f32x4_t*>(p.w1 + (size_t)(i >> 3) * p.out.C + blockIdx.y * 32 + (i & 7) * 4);

  // ---- 1: expand GEMM over the halo patch.  A = weights (32 expanded channels x 16 k), B = pixels (32 halo pixels x 16 k); a lane's
  // 16 accumulators of a tile are channels 8 g + 4 (lane >> 5) + r (g, r = 0..3) of halo pixel 32 f + (lane & 31).
  f32x16_t acc[NFW];
#pragma unroll
  for (int j = 0; j < NFW; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[j][r] = 0.0f;
  const int KC = Cin >> 5;
  // staging plan: 16-byte piece i of this thread = (halo pixel, 8-channel part) q = tid + NT i of a chunk; element offset of the piece in
  // chunk 0, or -1 (outside the image / past the patch: zeros).  The pieces of chunk c + 1 are REQUESTED before the MFMAs of chunk c and
  // parked in registers: the loop is a latency chain (load -> LDS -> barrier -> fragments -> MFMA) of 1-6 links, one per 32 input channels
  constexpr int PCS = (NF * 32 * 4 + NT - 1) / NT;
  int x_off[PCS];
#pragma unroll
  for (int i = 0; i < PCS; ++i) {
    const int q = tid + NT * i, hp = q >> 2, part = q & 3;
    const int hy = hp / IW, hx = hp - hy * IW;
    const int gy = iy0 + hy, gx = ix0 + hx;
    const bool ok = q < NF * 32 * 4 && hp < HPX && (unsigned)gy < (unsigned)p.in.H && (unsigned)gx < (unsigned)p.in.W;
    x_off[i] = ok ? (gy * p.in.W + gx) * Cin + part * 8 : -1;
  }
  const int w_row = tid >> 2, w_part = tid & 3;
  const size_t w_off0 = (size_t)(c0 + (w_row & 31)) * Cin + w_part * 8;
  u32x4 rx_hi[PCS], rx_lo[X3 ? PCS : 1], rw_hi = zero4, rw_lo = zero4;
#define VP_MB_LOAD(C)                                                                                  \
  {                                                                                                    \
    _Pragma("unroll") for (int i = 0; i < PCS; ++i) {                                                  \
      const int o_ = (x_off[i] >= 0 ? x_off[i] : 0) + (C) * 32;                                        \
      const u32x4 vh_ = *reinterpret_cast<const u32x4*>(p.in.hi + o_);                                 \
      rx_hi[i] = x_off[i] >= 0 ? vh_ : zero4;                                                          \
      if constexpr (X3) {                                                                              \
        const u32x4 vl_ = *reinterpret_cast<const u32x4*>(p.in.lo + o_);                               \
        rx_lo[i] = x_off[i] >= 0 ? vl_ : zero4;                                                        \
      }                                                                                                \
    }                                                                                                  \
    if (tid < 128) {                                                                                   \
      rw_hi = *reinterpret_cast<const u32x4*>(p.w_hi + w_off0 + (C) * 32);                             \
      if constexpr (X3) rw_lo = *reinterpret_cast<const u32x4*>(p.w_lo + w_off0 + (C) * 32);           \
    }                                                                                                  \
  }
  VP_MB_LOAD(0)
  for (int c = 0; c < ((ABL & 1) ? 0 : KC); ++c) {
    if constexpr ((ABL & 32) != 0) {  // loads only: fold the pieces into one accumulator element so that they stay
      u32x4 t_ = rw_hi ^ rw_lo;
#pragma unroll
      for (int i = 0; i < PCS; ++i) t_ ^= X3 ? (rx_hi[i] ^ rx_lo[X3 ? i : 0]) : rx_hi[i];
      acc[0][0] += __uint_as_float((t_[0] ^ t_[1] ^ t_[2] ^ t_[3]) & 0x3f800000u);
      if (c + 1 < KC) VP_MB_LOAD(c + 1)
      continue;
    }
#pragma unroll
    for (int i = 0; i < PCS; ++i) {
      const int q = tid + NT * i;
      if (q < NF * 32 * 4) {
        *reinterpret_cast<u32x4*>(xs_hi + (q >> 2) * XP + (q & 3) * 16) = rx_hi[i];
        if constexpr (X3) *reinterpret_cast<u32x4*>(xs_lo + (q >> 2) * XP + (q & 3) * 16) = rx_lo[X3 ? i : 0];
      }
    }
    if (tid < 128) {  // 32 rows x 4 pieces per plane
      *reinterpret_cast<u32x4*>(ws_hi + w_row * XP + w_part * 16) = rw_hi;
      if constexpr (X3) *reinterpret_cast<u32x4*>(ws_lo + w_row * XP + w_part * 16) = rw_lo;
    }
    __syncthreads();
    if constexpr ((ABL & 16) == 0) {
      if (c + 1 < KC) VP_MB_LOAD(c + 1)
    }
#pragma unroll
    for (int ss = 0; ss < 2; ++ss) {
      const int fo = (lane & 31) * XP + ss * 32 + (lane >> 5) * 16;
      const h8_t a_hi = *reinterpret_cast<const h8_t*>(ws_hi + fo);
      h8_t a_lo;
      if constexpr (X3) a_lo = *reinterpret_cast<const h8_t*>(ws_lo + fo);
#pragma unroll
      for (int j = 0; j < NFW; ++j) {
        const int f = wave + NW * j;
        if (f < NF) {
          const h8_t b_hi = *reinterpret_cast<const h8_t*>(xs_hi + f * 32 * XP + fo);
          if constexpr (X3) {
            const h8_t b_lo = *reinterpret_cast<const h8_t*>(xs_lo + f * 32 * XP + fo);
            acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a_lo, b_hi, acc[j], 0, 0, 0);
            acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a_hi, b_lo, acc[j], 0, 0, 0);
          }
          acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a_hi, b_hi, acc[j], 0, 0, 0);
        }
      }
    }
    __syncthreads();  // the chunk is consumed: the next one (or the expanded tile) may overwrite it
  }
#undef VP_MB_LOAD

  // ---- 2: bias + SiLU, zero outside the image, fp32 tile -> LDS (over the staging buffers)
  {
    f32x4_t be[4], se4[4];   // se4: 2^-prescale of the expand weight rows (MbFrontParams::s_exp), exact product
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      be[g] = *reinterpret_cast<const f32x4_t*>(p.b_exp + c0 + 8 * g + 4 * (lane >> 5));
      se4[g] = *reinterpret_cast<const f32x4_t*>(p.s_exp + c0 + 8 * g + 4 * (lane >> 5));
    }
#pragma unroll
    for (int j = 0; j < NFW; ++j) {
      const int f = wave + NW * j;
      const int hp = f * 32 + (lane & 31);
      if (f < NF && hp < HPX) {
        const int hy = hp / IW, hx = hp - hy * IW;
        const bool in_img = (unsigned)(iy0 + hy) < (unsigned)p.in.H && (unsigned)(ix0 + hx) < (unsigned)p.in.W;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          f32x4_t v;
#pragma unroll
          for (int r = 0; r < 4; ++r) v[r] = in_img ? ((ABL & 8) ? fmaf(acc[j][4 * g + r], se4[g][r], be[g][r]) : silu_mb(fmaf(acc[j][4 * g + r], se4[g][r], be[g][r]))) : 0.0f;
          *reinterpret_cast<f32x4_t*>(es + hp * EP + (8 * g + 4 * (lane >> 5)) * 4) = v;
        }
      }
    }
  }
  __syncthreads();

  // ---- 3: depthwise + SiLU + store + pool.  Item = (output pixel of the patch, channel octet): 128 x 4 (stride 1) or 64 x 4 items.  Pool: an
  // LDS atomic per value (2^24 fixed-point integers: any order gives the same bits).  Measured alternative, not kept: summing a thread's
  // items in registers and the lanes of an octet by shuffles before ONE atomic per (wave, octet) is ~1 us SLOWER per launch (64 64-bit
  // ds_bpermutes against 16 same-address LDS atomics, which the LDS resolves at a lane a clock).
  constexpr int ITEMS = TH * TW * 4;
  for (int it = tid; it < ITEMS; it += NT) {
    const int og = it & 3, pl = it >> 2;
    const int ty = pl / TW, tx = pl - ty * TW;
    const int oy = oy0 + ty, ox = ox0 + tx;
    if (oy >= OH || ox >= OW) continue;
    const int cc = c0 + og * 8;
    float a8[8];
    {
      const f32x4_t b0 = *reinterpret_cast<const f32x4_t*>(p.b_dw + cc), b1 = *reinterpret_cast<const f32x4_t*>(p.b_dw + cc + 4);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        a8[i] = b0[i];
        a8[4 + i] = b1[i];
      }
    }
#pragma unroll
    for (int ky = 0; ky < ((ABL & 2) ? 1 : K); ++ky)
#pragma unroll
      for (int kx = 0; kx < ((ABL & 2) ? 1 : K); ++kx) {
        const unsigned char* src = es + ((ty * S + ky) * IW + (tx * S + kx)) * EP + og * 32;
        const f32x4_t v0 = *reinterpret_cast<const f32x4_t*>(src), v1 = *reinterpret_cast<const f32x4_t*>(src + 16);
        const float* wk = wl + (ky * K + kx) * 32 + og * 8;
        const f32x4_t w0 = *reinterpret_cast<const f32x4_t*>(wk), w1 = *reinterpret_cast<const f32x4_t*>(wk + 4);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          a8[i] = fmaf(v0[i], w0[i], a8[i]);
          a8[4 + i] = fmaf(v1[i], w1[i], a8[4 + i]);
        }
      }
    if (!(ABL & 8)) {
#pragma unroll
      for (int i = 0; i < 8; ++i) a8[i] = silu_mb(a8[i]);
    }
    store8(p.out, ((size_t)oy * OW + ox) * Cexp + cc, a8);
    if (!(ABL & 4)) {
#pragma unroll
      for (int i = 0; i < 8; ++i) atomicAdd(&red64[og * 8 + i], (unsigned long long)(long long)__float2ll_rn(a8[i] * 16777216.0f));
    }
  }
  __syncthreads();
  if (ABL & 4) return;
  if (p.sums && tid < 32) {  // the per-channel sums: only for a back half that rebuilds the means itself (device-scope atomics execute at the
    const unsigned long long v = red64[tid];   // memory side: the 32 + sq of a workgroup were 3-6 us at the end of every launch)
    if (v != 0ull) atomicAdd(p.sums + (size_t)(blockIdx.x & (p.replicas - 1)) * Cexp + c0 + tid, v);
  }
  // squeeze FC of the squeeze-excite, this workgroup's share (32 channels x its patch): linear in the sums, so it can be taken here and
  // added up as integers like them -- mbconv_back then reads sq numbers instead of streaming the FC matrix through every workgroup
  if (p.w1 && tid >= 64 && tid < 64 + p.sq) {
    const int j = tid - 64;
    const f32x4_t* wr = reinterpret_cast<const f32x4_t*>(w1s + j * 32);
    float z = 0.f;
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const f32x4_t w4 = wr[q];
#pragma unroll
      for (int i = 0; i < 4; ++i) z = fmaf(w4[i], (float)((double)(long long)red64[4 * q + i] * (1.0 / 16777216.0)), z);
    }
    atomicAdd(p.zsums + (size_t)(blockIdx.x & (p.replicas - 1)) * 64 + j, (unsigned long long)(long long)__float2ll_rn(z * 16777216.0f));
  }
}

bool mbconv_front_supported(const MbFrontParams& p) {
  const bool x3 = p.in.lo != nullptr;   // one plane everywhere (VP_FP16) or two everywhere (VP_FP16X3)
  return p.in.hi && p.out.hi && p.w_hi && (x3 ? (p.out.lo && p.w_lo) : (!p.out.lo && !p.w_lo)) && p.b_exp && p.s_exp && p.w_dw && p.b_dw && (p.sums || (p.w1 && p.zsums)) && (p.k == 3 || p.k == 5) &&
         (p.stride == 1 || p.stride == 2) && (!p.w1 || (p.zsums && p.sq >= 1 && p.sq <= 64)) && (p.in.C & 31) == 0 && (p.out.C & 31) == 0 && p.replicas >= 1 && (p.replicas & (p.replicas - 1)) == 0 &&
         p.out.H == p.in.H / p.stride && p.out.W == p.in.W / p.stride && p.in.H % p.stride == 0 && p.in.W % p.stride == 0;
}

template <int K, int S, int NW, int ABL = 0, bool X3 = true>
static hipError_t launch_mb_nw(const MbFrontParams& p, hipStream_t st) {
  using T = MbTile<K, S>;
  static_assert(T::LDS <= 160 * 1024, "LDS budget");
  static LdsAttrOnce once;
  auto k = mbconv_front_kernel<K, S, NW, ABL, X3>;
  if (hipError_t e = set_max_dynamic_lds(once, reinterpret_cast<const void*>(k), T::LDS); e != hipSuccess) return e;
  const dim3 grid(((p.out.H + T::TH - 1) / T::TH) * ((p.out.W + T::TW - 1) / T::TW), p.out.C / 32);
  hipLaunchKernelGGL(k, grid, dim3(64 * NW), T::LDS, st, p);
  return hipGetLastError();
}
template <int K, int S, int ABL = 0>
static hipError_t launch_mb(const MbFrontParams& p, hipStream_t st) {
  // maps up to 40x80 (<= 200 workgroups: one per CU): eight waves; the big maps keep four (three workgroups per CU)
  if (p.in.lo == nullptr) return p.out.H * p.out.W <= 3200 ? launch_mb_nw<K, S, 8, ABL, false>(p, st) : launch_mb_nw<K, S, 4, ABL, false>(p, st);
  return p.out.H * p.out.W <= 3200 ? launch_mb_nw<K, S, 8, ABL>(p, st) : launch_mb_nw<K, S, 4, ABL>(p, st);
}

hipError_t launch_mbconv_front(const MbFrontParams& p, hipStream_t st) {
  if (!mbconv_front_supported(p)) return hipErrorInvalidValue;
  if (p.k == 3 && p.stride == 1) return launch_mb<3, 1>(p, st);
  if (p.k == 5 && p.stride == 1) return launch_mb<5, 1>(p, st);
  if (p.k == 3 && p.stride == 2) return launch_mb<3, 2>(p, st);
  return launch_mb<5, 2>(p, st);
}

// ------------------------------------------------------------------------------------------------------------------- back half
// MBConv back half in ONE launch: squeeze-excite tail (means -> squeeze FC + SiLU -> excite FC + sigmoid) -> projection 1x1 (+BN)
// with the gate folded into its K axis (+ residual) -- torchvision MBConv.block[2..3] + the stochastic-depth-free skip
// (Models/model_components/backbone.py:9-22).  It replaces se_gate_scale + the projection GEMM + (on the <= 20x40 maps) its
// split-K finish: three dependent launches of 7-18 + 6-13 + 5-6 us in the replayed graph for a few hundred MFLOP.
//   * workgroup = 32 WM pixels x 32 output channels, ALL of K.  WM = 1 (maps up to 40x80: 42 ... 200 workgroups): EIGHT waves take
//     an eighth of the K steps each and meet in LDS (fixed order: bit-deterministic); 512 threads also halve the round trips of the
//     two FC phases.  WM = 4 (80x160, 160x320, K = 32 ... 160): four waves, each owns 32 pixels and all of K, so the gate is rebuilt
//     by 100 / 400 workgroups instead of 400 / 1600.
//   * every workgroup REBUILDS the gate itself (the two FC matrices are <= 2 x 221 KB, L2-resident): no launch boundary, no
//     device-wide hand-off (a buffer_wbl2 per workgroup costs more than the FCs: profiles/r03_splitk_fold_ab.tsv).
//   * the operands do not pass through LDS: a lane's MFMA fragments ARE 16 / 32 contiguous bytes of a pixel row / weight row, loaded
//     straight from global (L2) memory, three K steps in flight; the first batch and the residual are requested BEFORE the gate
//     phases, so the FC round trips and the operand round trip overlap.
//   * W'[n][k] = W[n][k] * gate[k], (hi, lo) split, per fragment in registers -- the arithmetic of se_gate_scale's phase 4.
struct MbBackFrag {
  f32x4_t w0, w1;  // 8 fp32 projection weights of this lane's row
  h8_t xh, xl;     // 8 channels of this lane's pixel, (hi, lo)
};

template <int WM, int NW>
struct MbBack {
  static constexpr int NT = 64 * NW;           // threads
  static constexpr int NSH = NW / WM;          // K shares (waves per pixel tile)
  static constexpr int RP = 32 * 4 + 16;       // bytes per pixel row of a wave's partial tile
  static constexpr int ITEMS = WM * 256;       // epilogue items (pixel, 4 channels) of the workgroup
  static constexpr int IPT = (ITEMS + NT - 1) / NT;
  static size_t lds(int C) { return (size_t)C * (8 * se_acc_rows(C, NT) + 8) + (size_t)NW * 32 * RP; }
};

// ABL (tools/mbb_check.hip only): 1 = no gate phases (gate = 0.5), 2 = no projection loop, 4 = no means, 8 = no squeeze FC, 16 = no excite FC
template <int WM, int NW, int ABL = 0, bool X3 = true>
__global__ __launch_bounds__(64 * NW) __attribute__((amdgpu_waves_per_eu(WM == 4 ? 3 : 2))) void mbconv_back_kernel(const MbBackParams p) {
  using T = MbBack<WM, NW>;
  constexpr int U = 3;                  // K steps per batch in flight
  constexpr int NT = T::NT, NSH = T::NSH, RP = T::RP, IPT = T::IPT;
  extern __shared__ __attribute__((aligned(16))) unsigned char bk_smem[];
  const int C = p.se.C;
  const int acc_rows = se_acc_rows(C, NT);
  unsigned long long* const accs = reinterpret_cast<unsigned long long*>(bk_smem);             // [acc_rows][C]
  float* const mean = reinterpret_cast<float*>(bk_smem + (size_t)acc_rows * C * 8);            // [C]
  float* const gate = mean + C;                                                                 // [C]
  unsigned char* const rb = reinterpret_cast<unsigned char*>(gate + C);                         // [NW waves][32 pixels][RP]
  __shared__ __attribute__((aligned(16))) float s1[64];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int M = p.in.H * p.in.W, Cout = p.out.C;
  const int pxw = blockIdx.x * 32 * WM, co0 = blockIdx.y * 32;   // first pixel of the workgroup
  const int px0 = pxw + (wave % WM) * 32;                        // ... and of this wave's tile
  // this wave's K steps (16 channels each)
  const int KS = C >> 4, sh = wave / WM;
  const int ks0 = sh * KS / NSH, nst = (sh + 1) * KS / NSH - ks0;
  const int pxl = min(px0 + (lane & 31), M - 1);        // rows past the map repeat its last pixel (never stored)
  const float* const wrow = p.w + (size_t)(co0 + (lane & 31)) * C + 8 * (lane >> 5);
  const size_t xoff = (size_t)pxl * C + 8 * (lane >> 5);
  MbBackFrag nxt[U];
#define VP_MBB_LOAD(S0)                                                               \
  {                                                                                   \
    _Pragma("unroll") for (int u = 0; u < U; ++u) {                                   \
      const int k_ = min(ks0 + (S0) + u, KS - 1) * 16;                                \
      nxt[u].w0 = *reinterpret_cast<const f32x4_t*>(wrow + k_);                       \
      nxt[u].w1 = *reinterpret_cast<const f32x4_t*>(wrow + k_ + 4);                   \
      nxt[u].xh = *reinterpret_cast<const h8_t*>(p.in.hi + xoff + k_);                \
      if constexpr (X3) nxt[u].xl = *reinterpret_cast<const h8_t*>(p.in.lo + xoff + k_); \
    }                                                                                 \
  }
  VP_MBB_LOAD(0)
  // epilogue items of this thread: item tid + NT t = (pixel tile it >> 8, pixel (it & 255) >> 3 of the tile, channels 4 (it & 7) .. + 4);
  // the residual and the bias travel under the gate phases
  typedef _Float16 h4_t __attribute__((ext_vector_type(4)));
  h4_t r_hi[IPT], r_lo[IPT];
  const int ec = co0 + 4 * (tid & 7);
#pragma unroll
  for (int t = 0; t < IPT; ++t) {
    const int it = tid + NT * t, px = pxw + 32 * (it >> 8) + ((it & 255) >> 3);
    r_hi[t] = r_lo[t] = h4_t{0, 0, 0, 0};
    if (p.res.hi && it < T::ITEMS && px < M) {
      r_hi[t] = *reinterpret_cast<const h4_t*>(p.res.hi + (size_t)px * Cout + ec);
      if constexpr (X3) r_lo[t] = *reinterpret_cast<const h4_t*>(p.res.lo + (size_t)px * Cout + ec);
    }
  }
  const f32x4_t bias4 = *reinterpret_cast<const f32x4_t*>(p.bias + ec);
  const f32x4_t ws4 = *reinterpret_cast<const f32x4_t*>(p.wscale + ec);   // 2^-prescale of the projection rows: exact product

  // ---- gate: means, squeeze FC (shared with se_gate_scale_kernel), excite FC + sigmoid for ALL channels
  if constexpr (ABL & 1) {
    for (int c = tid; c < C; c += NT) gate[c] = 0.5f;
  } else {
    // WM = 4: C <= 160 and the register budget is 168 (three workgroups per CU).  WM = 1: one workgroup per CU, 256 registers: ALL of a
    // thread's FC weights are requested at once.
    if (p.zsums) {
      // the squeeze FC arrived with the pool (mbconv_front): replica rows -> sq numbers, SiLU
      if (tid < 64) {
        float v = 0.0f;
        if (tid < p.se.sq) {
          long long t = 0;
#pragma unroll 8
          for (int r = 0; r < p.se.replicas; ++r) t += (long long)p.zsums[(size_t)r * 64 + tid];
          v = silu_f((float)((double)t * (1.0 / 16777216.0)) * p.se.inv_hw + p.se.b1[tid]);
        }
        s1[tid] = v;
      }
    } else {
    if constexpr (ABL & 4) {
      for (int c = tid; c < C; c += NT) mean[c] = 0.25f;
      __syncthreads();
    } else {
      se_means<NT>(p.se, accs, mean);
    }
    if constexpr (ABL & 8) {
      if (tid < 64) s1[tid] = tid < p.se.sq ? 0.3f : 0.0f;
    } else {
      // squeeze FC, COALESCED: a wave owns units wave, wave + NW, ...; its lanes walk a unit's row 16 bytes apiece (1 KB per load
      // instruction: 8 full lines -- a thread walking its own segment of a row touches 64 lines per instruction, one line a cycle:
      // 6.6 us of a 1152-channel block, tools/mbb_check.hip), butterfly sum over the lanes (fixed order)
      constexpr int MU = WM == 4 ? 4 : 6, MQ = WM == 4 ? 1 : 5;   // units per wave, 64-lane passes over a row, per batch
      const int C4 = C >> 2;
      const f32x4_t* m4 = reinterpret_cast<const f32x4_t*>(mean);
      for (int j0 = wave; j0 < p.se.sq; j0 += NW * MU) {
        float sj[MU];
#pragma unroll
        for (int u = 0; u < MU; ++u) sj[u] = 0.f;
        for (int q0 = lane; q0 < C4; q0 += 64 * MQ) {
          f32x4_t a[MU][MQ];
#pragma unroll
          for (int u = 0; u < MU; ++u)
#pragma unroll
            for (int i = 0; i < MQ; ++i)
              if (j0 + NW * u < p.se.sq && q0 + 64 * i < C4) a[u][i] = reinterpret_cast<const f32x4_t*>(p.se.w1 + (size_t)(j0 + NW * u) * C)[q0 + 64 * i];
#pragma unroll
          for (int u = 0; u < MU; ++u)
#pragma unroll
            for (int i = 0; i < MQ; ++i)
              if (j0 + NW * u < p.se.sq && q0 + 64 * i < C4) {
                const f32x4_t m = m4[q0 + 64 * i];
                sj[u] += (a[u][i][0] * m[0] + a[u][i][1] * m[1]) + (a[u][i][2] * m[2] + a[u][i][3] * m[3]);
              }
        }
#pragma unroll
        for (int u = 0; u < MU; ++u) {
          float t = sj[u];
#pragma unroll
          for (int o = 32; o > 0; o >>= 1) t += __shfl_xor(t, o);
          if (lane == 0 && j0 + NW * u < p.se.sq) s1[j0 + NW * u] = silu_f(t + p.se.b1[j0 + NW * u]);
        }
      }
      if (tid >= p.se.sq && tid < 64) s1[tid] = 0.0f;
    }
    }
    __syncthreads();
    const int nq = p.sqp >> 2;
    const f32x4_t* s4 = reinterpret_cast<const f32x4_t*>(s1);
    constexpr int EQ = WM == 4 ? 4 : 12;   // weights (x 16 bytes) per channel and batch: sq <= 16 on the big maps, <= 48 anywhere in the encoder
    constexpr int CI = WM == 4 ? 1 : 3;    // channels per thread and batch (C <= 3 x 512 in one batch)
    for (int cb = tid; cb < C; cb += NT * CI) {
      if constexpr (ABL & 16) {
        gate[cb] = 0.5f * s1[cb & 7];
        if (CI > 1) break;
        continue;
      }
      float s[CI];
#pragma unroll
      for (int i = 0; i < CI; ++i) s[i] = 0.f;
      for (int q0 = 0; q0 < nq; q0 += EQ) {
        f32x4_t a[CI][EQ];
#pragma unroll
        for (int i = 0; i < CI; ++i)
#pragma unroll
          for (int q = 0; q < EQ; ++q)
            if (q0 + q < nq && cb + NT * i < C) a[i][q] = reinterpret_cast<const f32x4_t*>(p.w2q)[(size_t)(q0 + q) * C + cb + NT * i];
#pragma unroll
        for (int i = 0; i < CI; ++i)
#pragma unroll
          for (int q = 0; q < EQ; ++q)
            if (q0 + q < nq && cb + NT * i < C) {
              const f32x4_t m = s4[q0 + q];
              s[i] += (a[i][q][0] * m[0] + a[i][q][1] * m[1]) + (a[i][q][2] * m[2] + a[i][q][3] * m[3]);
            }
      }
#pragma unroll
      for (int i = 0; i < CI; ++i) {
        const int c = cb + NT * i;
        if (c < C) gate[c] = c < p.se.Creal ? sigmoid_f(s[i] + p.b2[c]) : 0.0f;
      }
    }
  }
  __syncthreads();

  // ---- projection: A = scaled weights (32 output channels x 16 k), B = pixels (32 x 16 k); a lane's 16 accumulators are channels
  // 8 g + 4 (lane >> 5) + r (g, r = 0..3) of pixel lane & 31
  f32x16_t acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
  for (int s = 0; s < ((ABL & 2) ? 0 : nst); s += U) {
    MbBackFrag cur[U];
#pragma unroll
    for (int u = 0; u < U; ++u) cur[u] = nxt[u];
    if (s + U < nst) VP_MBB_LOAD(s + U)
#pragma unroll
    for (int u = 0; u < U; ++u) {
      if (s + u < nst) {
        const float* gk = gate + (ks0 + s + u) * 16 + 8 * (lane >> 5);
        const f32x4_t g0 = *reinterpret_cast<const f32x4_t*>(gk), g1 = *reinterpret_cast<const f32x4_t*>(gk + 4);
        h8_t a_hi, a_lo;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          float v = (i < 4 ? cur[u].w0[i] * g0[i] : cur[u].w1[i - 4] * g1[i - 4]);
          // v must be ONE fp32 value for both planes: left to itself the backend folds the product into the hi conversion
          // (v_fma_mixlo_f16: the exact product rounded once) but takes lo against the fp32 product, and the planes of a few elements in
          // 2^12 disagree by one fp16 ulp (1e-4 of a layer's output, tools/mbb_check.hip).  The empty asm pins v in a register.
          asm volatile("" : "+v"(v));
          a_hi[i] = (half_t)v;
          if constexpr (X3) a_lo[i] = (half_t)(v - (float)a_hi[i]);
        }
        if constexpr (X3) {
          acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a_lo, cur[u].xh, acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a_hi, cur[u].xl, acc, 0, 0, 0);
        }
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a_hi, cur[u].xh, acc, 0, 0, 0);
      }
    }
  }
#undef VP_MBB_LOAD
  // ---- the K shares of a pixel tile meet in LDS (fixed order), bias, residual, (hi, lo) store
  {
    unsigned char* const mine = rb + wave * 32 * RP + (lane & 31) * RP + 16 * (lane >> 5);
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      f32x4_t v;
#pragma unroll
      for (int r = 0; r < 4; ++r) v[r] = acc[4 * g + r];
      *reinterpret_cast<f32x4_t*>(mine + 32 * g) = v;
    }
  }
  __syncthreads();
#pragma unroll
  for (int t = 0; t < IPT; ++t) {
    const int it = tid + NT * t, tile = it >> 8, px = pxw + 32 * tile + ((it & 255) >> 3);
    if (it >= T::ITEMS || px >= M) continue;
    const unsigned char* src = rb + tile * 32 * RP + ((it & 255) >> 3) * RP + 16 * (it & 7);   // wave `tile` = share 0 of that tile
    f32x4_t v = *reinterpret_cast<const f32x4_t*>(src);
#pragma unroll
    for (int j = 1; j < NSH; ++j) {
      const f32x4_t q = *reinterpret_cast<const f32x4_t*>(src + j * WM * 32 * RP);
#pragma unroll
      for (int r = 0; r < 4; ++r) v[r] += q[r];
    }
    h4_t o_hi, o_lo;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      float y = fmaf(v[r], ws4[r], bias4[r]);
      if (p.res.hi) y += X3 ? (float)r_hi[t][r] + (float)r_lo[t][r] : (float)r_hi[t][r];
      o_hi[r] = (half_t)y;
      if constexpr (X3) o_lo[r] = (half_t)(y - (float)o_hi[r]);
    }
    *reinterpret_cast<h4_t*>(p.out.hi + (size_t)px * Cout + ec) = o_hi;
    if constexpr (X3) *reinterpret_cast<h4_t*>(p.out.lo + (size_t)px * Cout + ec) = o_lo;
  }
}

bool mbconv_back_supported(const MbBackParams& p) {
  const SeParams& se = p.se;
  const bool x3 = p.in.lo != nullptr;   // one plane everywhere (VP_FP16) or two everywhere (VP_FP16X3)
  return p.in.hi && p.out.hi && (x3 ? p.out.lo != nullptr : p.out.lo == nullptr) && se.sums && se.w1 && se.b1 && p.w2q && p.b2 && p.w && p.bias && p.wscale && se.frames <= 1 &&
         se.C == p.in.C && (se.C & 31) == 0 && se.C >= 32 && (p.out.C & 31) == 0 && se.sq >= 1 && se.sq <= 64 && p.sqp >= se.sq && (p.sqp & 3) == 0 && p.sqp <= 64 &&
         se.replicas >= 1 && p.out.H == p.in.H && p.out.W == p.in.W && p.in.H * p.in.W >= 1 &&
         (!p.res.hi || ((x3 ? p.res.lo != nullptr : p.res.lo == nullptr) && p.res.C == p.out.C && p.res.H == p.out.H && p.res.W == p.out.W));
}

template <int WM, int NW, int ABL = 0>
static hipError_t launch_mbb(const MbBackParams& p, hipStream_t st) {
  using T = MbBack<WM, NW>;
  const int M = p.in.H * p.in.W;
  if (p.in.lo == nullptr) {
    hipLaunchKernelGGL((mbconv_back_kernel<WM, NW, ABL, false>), dim3((M + 32 * WM - 1) / (32 * WM), p.out.C / 32), dim3(T::NT), T::lds(p.se.C), st, p);
    return hipGetLastError();
  }
  hipLaunchKernelGGL((mbconv_back_kernel<WM, NW, ABL>), dim3((M + 32 * WM - 1) / (32 * WM), p.out.C / 32), dim3(T::NT), T::lds(p.se.C), st, p);
  return hipGetLastError();
}

hipError_t launch_mbconv_back(const MbBackParams& p, hipStream_t st) {
  if (!mbconv_back_supported(p)) return hipErrorInvalidValue;
  // 80x160 and 160x320 (K = 2 ... 10 steps): a wave per pixel tile, four waves; the smaller maps: eight waves, an eighth of K each.  (64 pixels
  // x eight waves on the 80x160 maps measured: 7.1 / 7.4 us against 6.3 / 8.7 for K = 96 / 160 -- a wash, not kept: tools/mbb_check.hip.)
  return p.in.H * p.in.W >= 12800 ? launch_mbb<4, 4>(p, st) : launch_mbb<1, 8>(p, st);
}

}  // namespace vp
