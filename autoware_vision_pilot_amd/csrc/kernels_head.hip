// The LAST convolution of every head: 3x3, 64 or 128 channels -> 1..3 logit channels on the full-resolution map, fp32 NCHW
// logits + the decoded class map (scene_seg_head.py:19, scene_3d_head.py:22, domain_seg_head.py:19, ego_lanes_head.py:16).
//
// 0.5-0.7 GFLOP against 52-105 MB of input (parity mode): a byte mover.  Through the 32-channel tile of the halo kernel it
// ran at 0.8-0.95 TB/s (66 us SceneSeg, 110 us Scene3D): 29 of the 32 MFMA rows multiply zeros, the weight fragments are
// re-read from LDS for every pixel fragment (more LDS bytes than the pixels themselves), and the halo travels global ->
// registers -> LDS.  Here:
//   * v_mfma_f32_16x16x32_f16: 16 output rows (3 real) x 16 pixels x 32 channels per instruction -- twice the pixel x K volume
//     per matrix-pipe cycle of the 32x32x16 shape when the row dimension is mostly padding;
//   * the weights are STATIONARY IN REGISTERS: a wave owns one 64-channel slab = 18 (tap, 32-channel) A-fragments x (hi, lo) =
//     144 VGPRs, loaded once per launch (only the lanes of real rows load; the others hold zeros); the workgroup is persistent;
//   * the halo tile goes global -> LDS by LDS-DMA (coalesced, no VGPRs, no ds_write); pixels outside the image and the pad
//     slots are fetched from a ZERO PAGE in global memory -- the border needs no instruction at all;
//   * LDS pixel rows are pitched at 8 KS + 2 sixteen-byte slots: ds_read_b128 serves the fixed lane groups {0-3, 12-15, 20-27}
//     and {4-11, 16-19, 28-31} (conv_epilogue.hpp), i.e. eight pixels with channel-octet kg and eight with kg + 1; with the
//     pitch = 2 (mod 16) slots the first eight land on even slots, the others on odd ones, all sixteen distinct (pitch 9 is
//     2-way conflicted on seven of eight);
//   * Cin = 128: two waves per 64-channel slab, partial sums of slab 1 meet slab 0 in LDS (16 lanes x 16 bytes per pixel group);
//   * the 16 lanes that hold a pixel group's channels 0..3 add the bias, store the fp32 logits (64 contiguous bytes per channel)
//     and decode the class / lane label / threshold mask (RunModelNode::onImage argmax / threshold loops,
//     run_model_node.cpp:144-171; decode_mask_kernel's rules, same bits).
// Two workgroups per CU drift out of phase: one's DMA round trip runs under the other's MFMAs.
#include <cstdlib>

#include "conv_epilogue.hpp"
#include "lds_dma.hpp"

namespace vp {

typedef float f32x4_acc __attribute__((ext_vector_type(4)));

// KS: 64-channel slabs (Cin = 64 KS); tile = 8 / KS rows x 16 pixels; SPLIT: (hi, lo) planes
template <int KS, bool SPLIT>
__global__ __launch_bounds__(256, 2) void head_conv3x3_kernel(const ConvGemmParams p, const half_t* __restrict__ zeros) {
  constexpr int TH = 8 / KS, HWD = 18, HPX = (TH + 2) * HWD;
  constexpr int PC = 8 * KS + 2, PITCH = PC * 16;                 // slots per halo pixel: data + 2 pad, = 2 (mod 16) for KS = 1 (10) and 2 (18)
  constexpr int PLANES = SPLIT ? 2 : 1;
  constexpr int SLOTS = HPX * PC, NI = (SLOTS + 63) / 64, PLANE_BYTES = NI * 1024;
  constexpr int NINSTR = PLANES * NI, DPW = (NINSTR + 3) / 4;     // DMA instructions per tile, per wave (waves beyond the count skip)
  constexpr int KC = 18;                                          // (tap, 32-channel half) steps of one slab
  constexpr int NG = 2;                                           // pixel groups (tile rows) per wave
  static_assert(PC % 16 == 2 || PC % 16 == 10, "conflict-free pitch");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* const tile = smem;                                        // [PLANES][PLANE_BYTES]
  float* const red = reinterpret_cast<float*>(smem + PLANES * PLANE_BYTES);  // KS == 2: [TH groups][16 px][4]

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int slab = wave % KS, gsel = wave / KS;                   // this wave: channels [64 slab, 64 slab + 64), groups gsel + (4 / KS) i
  const int n = lane & 15, kg = lane >> 4;
  const int tiles_x = (p.W + 15) >> 4, tiles_y = (p.H + TH - 1) / TH, n_tiles = tiles_x * tiles_y;
  const int M = p.H * p.W;

  // ---- DMA plan: instruction ii = wave + 4 i moves slots [64 j, 64 j + 64) of plane ii / NI; slot -> (halo pixel, 16-byte piece)
  // (slot -> halo pixel / piece is recomputed per tile: divisions by constants, ~10 VALU per instruction -- the weights own the
  // register file, a stored plan spilled)
#define VP_DMA_TILE(TI)                                                                                         \
  {                                                                                                             \
    const int ty_ = (TI) / tiles_x, tx_ = (TI) - ty_ * tiles_x;                                                 \
    const int yb_ = ty_ * TH - 1, xb_ = tx_ * 16 - 1;                                                           \
    int lane_o_ = lane;                                                                                         \
    asm volatile("" : "+v"(lane_o_)); /* opaque per tile: keeps the slot arithmetic INSIDE the loop (hoisted, it spilled) */ \
    _Pragma("unroll") for (int i = 0; i < DPW; ++i) {                                                           \
      const int ii = wave + 4 * i;                                                                              \
      if (ii < NINSTR) {                                                                                        \
        const int slot_ = 64 * (ii >= NI ? ii - NI : ii) + lane_o_;                                             \
        const int hp_ = slot_ / PC, col_ = slot_ - hp_ * PC, hy_ = hp_ / HWD, hx_ = hp_ - hy_ * HWD;            \
        const int gy_ = yb_ + hy_, gx_ = xb_ + hx_;                                                             \
        const bool ok_ = slot_ < SLOTS && col_ < 8 * KS && (unsigned)gy_ < (unsigned)p.H && (unsigned)gx_ < (unsigned)p.W; \
        const half_t* base_ = (SPLIT && ii >= NI) ? p.in_lo : p.in_hi;                                          \
        const half_t* src_ = ok_ ? base_ + ((size_t)gy_ * p.W + gx_) * p.Cin + col_ * 8 : zeros; /* pad slots, pixels outside: zero page */ \
        VP_GLOBAL_LOAD_LDS16(src_, tile + ii * 1024);                                                           \
      }                                                                                                         \
    }                                                                                                           \
  }
  int t = blockIdx.x;
  if (t < n_tiles) VP_DMA_TILE(t)

  // ---- stationary weights.  Packing of the 32-channel halo tile (engine.cpp): element (co, tap, ci) at
  // (((ci >> 5) * 9 + tap) * CoutW + co) * 32 + (ci & 31); A-fragment of step c = (tap, half): row = lane & 15, 8 channels at 8 kg
  h8_t ah[KC], al[SPLIT ? KC : 1];
  {
    const bool real = n < p.Creal;
    const h8_t z8 = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
    for (int c = 0; c < KC; ++c) {
      const int tap = c >> 1, half = c & 1;
      const size_t o = ((size_t)((2 * slab + half) * 9 + tap) * p.CoutW + n) * 32 + kg * 8;
      ah[c] = real ? *reinterpret_cast<const h8_t*>(p.w_hi + o) : z8;
      if constexpr (SPLIT) al[c] = real ? *reinterpret_cast<const h8_t*>(p.w_lo + o) : z8;
    }
  }
  float bias4[4], ws4[4];   // ws4: 2^-prescale of the weight rows (ConvGemmParams::wscale), exact product
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    bias4[r] = r < p.Creal ? p.bias[r] : 0.0f;
    ws4[r] = r < p.Creal ? p.wscale[r] : 1.0f;
  }
  const int b_lane = n * PITCH + (slab * 8 + kg) * 16;            // + ((row + dy) * HWD + dx) * PITCH + half * 64

  for (; t < n_tiles; t += gridDim.x) {
    VP_WAIT_VMCNT(0);     // this wave's pieces of the tile have landed (and the previous tile's few stores)
    VP_LDS_BARRIER();     // everyone's have
    float v[NG][4];
    // one pixel group (tile row) at a time: three accumulator chains (the three products of the (hi, lo) pairs), pixel fragments
    // ONE step ahead of their MFMAs and no further (without the fences the scheduler hoists a dozen steps of ds_read_b128 and spills)
#pragma unroll
    for (int g = 0; g < NG; ++g) {
      f32x4_acc acc[3];
#pragma unroll
      for (int q = 0; q < 3; ++q) acc[q] = f32x4_acc{0.f, 0.f, 0.f, 0.f};
      const char* const gsrc = tile + b_lane + (gsel + (4 / KS) * g) * HWD * PITCH;
      h8_t fb[2], fbl[2];
#define VP_READ_B(SET, C)                                                                                     \
  {                                                                                                           \
    constexpr int tap_ = (C) >> 1, half_ = (C) & 1, dy_ = tap_ / 3, dx_ = tap_ - dy_ * 3;                     \
    fb[SET] = *reinterpret_cast<const h8_t*>(gsrc + (dy_ * HWD + dx_) * PITCH + half_ * 64);                  \
    if constexpr (SPLIT) fbl[SET] = *reinterpret_cast<const h8_t*>(gsrc + (dy_ * HWD + dx_) * PITCH + half_ * 64 + PLANE_BYTES); \
  }
#define VP_STEP(C)                                                                                            \
  {                                                                                                           \
    if constexpr ((C) + 1 < KC) VP_READ_B(((C) + 1) & 1, ((C) + 1 < KC ? (C) + 1 : 0))                         \
    __builtin_amdgcn_sched_barrier(0);                                                                        \
    if constexpr (SPLIT) {                                                                                    \
      acc[0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al[C], fb[(C) & 1], acc[0], 0, 0, 0);                   \
      acc[1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[C], fbl[(C) & 1], acc[1], 0, 0, 0);                  \
    }                                                                                                         \
    acc[2] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[C], fb[(C) & 1], acc[2], 0, 0, 0);                     \
    __builtin_amdgcn_sched_barrier(0);                                                                        \
  }
      VP_READ_B(0, 0)
      VP_STEP(0) VP_STEP(1) VP_STEP(2) VP_STEP(3) VP_STEP(4) VP_STEP(5) VP_STEP(6) VP_STEP(7) VP_STEP(8)
      VP_STEP(9) VP_STEP(10) VP_STEP(11) VP_STEP(12) VP_STEP(13) VP_STEP(14) VP_STEP(15) VP_STEP(16) VP_STEP(17)
#undef VP_STEP
#undef VP_READ_B
#pragma unroll
      for (int r = 0; r < 4; ++r) v[g][r] = SPLIT ? (acc[0][r] + acc[1][r]) + acc[2][r] : acc[2][r];
    }
    if constexpr (KS == 2) {  // slab 1's partial sums -> LDS (lanes 0..15 hold channels 0..3 of their pixel)
      if (slab == 1 && kg == 0) {
#pragma unroll
        for (int g = 0; g < NG; ++g) *reinterpret_cast<f32x4_t*>(red + ((gsel + (4 / KS) * g) * 16 + n) * 4) = f32x4_t{v[g][0], v[g][1], v[g][2], v[g][3]};
      }
    }
    VP_LDS_BARRIER();     // every wave is done with the tile (and slab 1's sums are visible)
    const int tcur = t, tnext = t + gridDim.x;
    if (tnext < n_tiles) VP_DMA_TILE(tnext)
    if (slab == 0 && kg == 0) {
      const int ty = tcur / tiles_x, tx = tcur - ty * tiles_x;
#pragma unroll
      for (int g = 0; g < NG; ++g) {
        const int row = gsel + (4 / KS) * g;
        const int y = ty * TH + row, x = tx * 16 + n;
        if (y >= p.H || x >= p.W) continue;
        float o[4];
        if constexpr (KS == 2) {
          const f32x4_t s = *reinterpret_cast<const f32x4_t*>(red + (row * 16 + n) * 4);
#pragma unroll
          for (int r = 0; r < 4; ++r) o[r] = fmaf(v[g][r] + s[r], ws4[r], bias4[r]);
        } else {
#pragma unroll
          for (int r = 0; r < 4; ++r) o[r] = fmaf(v[g][r], ws4[r], bias4[r]);
        }
        const int m = y * p.W + x;
#pragma unroll
        for (int r = 0; r < 4; ++r)
          if (r < p.Creal) p.out_f32[(size_t)r * M + m] = o[r];
        if (p.mask_out != nullptr) p.mask_out[m] = decode_pixel(o, p.Creal, p.decode_mode);
      }
    }
    if constexpr (KS == 2) {
      // `red` is rewritten after the NEXT tile's compute, i.e. behind the next iteration's first barrier, which slab 0 only
      // passes after these reads: no third barrier needed
    }
  }
#undef VP_DMA_TILE
}

namespace {
template <int KS, bool SPLIT>
hipError_t launch_head_cfg(const ConvGemmParams& p, const half_t* zeros, hipStream_t st) {
  constexpr int TH = 8 / KS, PC = 8 * KS + 2, NI = ((TH + 2) * 18 * PC + 63) / 64;
  constexpr int lds = (SPLIT ? 2 : 1) * NI * 1024 + (KS == 2 ? TH * 16 * 4 * 4 : 0);
  static_assert(lds <= 80 * 1024, "two workgroups per CU");
  auto k = head_conv3x3_kernel<KS, SPLIT>;
  static LdsAttrOnce attr_once;
  if (hipError_t e = set_max_dynamic_lds(attr_once, reinterpret_cast<const void*>(k), lds); e != hipSuccess) return e;
  const int n_tiles = ((p.W + 15) / 16) * ((p.H + TH - 1) / TH);
  hipLaunchKernelGGL(k, dim3(std::min(n_tiles, 512)), dim3(256), lds, st, p, zeros);
  return hipGetLastError();
}
}  // namespace

bool head_conv_supported(const ConvGemmParams& p) {
  return p.ks == 3 && p.stride <= 1 && p.store_mode == STORE_NCHW_F32 && p.act == ACT_NONE && p.res_mode == RES_NONE && p.post_act == ACT_NONE &&
         p.nsplit == 1 && p.Cin2 == 0 && (p.Cin == 64 || p.Cin == 128) && p.Creal >= 1 && p.Creal <= 4 && p.CoutW >= 16 && p.out_f32 != nullptr &&
         p.w_hi != nullptr;   // fp16 planes (an fp8-storage layer stays on the halo kernel)
}

// zeros: >= 16 bytes of zeros in device memory (the engine's zero page)
hipError_t launch_head_conv(const ConvGemmParams& p, const void* zeros, hipStream_t st) {
  if (!head_conv_supported(p) || zeros == nullptr) return hipErrorInvalidValue;
  const half_t* z = static_cast<const half_t*>(zeros);
  const bool split = p.in_lo != nullptr;
  if (p.Cin == 64) return split ? launch_head_cfg<1, true>(p, z, st) : launch_head_cfg<1, false>(p, z, st);
  return split ? launch_head_cfg<2, true>(p, z, st) : launch_head_cfg<2, false>(p, z, st);
}

}  // namespace vp
