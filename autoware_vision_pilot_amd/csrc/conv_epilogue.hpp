// Shared epilogue of the MFMA conv kernels.
//
// The MFMA result layout gives every lane 4 consecutive output channels of ONE pixel per register group; storing
// that directly means 8-byte pieces scattered over 32 pixels per store instruction (measured: the memory-bound
// upsample / skip layers ran at 0.3-0.8 TB/s).  Instead the accumulators go through LDS:
//   phase 1 (fully unrolled, a few dozen ds_write_b128): fp32 accumulators -> stage[pixel][channel]
//   phase 2 (compact runtime loop): each lane takes 8 consecutive channels of one pixel: bias + activation +
//            residual in fp32, (hi, lo) split, ONE 16-byte store per plane; 8 lanes cover 128 contiguous bytes.
// Keeping phase 2 out of the unrolled code also keeps the kernels small enough for full unrolling, which is what
// keeps the accumulator array in registers (a partially unrolled epilogue pushed it to scratch memory).
// Phase 2 is instantiated per (store mode, residual mode, activation) combination the networks use, so the row
// loop carries no mode branches (the all-runtime version was ~1600 instructions per pass; SQ_ACTIVE_INST showed
// waves of the memory-bound layers spending 40 % of their life issuing it).
#pragma once
#include <atomic>
#include <type_traits>

#include "kernels.hpp"

namespace vp {

// Class map of one pixel from its logits v[0..Creal) (RunModelNode::onImage argmax / threshold loops, run_model_node.cpp:144-171;
// createMaskFromTensor{CUDA,HIP}; createEgoLanesMaskFromTensorCUDA) -- decode_mask_kernel's rules (kernels_misc.hip), same bits.
// NV: how many entries of v exist (compile-time bound of the argmax loop).
template <int NV>
__device__ __forceinline__ uint8_t decode_pixel_n(const float* v, int Creal, int mode) {
  if (mode == 1) return v[2 < NV ? 2 : 0] > 0.0f ? 2 : (v[1 < NV ? 1 : 0] > 0.0f ? 1 : (v[0] > 0.0f ? 0 : 255));
  if (Creal > 1) {
    float best = -1e9f;
    int cls = 0;
#pragma unroll
    for (int c = 0; c < NV; ++c)
      if (c < Creal && v[c] > best) {
        best = v[c];
        cls = c;
      }
    return mode == 2 ? (uint8_t)cls : (cls == 1 ? 255 : 0);
  }
  return v[0] > 0.0f ? 255 : 0;
}
__device__ __forceinline__ uint8_t decode_pixel(const float (&v)[4], int Creal, int mode) { return decode_pixel_n<4>(v, Creal, mode); }

// bias + activation + residual + split + store for 8 consecutive output channels of pixel m.
// STORE / RES / ACT >= 0 fix the mode at compile time; -1 reads it from the parameter block.
template <int STORE = -1, int RES = -1, int ACT = -1>
__device__ __forceinline__ void epilogue_store8(const ConvGemmParams& p, int M, int m, int co, float v[8], const f32x4_t& b0,
                                                const f32x4_t& b1, const f32x4_t& s0, const f32x4_t& s1, long long o_pre = -1) {
  const int act = ACT >= 0 ? ACT : p.act;
  const int res_mode = RES >= 0 ? RES : p.res_mode;
  const int store_mode = STORE >= 0 ? STORE : p.store_mode;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    v[r] = apply_act(fmaf(v[r], s0[r], b0[r]), act);   // s = 2^-prescale of the weight row (ConvGemmParams::wscale): exact
    v[4 + r] = apply_act(fmaf(v[4 + r], s1[r], b1[r]), act);
  }
  if (store_mode == STORE_NCHW_F32) {
#pragma unroll
    for (int r = 0; r < 8; ++r)
      if (co + r < p.Creal) p.out_f32[(size_t)(co + r) * M + m] = v[r];
    // fused decode (RunModelNode::onImage argmax / threshold loops, run_model_node.cpp:144-171; createMaskFromTensor{CUDA,HIP};
    // createEgoLanesMaskFromTensorCUDA): this lane holds ALL of the pixel's logits (the heads have <= 3 channels), exactly the
    // values it just stored -- same rules, same bits as decode_mask_kernel on the stored tensor
    if (p.mask_out != nullptr && co == 0 && p.Creal <= 8) p.mask_out[m] = decode_pixel_n<8>(v, p.Creal, p.decode_mode);
    return;
  }
  size_t o;
  if (o_pre >= 0) {
    o = (size_t)o_pre;  // caller tracked the output element offset incrementally (no divisions in the row loop)
  } else if (store_mode == STORE_SHUFFLE2) {
    const int q = co / p.Cstore, c = co - q * p.Cstore;
    const int y = m / p.W, x = m - y * p.W;
    o = ((size_t)(2 * y + (q >> 1)) * (2 * p.W) + (2 * x + (q & 1))) * p.Cstore + c;
  } else {
    o = (size_t)m * p.Cstore + co;
  }
  if (res_mode != RES_NONE) {
    const h8_t rh = *reinterpret_cast<const h8_t*>(p.res_hi + o);
    float r8[8];
#pragma unroll
    for (int r = 0; r < 8; ++r) r8[r] = (float)rh[r];
    if (p.res_lo) {
      const h8_t rl = *reinterpret_cast<const h8_t*>(p.res_lo + o);
#pragma unroll
      for (int r = 0; r < 8; ++r) r8[r] += (float)rl[r];
    }
#pragma unroll
    for (int r = 0; r < 8; ++r) v[r] = (res_mode == RES_ADD) ? (v[r] + r8[r]) : (v[r] * r8[r] + r8[r]);
  }
  if (STORE < 0 && p.post_act != ACT_NONE) {  // only the all-runtime instantiation carries it (the engine routes such ops there)
#pragma unroll
    for (int r = 0; r < 8; ++r) v[r] = apply_act(v[r], p.post_act);
  }
  h8_t hi;
#pragma unroll
  for (int r = 0; r < 8; ++r) hi[r] = (half_t)v[r];
  *reinterpret_cast<h8_t*>(p.out_hi + o) = hi;
  if (p.out_lo) {
    h8_t lo;
#pragma unroll
    for (int r = 0; r < 8; ++r) lo[r] = (half_t)(v[r] - (float)hi[r]);
    *reinterpret_cast<h8_t*>(p.out_lo + o) = lo;
  }
}

// Pixel maps: local pixel index of the workgroup tile -> linear pixel m of the image, or -1 outside.
struct PixLinear {
  int m0, M;
  __device__ __forceinline__ int operator()(int q) const {
    const int m = m0 + q;
    return m < M ? m : -1;
  }
};
// Lane -> pixel map of a 32-pixel MFMA column tile over a 2 x 16 pixel patch.  ds_read_b128 serves a wave in the
// fixed 16-lane groups {0-3,12-15,20-27} / {4-11,16-19,28-31}; giving each group ONE 16-pixel row makes its 16
// addresses consecutive halo pixels, which the 80-byte pitch spreads over all 16 bank slots (the natural
// lanes 0-15 = row 0 / 16-31 = row 1 map is 2-way conflicted: measured 41 % of LDS cycles).
__device__ __forceinline__ void lane_to_px16(int c, int& rowbit, int& px) {
  if (c < 4) { rowbit = 0; px = c; }
  else if (c < 12) { rowbit = 1; px = c - 4; }
  else if (c < 16) { rowbit = 0; px = c - 8; }
  else if (c < 20) { rowbit = 1; px = c - 8; }
  else if (c < 28) { rowbit = 0; px = c - 12; }
  else { rowbit = 1; px = c - 16; }
}
template <int TW>
struct PixPatch {
  static_assert(TW == 16, "patch kernels use 16-pixel-wide tiles");
  int y0, x0, H, W;
  __device__ __forceinline__ int operator()(int q) const {
    int rowbit, px;
    lane_to_px16(q & 31, rowbit, px);
    const int y = y0 + 2 * (q >> 5) + rowbit, x = x0 + px;
    return (y < H && x < W) ? y * W + x : -1;
  }
};

// phase 2 row loop, one instantiation per mode combination (see header comment)
template <int STORE, int RES, int ACT, int PXT, int RPI, int PITCH, class PixMap>
__device__ __forceinline__ void epilogue_rows(const ConvGemmParams& p, const char* stage, int r0, int c8, int co, const PixMap& pix,
                                              int M) {
  const f32x4_t b0 = *reinterpret_cast<const f32x4_t*>(p.bias + co), b1 = *reinterpret_cast<const f32x4_t*>(p.bias + co + 4);
  const f32x4_t s0 = *reinterpret_cast<const f32x4_t*>(p.wscale + co), s1 = *reinterpret_cast<const f32x4_t*>(p.wscale + co + 4);
  if constexpr (STORE == STORE_SHUFFLE2 && std::is_same<PixMap, PixLinear>::value) {
    // Pixel-shuffle store over a linear pixel tile: the two integer divisions (quadrant of the channel, row of the
    // pixel) are done ONCE per thread; the row loop then advances (y, x) and the output offset incrementally.  The
    // division-per-row version cost ~110 VALU instructions per 16-byte store and made the up-sampling GEMMs
    // VALU-bound (37 us for 65 MB of traffic).
    const int q = co / p.Cstore, c = co - q * p.Cstore;
    int m = pix.m0 + r0;
    int y = m / p.W, x = m - y * p.W;
    const int W2 = 2 * p.W;
    for (int r = r0; r < PXT && m < pix.M; r += RPI) {
      const long long o = ((long long)(2 * y + (q >> 1)) * W2 + (2 * x + (q & 1))) * p.Cstore + c;
      const f32x4_t q0 = *reinterpret_cast<const f32x4_t*>(stage + r * PITCH + c8 * 32);
      const f32x4_t q1 = *reinterpret_cast<const f32x4_t*>(stage + r * PITCH + c8 * 32 + 16);
      float v[8] = {q0[0], q0[1], q0[2], q0[3], q1[0], q1[1], q1[2], q1[3]};
      epilogue_store8<STORE, RES, ACT>(p, M, m, co, v, b0, b1, s0, s1, o);
      m += RPI;
      x += RPI;
      while (x >= p.W) {
        x -= p.W;
        ++y;
      }
    }
  } else {
#pragma unroll 2
    for (int r = r0; r < PXT; r += RPI) {
      const int m = pix(r);
      if (m < 0) continue;
      const f32x4_t q0 = *reinterpret_cast<const f32x4_t*>(stage + r * PITCH + c8 * 32);
      const f32x4_t q1 = *reinterpret_cast<const f32x4_t*>(stage + r * PITCH + c8 * 32 + 16);
      float v[8] = {q0[0], q0[1], q0[2], q0[3], q1[0], q1[1], q1[2], q1[3]};
      epilogue_store8<STORE, RES, ACT>(p, M, m, co, v, b0, b1, s0, s1);
    }
  }
}

// One epilogue pass over the channel slice [co_base, co_base + 32*WCO) of a PXT-pixel workgroup tile.
//   accs[j] : this wave's NT accumulator tiles of the pass (32 channels x 32 pixels each, MFMA C layout:
//             lane&31 = pixel, register 4g+r = channel 8g + 4*(lane>>5) + r)
//   wco,wpx : wave coordinates; the wave's channel sub-slice is co_base + wco*32, its pixels (wpx*NT + j)*32 + lane&31.
// Caller guarantees all LDS reads of the main loop are done (a barrier was passed) and calls this from all 256 threads.
template <int PXT, int WCO, int NT, class PixMap>
__device__ __forceinline__ void epilogue_pass(const ConvGemmParams& p, char* stage, const f32x16_t (&accs)[NT], int co_base, int wco,
                                              int wpx, const PixMap& pix, int M, int zsplit) {
  constexpr int COP = 32 * WCO;           // channels per pass
  constexpr int PITCH = COP * 4 + 16;     // bytes per staged pixel row (+16: conflict-free ds_write_b128)
  constexpr int CPR = COP / 8;            // 8-channel pieces per row
  constexpr int RPI = 256 / CPR;          // rows per iteration of phase 2
  const int tid = threadIdx.x, lane = tid & 63;
  // ---- phase 1
#pragma unroll
  for (int j = 0; j < NT; ++j) {
    char* row = stage + ((wpx * NT + j) * 32 + (lane & 31)) * PITCH + (wco * 32 + 4 * (lane >> 5)) * 4;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      f32x4_t v = {accs[j][4 * g + 0], accs[j][4 * g + 1], accs[j][4 * g + 2], accs[j][4 * g + 3]};
      *reinterpret_cast<f32x4_t*>(row + g * 32) = v;
    }
  }
  __syncthreads();
  // ---- phase 2
  const int c8 = tid % CPR, r0 = tid / CPR;
  const int co = co_base + c8 * 8;
  if (co < p.Ncols) {
    if (p.nsplit > 1) {
      for (int r = r0; r < PXT; r += RPI) {
        const int m = pix(r);
        if (m < 0) continue;
        float* dst = p.partial + ((size_t)zsplit * M + m) * p.CoutW + co;
        *reinterpret_cast<f32x4_t*>(dst) = *reinterpret_cast<const f32x4_t*>(stage + r * PITCH + c8 * 32);
        *reinterpret_cast<f32x4_t*>(dst + 4) = *reinterpret_cast<const f32x4_t*>(stage + r * PITCH + c8 * 32 + 16);
      }
    } else {
      const int key = p.post_act != ACT_NONE ? -1 : p.store_mode * 100 + p.res_mode * 10 + p.act;
#define VP_ROWS(S, R, A) epilogue_rows<S, R, A, PXT, RPI, PITCH>(p, stage, r0, c8, co, pix, M)
      switch (key) {
        case STORE_NHWC * 100 + RES_NONE * 10 + ACT_GELU: VP_ROWS(STORE_NHWC, RES_NONE, ACT_GELU); break;   // decoder 3x3
        case STORE_NHWC * 100 + RES_NONE * 10 + ACT_SILU: VP_ROWS(STORE_NHWC, RES_NONE, ACT_SILU); break;   // MBConv expand
        case STORE_NHWC * 100 + RES_NONE * 10 + ACT_GELU_F16: VP_ROWS(STORE_NHWC, RES_NONE, ACT_GELU_F16); break;
        case STORE_NHWC * 100 + RES_NONE * 10 + ACT_SILU_F16: VP_ROWS(STORE_NHWC, RES_NONE, ACT_SILU_F16); break;
        case STORE_NHWC * 100 + RES_NONE * 10 + ACT_NONE: VP_ROWS(STORE_NHWC, RES_NONE, ACT_NONE); break;   // MBConv project
        case STORE_NHWC * 100 + RES_ADD * 10 + ACT_NONE: VP_ROWS(STORE_NHWC, RES_ADD, ACT_NONE); break;     // residual / skip link
        case STORE_SHUFFLE2 * 100 + RES_NONE * 10 + ACT_NONE: VP_ROWS(STORE_SHUFFLE2, RES_NONE, ACT_NONE); break;  // ConvTranspose
        case STORE_NCHW_F32 * 100 + RES_NONE * 10 + ACT_NONE: VP_ROWS(STORE_NCHW_F32, RES_NONE, ACT_NONE); break;  // logits
        default: VP_ROWS(-1, -1, -1); break;                                                                 // e.g. ctx mul-add
      }
#undef VP_ROWS
    }
  }
  __syncthreads();
}

template <int PXT, int WCO>
constexpr int epilogue_stage_bytes() {
  return PXT * (32 * WCO * 4 + 16);
}

// Single-pass REGISTER epilogue for the VP_FP16 hot cases (one fp16 plane, bias + {none, GELU, SiLU}, no residual, no
// split-K; NHWC or pixel-shuffle store): bias + activation + fp16 conversion happen in registers on the whole
// accumulator set, the fp16 tile is staged ONCE in LDS as [pixel][CO channels] and leaves as 16-byte row pieces
// (CO*2 bytes contiguous per pixel).  Versus the generic fp32-staged MT-pass epilogue: half the LDS traffic, one
// barrier pair instead of MT, ~12 instead of ~35 instructions per 16-byte store -- the short-K GEMMs (ConvTranspose:
// K = 128..544) spent more issue slots in the epilogue than in MFMAs, and VALU does not overlap MFMA on this part.
// ACC_AGPR: the 3x3 kernels keep accumulators in AGPRs; reading them one group at a time (inline v_accvgpr_read)
// stops the compiler from bursting all of them into VGPRs after the main loop (which halved occupancy).
template <int PXT, int CO, int WCO, int MT, int NT, int ACT, int STORE, bool ACC_AGPR, class PixMap>
__device__ __forceinline__ void epilogue_regs_fp16(const ConvGemmParams& p, char* stage, f32x16_t (&acc)[MT][NT], int co0, int wco,
                                                   int wpx, const PixMap& pix) {
  constexpr int PITCH = CO * 2 + 16;
  const int tid = threadIdx.x, lane = tid & 63;
  __syncthreads();  // the stage aliases the main loop's LDS buffers
#pragma unroll
  for (int i = 0; i < MT; ++i) {
    const int cl = (i * WCO + wco) * 32 + 4 * (lane >> 5);  // channel (within the CO tile) of register group g=0, r=0
    f32x4_t b[4], sc[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      b[g] = *reinterpret_cast<const f32x4_t*>(p.bias + co0 + cl + 8 * g);
      sc[g] = *reinterpret_cast<const f32x4_t*>(p.wscale + co0 + cl + 8 * g);
    }
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      char* row = stage + ((wpx * NT + j) * 32 + (lane & 31)) * PITCH + cl * 2;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        h4_t h;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float v;
          if constexpr (ACC_AGPR) {
            asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(v) : "a"(acc[i][j][4 * g + r]));
          } else {
            v = acc[i][j][4 * g + r];
          }
          h[r] = (half_t)apply_act(fmaf(v, sc[g][r], b[g][r]), ACT);
        }
        *reinterpret_cast<h4_t*>(row + g * 16) = h;
        if constexpr (ACT != ACT_NONE) __builtin_amdgcn_sched_barrier(0);  // keep activation chains from interleaving (VGPR pressure)
      }
    }
  }
  __syncthreads();
  constexpr int CPR = CO / 8, RPI = 256 / CPR;
  static_assert(256 % CPR == 0 && PXT % RPI == 0, "row loop shape");
  const int c8 = tid % CPR, r0 = tid / CPR;
  const int co = co0 + c8 * 8;
  if (co >= p.Ncols) return;
  if constexpr (STORE == STORE_SHUFFLE2 && std::is_same<PixMap, PixLinear>::value) {
    // pixel-shuffle store: quadrant and pixel row divided ONCE, then (y, x) advance incrementally
    const int q = co0 / p.Cstore, c = co - q * p.Cstore;  // the CO tile lies inside one quadrant (launcher checks)
    int m = pix.m0 + r0;
    int y = m / p.W, x = m - y * p.W;
    const int W2 = 2 * p.W;
    for (int r = r0; r < PXT && m < pix.M; r += RPI) {
      const size_t o = ((size_t)(2 * y + (q >> 1)) * W2 + (2 * x + (q & 1))) * p.Cstore + c;
      *reinterpret_cast<h8_t*>(p.out_hi + o) = *reinterpret_cast<const h8_t*>(stage + r * PITCH + c8 * 16);
      m += RPI;
      x += RPI;
      while (x >= p.W) {
        x -= p.W;
        ++y;
      }
    }
  } else {
    static_assert(STORE == STORE_NHWC, "register epilogue stores NHWC or pixel-shuffled NHWC");
#pragma unroll 4
    for (int r = r0; r < PXT; r += RPI) {
      const int m = pix(r);
      if (m < 0) continue;
      *reinterpret_cast<h8_t*>(p.out_hi + (size_t)m * p.Cstore + co) = *reinterpret_cast<const h8_t*>(stage + r * PITCH + c8 * 16);
    }
  }
}
// which (activation, store) combinations have a register-epilogue instantiation; 0 = none, else a small case id
__host__ __device__ inline int regepi_case(const ConvGemmParams& p, int co_tile, bool split) {
  if (split || p.out_lo || p.res_mode != RES_NONE || p.nsplit != 1 || p.post_act != ACT_NONE) return 0;
  if (p.store_mode == STORE_NHWC) {
    if (p.act == ACT_GELU_F16) return 1;
    if (p.act == ACT_SILU_F16) return 2;
    if (p.act == ACT_NONE) return 3;
    return 0;
  }
  if (p.store_mode == STORE_SHUFFLE2 && p.act == ACT_NONE && p.Cstore % co_tile == 0) return 4;
  return 0;
}
template <int PXT, int CO>
constexpr int epilogue_fp16_stage_bytes() {
  return PXT * (CO * 2 + 16);
}

hipError_t launch_splitk_finish(const ConvGemmParams& p, hipStream_t st);  // kernels_conv.hip

// hipFuncAttributeMaxDynamicSharedMemorySize is a PER-DEVICE function attribute: an engine on a second gpu_id in the same
// process (thread-per-GPU hosts) must set it again.  One bit per device id, set after the attribute call succeeded; racing
// first launches from two threads both make the (idempotent) call.  Read-mostly atomic, no lock on the launch path.
struct LdsAttrOnce {
  std::atomic<unsigned long long> done[4] = {};
};
inline hipError_t set_max_dynamic_lds(LdsAttrOnce& once, const void* fn, int lds_bytes) {
  int dev = 0;
  hipError_t e = hipGetDevice(&dev);
  if (e != hipSuccess) return e;
  std::atomic<unsigned long long>& word = once.done[(dev >> 6) & 3];
  const unsigned long long bit = 1ull << (dev & 63);
  if (word.load(std::memory_order_acquire) & bit) return hipSuccess;
  e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
  if (e != hipSuccess) return e;
  word.fetch_or(bit, std::memory_order_release);
  return hipSuccess;
}


}  // namespace vp
