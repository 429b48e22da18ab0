// Shared device/host definitions for libvp_hip (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace vp {

typedef _Float16 half_t;
typedef _Float16 h8_t __attribute__((ext_vector_type(8)));
typedef _Float16 h4_t __attribute__((ext_vector_type(4)));
typedef float f32x16_t __attribute__((ext_vector_type(16)));
typedef float f32x4_t __attribute__((ext_vector_type(4)));

// Activation tensor in HBM: NHWC, channels padded to a multiple of 32 (pad channels are always 0).
// Precision modes:
//   fp16   : value = hi
//   fp16x3 : value = hi + lo  (lo = fp16(x - fp16(x))): ~22 mantissa bits carried on the fp16 MFMA pipe;
//            a product a*b is evaluated as ahi*bhi + ahi*blo + alo*bhi with fp32 accumulation.
struct ActView {
  half_t* hi;
  half_t* lo;  // nullptr in fp16 mode
  int H, W, C;  // C = padded channel count
};

// ACT_*_F16 (base | 4): the VP_FP16 engines' variants -- results are rounded to fp16 (2^-11 relative) anyway, so the
// activation may trade fp32-class accuracy for VALU instructions (measured on gfx950: VALU work does NOT hide behind
// another wave's MFMAs, tools/mfma_valu_overlap.hip, so epilogue instructions are paid in full).  VP_FP16X3 (the
// parity mode) always uses the base codes.
enum ActFn { ACT_NONE = 0, ACT_GELU = 1, ACT_SILU = 2, ACT_SIGMOID = 3, ACT_F16 = 4, ACT_GELU_F16 = 5, ACT_SILU_F16 = 6, ACT_SIGMOID_F16 = 7,
             ACT_RELU = 8, ACT_TANH = 9, ACT_SILU2 = 10 /* SiLU(SiLU(x)): common_layers.py:211-213 */ };
enum ResMode { RES_NONE = 0, RES_ADD = 1, RES_MULADD = 2 };  // MULADD: out = v*res + res  (scene_context.py:56)
enum StoreMode { STORE_NHWC = 0, STORE_SHUFFLE2 = 1, STORE_NCHW_F32 = 2 };

// Implicit-GEMM convolution launch parameters (see kernels_conv.hip).
struct ConvGemmParams {
  const half_t* in_hi;
  const half_t* in_lo;
  int H, W, Cin;       // input spatial size, padded input channels (multiple of BK)
  const half_t* w_hi;  // [taps][CoutW][Cin]
  const half_t* w_lo;
  const float* bias;   // [CoutW]
  // [CoutW] 2^-s of the per-output-row power-of-two PRESCALE the engine folds into the weights at load (w_hi + w_lo = w * 2^s, row maximum in
  // [2^13, 2^14): neither fp16 plane of a weight is subnormal -- engine_internal.hpp prescale_exp); every epilogue evaluates
  // fmaf(acc, wscale[co], bias[co]): the product is exact (power of two), so the result is that of un-scaled weights carried exactly
  const float* wscale;
  // VP_WEIGHTS_FP8, real storage (round 4): when non-null the weights are OCP e4m3 BYTES in the kernel's own weight layout (one byte per
  // element where w_hi has a half), w_hi / w_lo are null, wscale[co] is the row's quantisation scale (not a power of two), and the weight
  // staging converts 8 codes -> 8 fp16 values on their way to LDS (exact: every e4m3 value is an fp16 value; the lo plane is zero).
  // Honoured by conv_gemm_kernel and conv3x3_halo_kernel (register-staged weights) -- every matrix layer of AutoDrive.
  const uint8_t* w8;
  int ks;              // 1 or 3 (stride 1, pad ks/2)
  int Ncols;           // GEMM columns to store (multiple of 32): Cout_pad, or 4*Cout_pad for STORE_SHUFFLE2
  int CoutW;           // weight rows allocated (multiple of the CO tile)
  int act, res_mode, store_mode;
  const half_t* res_hi;
  const half_t* res_lo;
  half_t* out_hi;
  half_t* out_lo;
  int Cstore;          // channel stride of the output tensor (padded C); for SHUFFLE2 Ncols == 4*Cstore
  float* out_f32;      // STORE_NCHW_F32: [Creal][H*W]
  int Creal;
  int nsplit;          // split-K factor (grid.z); >1 -> partial sums go to `partial`
  // K extension of the ConvTranspose GEMM (STORE_SHUFFLE2 only): the 1x1 skip-link conv of the SAME output pixel is
  // accumulated in the same pass.  K columns [Cin, Cin+Cin2) read a second tensor of size 2H x 2W x Cin2 at the
  // output pixel (2y+dy, 2x+dx) of the workgroup's quadrant; weight rows are Cin+Cin2 wide.  in2_delta_* = element
  // distance from in_hi/in_lo to that tensor's planes (one base pointer keeps the loads plain global loads).
  int Cin2;
  long long in2_delta_hi, in2_delta_lo;
  // strided 3x3 (AutoDrive Conv k3 s2, common_layers.py:5-14): H, W stay the OUTPUT size; the taps read the
  // Hin x Win input at (y*stride + dy, x*stride + dx).  stride <= 1 means 1 and Hin = H, Win = W.
  int stride, Hin, Win;
  // activation applied AFTER the residual (CTX: SiLU(c4*x + x), common_layers.py:222-224); 0 = none
  int post_act;
  float* partial;      // [nsplit][M][CoutW] fp32 scratch
  // STORE_NCHW_F32 (the logits convolution of a head): the lane that stores a pixel's first 8 channels also decodes them
  // (argmax / threshold / lane priority, kernels_misc.hip decode_mask_kernel's rules) into mask_out[pixel]; nullptr = off
  uint8_t* mask_out;
  int decode_mode;
  // >= 16 bytes of zeros in device memory (the engine's zero page): LDS-DMA source for pixels outside the map (kernels_conv3x3_x3.hip)
  const half_t* zeros;
  // convt_rs_kernel: cap on the persistent pixel-tile groups, resolved when the PLAN is built (developer option VP_CONVT_RS_GROUPS; 0 = no cap).
  // It used to be read at launch / capture time: an option set later changed existing engines behind their plan hash
  int rs_groups;
};

// nn.GELU() (exact erf form, scene_neck.py:8).  ~300 M activations per frame: libm's erff (~45 VALU ops, branchy)
// cost tens of microseconds per layer inside the conv epilogues, so erf is evaluated with Abramowitz-Stegun 7.1.26
// (5-term polynomial in 1/(1+p|u|) times exp(-u^2), |erf error| <= 1.5e-7): ~14 ops, v_exp_f32 / v_rcp_f32.
// Measured against the fp64 erf form over [-12,12]: max |gelu error| 3.3e-7 (fp32 rounding class).
__device__ __forceinline__ float gelu_exact(float x) {
  const float u = fabsf(x) * 0.70710678118654752440f;
  const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, u, 1.0f));
  float poly = fmaf(t, 1.061405429f, -1.453152027f);
  poly = fmaf(t, poly, 1.421413741f);
  poly = fmaf(t, poly, -0.284496736f);
  poly = fmaf(t, poly, 0.254829592f);
  const float P = poly * t * __expf(-u * u);  // = 1 - erf(u)
  const float hx = 0.5f * x;
  return x >= 0.0f ? fmaf(-hx, P, x) : hx * P;
}
// OCP e4m3 (sign, 4 exponent bits with bias 7, 3 mantissa bits; no infinities, 0x7f / 0xff = NaN never produced by the quantiser) -> fp32, exactly,
// in integer arithmetic (no dependence on the fp16 denormal mode): normal codes are re-biased into the fp32 exponent, the seven subnormal
// codes are m * 2^-9.
__host__ __device__ __forceinline__ float e4m3_to_float(unsigned b) {
  const unsigned e = (b >> 3) & 15u, m = b & 7u;
  union { unsigned u; float f; } v;
  v.u = ((e + 120u) << 23) | (m << 20);
  const float mag = e ? v.f : (float)m * 0.001953125f;
  return (b & 0x80u) ? -mag : mag;
}
// 8 e4m3 codes (two dwords, element i in byte i) -> 8 fp16 values packed as a 16-byte piece (what the fp16 weight layout holds there).  Two codes per
// step: sign to bit 15, (exponent, mantissa) to bits 13..7 of each half -- as fp16 bits that is value / 256, an fp16 SUBNORMAL for the seven subnormal
// codes (fp16 denormals are always honoured on this target) -- then one packed multiply by 256, exact.  ~7 instructions per pair (the element-wise
// fp32 route was ~90 per piece and cost AutoDrive 8 % of its rate).
typedef unsigned int vp_u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int vp_u32x2 __attribute__((ext_vector_type(2)));
typedef _Float16 vp_h2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned e4m3x2_to_half2(unsigned two_codes) {
  const unsigned x = (two_codes & 0xffu) | ((two_codes & 0xff00u) << 8);
  union { unsigned u; vp_h2 h; } v;
  v.u = ((x << 8) & 0x80008000u) | ((x << 7) & 0x3f803f80u);
  v.h = v.h * vp_h2{(_Float16)256.0f, (_Float16)256.0f};
  return v.u;
}
__device__ __forceinline__ vp_u32x4 e4m3x8_to_half8(unsigned lo, unsigned hi) {
  return vp_u32x4{e4m3x2_to_half2(lo), e4m3x2_to_half2(lo >> 16), e4m3x2_to_half2(hi), e4m3x2_to_half2(hi >> 16)};
}
__device__ __forceinline__ float sigmoid_f(float x) { return 1.0f / (1.0f + expf(-x)); }
__device__ __forceinline__ float silu_f(float x) { return x / (1.0f + expf(-x)); }

// VP_FP16 variants.  GELU(x) = x * Phi(x) with Phi(x) = sigmoid(g(x)), g = logit(Phi) approximated by an odd
// degree-9 polynomial (minimax fit of the GELU error over [0,10]; coefficients below carry the -log2(e) factor of
// exp2): max |error| 3.4e-6 in fp32 arithmetic, two orders below the fp16 rounding of the stored result, monotone tails
// (x -> +-inf gives x and -0).  7 packable fp32 ops + v_exp_f32 + v_rcp_f32 per value (the erf form: ~16 + 2).
__device__ __forceinline__ float gelu_f16(float x) {
  const float t = x * x;
  float q = fmaf(t, -3.228988134651445e-06f, 8.823812095215544e-05f);
  q = fmaf(q, t, 0.0003602745709940791f);
  q = fmaf(q, t, -0.10522668808698654f);
  q = fmaf(q, t, -2.3020453453063965f);
  return x * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(x * q));
}
__device__ __forceinline__ float sigmoid_f16(float x) { return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(x * -1.4426950408889634f)); }
__device__ __forceinline__ float silu_f16(float x) { return x * sigmoid_f16(x); }
__device__ __forceinline__ float apply_act(float v, int act) {
  if (act == ACT_GELU) return gelu_exact(v);
  if (act == ACT_SILU) return silu_f(v);
  if (act == ACT_SIGMOID) return sigmoid_f(v);
  if (act == ACT_GELU_F16) return gelu_f16(v);
  if (act == ACT_SILU_F16) return silu_f16(v);
  if (act == ACT_SIGMOID_F16) return sigmoid_f16(v);
  if (act == ACT_RELU) return fmaxf(v, 0.0f);
  if (act == ACT_TANH) return tanhf(v);
  if (act == ACT_SILU2) return silu_f(silu_f(v));
  return v;
}

}  // namespace vp
