// HBM-bound kernels of the hot path: preprocess, EfficientNet-B0 stem / depthwise / squeeze-excite,
// context MLP, feature fusion, decode and output resizes.  All activations NHWC, fp16 (hi[,lo]) in HBM,
// fp32 arithmetic in registers, 16-byte accesses along the channel axis.
#include "kernels.hpp"

namespace vp {

__device__ __forceinline__ void load8(const ActView& a, size_t off, float v[8]) {
  const h8_t h = *reinterpret_cast<const h8_t*>(a.hi + off);
#pragma unroll
  for (int i = 0; i < 8; ++i) v[i] = (float)h[i];
  if (a.lo) {
    const h8_t l = *reinterpret_cast<const h8_t*>(a.lo + off);
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] += (float)l[i];
  }
}
__device__ __forceinline__ void store8(const ActView& a, size_t off, const float v[8]) {
  h8_t h;
#pragma unroll
  for (int i = 0; i < 8; ++i) h[i] = (half_t)v[i];
  *reinterpret_cast<h8_t*>(a.hi + off) = h;
  if (a.lo) {
    h8_t l;
#pragma unroll
    for (int i = 0; i < 8; ++i) l[i] = (half_t)(v[i] - (float)h[i]);
    *reinterpret_cast<h8_t*>(a.lo + off) = l;
  }
}

// ------------------------------------------------------------------------------------------ preprocess
// Integer bilinear (definition: oracle/pre_post.py resize_bilinear_u8; modelled on cv::resize INTER_LINEAR,
// onnx_runtime_backend.cpp:44) + /255 + (x-mean)/std + HWC->CHW (onnx_runtime_backend.cpp:45-57,
// onnxruntime_engine.cpp:72-102).  Tap tables are built on the host so the device does integer math only.
__global__ __launch_bounds__(256) void preprocess_kernel(const PreprocessParams p) {
  const int x = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y;
  if (x >= p.out_w) return;
  const int4 xt = *reinterpret_cast<const int4*>(p.xtab + 4 * x);
  const int4 yt = *reinterpret_cast<const int4*>(p.ytab + 4 * y);
  const uint8_t* r0 = p.frame + (size_t)yt.x * p.stride;
  const uint8_t* r1 = p.frame + (size_t)yt.y * p.stride;
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    const int sc = p.src_c[c];
    const int s0 = (int)r0[xt.x * 3 + sc] * xt.z + (int)r0[xt.y * 3 + sc] * xt.w;
    const int s1 = (int)r1[xt.x * 3 + sc] * xt.z + (int)r1[xt.y * 3 + sc] * xt.w;
    int q = (((yt.z * (s0 >> 4)) >> 16) + ((yt.w * (s1 >> 4)) >> 16) + 2) >> 2;
    q = min(max(q, 0), 255);
    const float t = __fdiv_rn((float)q, 255.0f);
    p.out[((size_t)c * p.out_h + y) * p.out_w + x] = __fdiv_rn(__fsub_rn(t, p.mean[c]), p.stdv[c]);
  }
}

// -------------------------------------------------------------------------------------------------- stem
// features[0]: Conv 3x3 / s2 / p1, 3->32 (+BN folded) + SiLU, fp32 NCHW in -> NHWC act out (backbone.py:13).
__global__ __launch_bounds__(256) void stem_kernel(const StemParams p) {
  __shared__ float ws[27 * 32 + 32];
  for (int i = threadIdx.x; i < 27 * 32 + 32; i += 256) ws[i] = i < 27 * 32 ? p.w[i] : p.b[i - 27 * 32];
  __syncthreads();
  const int OH = p.H / 2, OW = p.W / 2;
  const int t = blockIdx.x * 256 + threadIdx.x;
  const int g = t & 3, pix = t >> 2;  // 4 threads per pixel, 8 output channels each
  if (pix >= OH * OW) return;
  const int oy = pix / OW, ox = pix - oy * OW;
  float acc[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) acc[i] = ws[27 * 32 + g * 8 + i];
#pragma unroll
  for (int ci = 0; ci < 3; ++ci)
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) {
      const int iy = 2 * oy + ky - 1;
#pragma unroll
      for (int kx = 0; kx < 3; ++kx) {
        const int ix = 2 * ox + kx - 1;
        const float v = ((unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W) ? p.in[((size_t)ci * p.H + iy) * p.W + ix] : 0.0f;
        const float* wk = ws + ((ci * 3 + ky) * 3 + kx) * 32 + g * 8;
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[i] = fmaf(v, wk[i], acc[i]);
      }
    }
#pragma unroll
  for (int i = 0; i < 8; ++i) acc[i] = silu_f(acc[i]);
  store8(p.out, (size_t)pix * 32 + g * 8, acc);
}

// -------------------------------------------------------------------------------------- depthwise conv
// MBConv depthwise k x k (k = 3|5), stride 1|2, pad (k-1)/2, BN folded, SiLU.  Thread = 8 channels of one pixel.
__global__ __launch_bounds__(256) void dwconv_kernel(const DwParams p) {
  const int CG = p.in.C >> 3;
  const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
  const int OH = p.out.H, OW = p.out.W;
  if (t >= (long long)OH * OW * CG) return;
  const int cg = (int)(t % CG);
  const int pix = (int)(t / CG);
  const int oy = pix / OW, ox = pix - oy * OW;
  const int pad = (p.k - 1) >> 1;
  float acc[8];
  {
    const f32x4_t b0 = *reinterpret_cast<const f32x4_t*>(p.b + cg * 8), b1 = *reinterpret_cast<const f32x4_t*>(p.b + cg * 8 + 4);
#pragma unroll
    for (int i = 0; i < 4; ++i) { acc[i] = b0[i]; acc[4 + i] = b1[i]; }
  }
  for (int ky = 0; ky < p.k; ++ky) {
    const int iy = oy * p.stride + ky - pad;
    if ((unsigned)iy >= (unsigned)p.in.H) continue;
    for (int kx = 0; kx < p.k; ++kx) {
      const int ix = ox * p.stride + kx - pad;
      if ((unsigned)ix >= (unsigned)p.in.W) continue;
      float v[8];
      load8(p.in, ((size_t)iy * p.in.W + ix) * p.in.C + cg * 8, v);
      const float* wk = p.w + (size_t)(ky * p.k + kx) * p.in.C + cg * 8;
      const f32x4_t w0 = *reinterpret_cast<const f32x4_t*>(wk), w1 = *reinterpret_cast<const f32x4_t*>(wk + 4);
#pragma unroll
      for (int i = 0; i < 4; ++i) { acc[i] = fmaf(v[i], w0[i], acc[i]); acc[4 + i] = fmaf(v[4 + i], w1[i], acc[4 + i]); }
    }
  }
#pragma unroll
  for (int i = 0; i < 8; ++i) acc[i] = silu_f(acc[i]);
  store8(p.out, (size_t)pix * p.out.C + cg * 8, acc);
}

// ------------------------------------------------------------------------------- channel sums (avg-pool)
// Deterministic two-level reduction: partial[slab][C] (fp32 sums over a slab of pixels).
__global__ __launch_bounds__(256) void pool_partial_kernel(const PoolParams p) {
  __shared__ float red[256 * 8];
  const int CG = p.in.C >> 3;
  const int HW = p.in.H * p.in.W;
  const int lanes_px = 256 / CG > 0 ? 256 / CG : 1;  // pixel lanes per block (C <= 2048)
  const int cg = threadIdx.x % CG, pl = threadIdx.x / CG;
  const int per = (HW + p.nslab - 1) / p.nslab;
  const int begin = blockIdx.x * per, end = min(begin + per, HW);
  float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  if (pl < lanes_px) {
    for (int px = begin + pl; px < end; px += lanes_px) {
      float v[8];
      load8(p.in, (size_t)px * p.in.C + cg * 8, v);
#pragma unroll
      for (int i = 0; i < 8; ++i) acc[i] += v[i];
    }
  }
#pragma unroll
  for (int i = 0; i < 8; ++i) red[threadIdx.x * 8 + i] = acc[i];
  __syncthreads();
  if (pl == 0 && threadIdx.x < CG) {
    for (int q = 1; q < lanes_px; ++q)
#pragma unroll
      for (int i = 0; i < 8; ++i) acc[i] += red[(q * CG + cg) * 8 + i];
#pragma unroll
    for (int i = 0; i < 8; ++i) p.partial[(size_t)blockIdx.x * p.in.C + cg * 8 + i] = acc[i];
  }
}

// ------------------------------------------------------------------------------------- squeeze-excite FCs
// mean -> fc1 -> SiLU -> fc2 -> sigmoid -> scale[C]   (torchvision SqueezeExcitation; 1 workgroup).
// Latency-bound (one workgroup, ~100 KB of weights): 16 waves, 16-byte loads, independent loads unrolled so
// they overlap instead of forming a dependent chain.
__global__ __launch_bounds__(1024) void se_fc_kernel(const SeParams p) {
  extern __shared__ float sm[];  // mean[C] + s1[sq]
  float* mean = sm;
  float* s1 = sm + p.C;
  for (int c = threadIdx.x; c < p.C; c += 1024) {
    float s = 0.f;
#pragma unroll 8
    for (int q = 0; q < p.nslab; ++q) s += p.partial[(size_t)q * p.C + c];
    mean[c] = s * p.inv_hw;
  }
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int C4 = p.C >> 2;
  const f32x4_t* m4 = reinterpret_cast<const f32x4_t*>(mean);
  for (int j = wave; j < p.sq; j += 16) {
    const f32x4_t* wr = reinterpret_cast<const f32x4_t*>(p.w1 + (size_t)j * p.C);
    float s = 0.f;
#pragma unroll 5
    for (int c = lane; c < C4; c += 64) {
      const f32x4_t a = wr[c], m = m4[c];
      s += a[0] * m[0] + a[1] * m[1] + a[2] * m[2] + a[3] * m[3];
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    if (lane == 0) s1[j] = silu_f(s + p.b1[j]);
  }
  __syncthreads();
  for (int c = threadIdx.x; c < p.C; c += 1024) {
    float s = p.b2[c];
    const float* wr = p.w2 + (size_t)c * p.sq;
#pragma unroll 8
    for (int j = 0; j < p.sq; ++j) s = fmaf(wr[j], s1[j], s);
    p.scale[c] = c < p.Creal ? sigmoid_f(s) : 0.0f;
  }
}

// Per-frame project weights: W'[n][k] = W[n][k] * scale[k]  (batch is 1, so the SE channel scale commutes
// into the K axis of the following 1x1 projection; the activation tensor is never re-written).
__global__ __launch_bounds__(256) void scale_weights_kernel(const ScaleWParams p) {
  const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
  const int CG = p.C >> 3;
  if (t >= (long long)p.rows * CG) return;
  const int cg = (int)(t % CG);
  const size_t off = (size_t)(t / CG) * p.C + cg * 8;
  h8_t h, l;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const float v = p.w[off + i] * p.scale[cg * 8 + i];
    h[i] = (half_t)v;
    l[i] = (half_t)(v - (float)h[i]);
  }
  *reinterpret_cast<h8_t*>(p.out_hi + off) = h;
  if (p.out_lo) *reinterpret_cast<h8_t*>(p.out_lo + off) = l;
}

// ------------------------------------------------------------------------------------------ context MLP
// out[n] = act(b[n] + sum_k W[n][k] x[k]), one wave per output (scene_context.py:28-38).
__global__ __launch_bounds__(256) void fc_kernel(const FcParams p) {
  const int lane = threadIdx.x & 63;
  const int n = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (n >= p.N) return;
  float s = 0.f;
  for (int k = lane; k < p.K; k += 64) {
    float xv;
    if (p.partial) {
      xv = 0.f;
      for (int q = 0; q < p.nslab; ++q) xv += p.partial[(size_t)q * p.Kstride + k];
      xv *= p.inv_hw;
    } else {
      xv = p.x[k];
    }
    s = fmaf(p.w[(size_t)n * p.K + k], xv, s);
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
  if (lane == 0) p.out[n] = apply_act(s + p.b[n], p.act);
}

// context_layer_3: Conv 3x3 1->128 on the 10x20 sigmoid map + GELU (scene_context.py:19,46-47).
__global__ __launch_bounds__(256) void ctx_conv1_kernel(const CtxConv1Params p) {
  const int CG = p.out.C >> 3;
  const int t = blockIdx.x * 256 + threadIdx.x;
  if (t >= p.H * p.W * CG) return;
  const int cg = t % CG, pix = t / CG;
  const int y = pix / p.W, x = pix - y * p.W;
  float acc[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) acc[i] = p.b[cg * 8 + i];
#pragma unroll
  for (int ky = 0; ky < 3; ++ky)
#pragma unroll
    for (int kx = 0; kx < 3; ++kx) {
      const int iy = y + ky - 1, ix = x + kx - 1;
      const float v = ((unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W) ? p.map[iy * p.W + ix] : 0.f;
#pragma unroll
      for (int i = 0; i < 8; ++i) acc[i] = fmaf(v, p.w[(ky * 3 + kx) * p.out.C + cg * 8 + i], acc[i]);
    }
#pragma unroll
  for (int i = 0; i < 8; ++i) acc[i] = gelu_exact(acc[i]);
  store8(p.out, (size_t)pix * p.out.C + cg * 8, acc);
}

// --------------------------------------------------------------------------------- EgoLanes feature fusion
// MaxPool2x2 applied n times == max over a 2^n x 2^n window; concat along C (backbone_feature_fusion.py:13-38).
__global__ __launch_bounds__(256) void fusion_kernel(const FusionParams p) {
  const int t = blockIdx.x * 256 + threadIdx.x;
  const int HW = p.out.H * p.out.W;
  if (t >= HW * p.out.C) return;
  const int c = t % p.out.C, pix = t / p.out.C;
  const int y = pix / p.out.W, x = pix - y * p.out.W;
  float best = 0.0f;
  if (c < p.Creal_out) {
    int lvl = 0, cc = c;
    while (cc >= p.creal[lvl]) { cc -= p.creal[lvl]; ++lvl; }
    const ActView& a = p.f[lvl];
    const int n = 1 << p.shift[lvl];
    best = -3.0e38f;
    for (int dy = 0; dy < n; ++dy)
      for (int dx = 0; dx < n; ++dx) {
        const size_t off = ((size_t)(y * n + dy) * a.W + (x * n + dx)) * a.C + cc;
        float v = (float)a.hi[off];
        if (a.lo) v += (float)a.lo[off];
        best = fmaxf(best, v);
      }
  }
  const half_t h = (half_t)best;
  p.out.hi[t] = h;
  if (p.out.lo) p.out.lo[t] = (half_t)(best - (float)h);
}

// ------------------------------------------------------------------------------------------------ decode
// mode 0: argmax over C planes, first max wins, 255 where class==1 (run_model_node.cpp:144-163,
//         cuda_visualization_kernels.cu:13-42); C==1 -> 255 where >0 (:164-171)
// mode 1: lane priority label {2,1,0,255} (cuda_visualization_kernels.cu:45-75)
// mode 2: raw class index (scene_seg_infer.py:52-55)
__global__ __launch_bounds__(256) void decode_mask_kernel(const float* logits, int C, int HW, int mode, uint8_t* out) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= HW) return;
  uint8_t r;
  if (mode == 1) {
    const bool b0 = logits[i] > 0.0f, b1 = logits[HW + i] > 0.0f, b2 = logits[2 * HW + i] > 0.0f;
    r = b2 ? 2 : (b1 ? 1 : (b0 ? 0 : 255));
  } else if (C > 1) {
    float best = -1e9f;
    int cls = 0;
    for (int c = 0; c < C; ++c) {
      const float s = logits[(size_t)c * HW + i];
      if (s > best) { best = s; cls = c; }
    }
    r = mode == 2 ? (uint8_t)cls : (cls == 1 ? 255 : 0);
  } else {
    r = logits[i] > 0.0f ? 255 : 0;
  }
  out[i] = r;
}

// cv::resize INTER_NEAREST (run_model_node.cpp:176-177); index tables from the host.
__global__ __launch_bounds__(256) void resize_nearest_kernel(const uint8_t* src, int sw, const int* ytab, const int* xtab,
                                                             int oh, int ow, uint8_t* dst) {
  const int x = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y;
  if (x >= ow) return;
  dst[(size_t)y * ow + x] = src[(size_t)ytab[y] * sw + xtab[x]];
}

// float bilinear up-resize of the depth plane (run_model_node.cpp:100-104); taps from the host; no FMA contraction.
__global__ __launch_bounds__(256) void resize_bilinear_f32_kernel(const float* src, int sw, const int* yi, const float* yf,
                                                                  const int* xi, const float* xf, int oh, int ow, float* dst) {
#pragma clang fp contract(off)  // ROCm's __fmul_rn/__fadd_rn are plain * and +: keep them from fusing into v_fma
  const int x = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y;
  if (x >= ow) return;
  const int x0 = xi[2 * x], x1 = xi[2 * x + 1], y0 = yi[2 * y], y1 = yi[2 * y + 1];
  const float a0 = xf[2 * x], a1 = xf[2 * x + 1], b0 = yf[2 * y], b1 = yf[2 * y + 1];
  const float h0 = __fadd_rn(__fmul_rn(src[(size_t)y0 * sw + x0], a0), __fmul_rn(src[(size_t)y0 * sw + x1], a1));
  const float h1 = __fadd_rn(__fmul_rn(src[(size_t)y1 * sw + x0], a0), __fmul_rn(src[(size_t)y1 * sw + x1], a1));
  dst[(size_t)y * ow + x] = __fadd_rn(__fmul_rn(h0, b0), __fmul_rn(h1, b1));
}

// ----------------------------------------------------------------------------------- layout conversions
// fp32 NCHW (host-visible test/debug format) <-> NHWC activation.
__global__ __launch_bounds__(256) void nchw_to_act_kernel(const float* src, int Creal, ActView a) {
  const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
  const int HW = a.H * a.W;
  if (t >= (long long)HW * a.C) return;
  const int c = (int)(t % a.C), pix = (int)(t / a.C);
  const float v = c < Creal ? src[(size_t)c * HW + pix] : 0.0f;
  const half_t h = (half_t)v;
  a.hi[t] = h;
  if (a.lo) a.lo[t] = (half_t)(v - (float)h);
}
__global__ __launch_bounds__(256) void act_to_nchw_kernel(ActView a, int Creal, float* dst) {
  const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
  const int HW = a.H * a.W;
  if (t >= (long long)HW * Creal) return;
  const int pix = (int)(t % HW), c = (int)(t / HW);
  const size_t off = (size_t)pix * a.C + c;
  float v = (float)a.hi[off];
  if (a.lo) v += (float)a.lo[off];
  dst[t] = v;
}

// ---------------------------------------------------------------------------------------------- launchers
#define VP_LAUNCH(kern, grid, block, shmem, st, ...)          \
  do {                                                        \
    hipLaunchKernelGGL(kern, grid, block, shmem, st, __VA_ARGS__); \
    return hipGetLastError();                                 \
  } while (0)

static inline unsigned nblk(long long n) { return (unsigned)((n + 255) / 256); }

hipError_t launch_preprocess(const PreprocessParams& p, hipStream_t st) {
  VP_LAUNCH(preprocess_kernel, dim3(nblk(p.out_w), p.out_h), dim3(256), 0, st, p);
}
hipError_t launch_stem(const StemParams& p, hipStream_t st) {
  VP_LAUNCH(stem_kernel, dim3(nblk((long long)(p.H / 2) * (p.W / 2) * 4)), dim3(256), 0, st, p);
}
hipError_t launch_dwconv(const DwParams& p, hipStream_t st) {
  VP_LAUNCH(dwconv_kernel, dim3(nblk((long long)p.out.H * p.out.W * (p.in.C >> 3))), dim3(256), 0, st, p);
}
hipError_t launch_pool_partial(const PoolParams& p, hipStream_t st) {
  VP_LAUNCH(pool_partial_kernel, dim3(p.nslab), dim3(256), 0, st, p);
}
hipError_t launch_se_fc(const SeParams& p, hipStream_t st) {
  VP_LAUNCH(se_fc_kernel, dim3(1), dim3(1024), (p.C + p.sq) * sizeof(float), st, p);
}
hipError_t launch_scale_weights(const ScaleWParams& p, hipStream_t st) {
  VP_LAUNCH(scale_weights_kernel, dim3(nblk((long long)p.rows * (p.C >> 3))), dim3(256), 0, st, p);
}
hipError_t launch_fc(const FcParams& p, hipStream_t st) {
  VP_LAUNCH(fc_kernel, dim3((p.N + 3) / 4), dim3(256), 0, st, p);
}
hipError_t launch_ctx_conv1(const CtxConv1Params& p, hipStream_t st) {
  VP_LAUNCH(ctx_conv1_kernel, dim3(nblk((long long)p.H * p.W * (p.out.C >> 3))), dim3(256), 0, st, p);
}
hipError_t launch_fusion(const FusionParams& p, hipStream_t st) {
  VP_LAUNCH(fusion_kernel, dim3(nblk((long long)p.out.H * p.out.W * p.out.C)), dim3(256), 0, st, p);
}
hipError_t launch_decode_mask(const float* logits, int C, int HW, int mode, uint8_t* out, hipStream_t st) {
  VP_LAUNCH(decode_mask_kernel, dim3(nblk(HW)), dim3(256), 0, st, logits, C, HW, mode, out);
}
hipError_t launch_resize_nearest(const uint8_t* src, int sw, const int* ytab, const int* xtab, int oh, int ow, uint8_t* dst,
                                 hipStream_t st) {
  VP_LAUNCH(resize_nearest_kernel, dim3(nblk(ow), oh), dim3(256), 0, st, src, sw, ytab, xtab, oh, ow, dst);
}
hipError_t launch_resize_bilinear_f32(const float* src, int sw, const int* yi, const float* yf, const int* xi, const float* xf,
                                      int oh, int ow, float* dst, hipStream_t st) {
  VP_LAUNCH(resize_bilinear_f32_kernel, dim3(nblk(ow), oh), dim3(256), 0, st, src, sw, yi, yf, xi, xf, oh, ow, dst);
}
hipError_t launch_nchw_to_act(const float* src, int Creal, const ActView& a, hipStream_t st) {
  VP_LAUNCH(nchw_to_act_kernel, dim3(nblk((long long)a.H * a.W * a.C)), dim3(256), 0, st, src, Creal, a);
}
hipError_t launch_act_to_nchw(const ActView& a, int Creal, float* dst, hipStream_t st) {
  VP_LAUNCH(act_to_nchw_kernel, dim3(nblk((long long)a.H * a.W * Creal)), dim3(256), 0, st, a, Creal, dst);
}

}  // namespace vp
