// Pre/post-processing and glue kernels of the hot path: preprocess, context conv 1->128, EgoLanes feature fusion,
// decode, output resizes, layout conversions.  (Encoder pieces: kernels_backbone.hip; convs: kernels_conv*.hip.)
#include "act_io.hpp"

namespace vp {

// ------------------------------------------------------------------------------------------ preprocess
// Integer bilinear (definition: oracle/pre_post.py resize_bilinear_u8; modelled on cv::resize INTER_LINEAR,
// onnx_runtime_backend.cpp:44) + /255 + (x-mean)/std + HWC->CHW (onnx_runtime_backend.cpp:45-57,
// onnxruntime_engine.cpp:72-102).  Tap tables are built on the host so the device does integer math only.
// The u8 -> [0, 1] step exists in two spellings that differ in the last bit for 322 of the 768 (byte, channel) pairs:
//   norm_form 0  q / 255            torchvision to_tensor (Models/inference/scene_seg_infer.py:15-20): one IEEE division
//   norm_form 1  q * fl(1 / 255)    cv::Mat::convertTo(CV_32FC3, 1.0 / 255.0) of the C++ front-ends (onnx_runtime_backend.cpp:45,
//                                   tensorrt_backend.cpp:164, onnxruntime_engine.cpp:85): the scale is rounded to float once, then multiplied
// followed in both by the float subtraction and the IEEE float division the sources spell out (cv::subtract / cv::divide by a Scalar
// converted to float, :48-49; the scalar loop (x - MEAN[c]) / STD[c], onnxruntime_engine.cpp:98).
__device__ __forceinline__ float unit_from_u8(int q, int norm_form) {
  return norm_form ? __fmul_rn((float)q, (float)(1.0 / 255.0)) : __fdiv_rn((float)q, 255.0f);
}
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
    const float t = unit_from_u8(q, p.norm_form);
    p.out[((size_t)c * p.out_h + y) * p.out_w + x] = __fdiv_rn(__fsub_rn(t, p.mean[c]), p.stdv[c]);
  }
}

// Pillow's 8-bit resample passes (kernels.hpp PilResampleParams): int32 accumulation from 1 << 21 of u8 x 22-bit coefficients,
// `>> 22`, clip to u8 (Resample.c clip8) -- integer work, bit-exact by construction.
__device__ __forceinline__ int pil_clip8(int acc) { return min(max(acc >> 22, 0), 255); }
__global__ __launch_bounds__(256) void pil_resample_h_kernel(const PilResampleParams p) {
  const int x = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y;
  if (x >= p.out_w) return;
  const int lo = p.hb[2 * x], n = p.hb[2 * x + 1];
  const int* k = p.hk + (size_t)x * p.hks;
  const uint8_t* row = p.frame + (size_t)y * p.stride + (size_t)lo * 3;
  int a0 = 1 << 21, a1 = 1 << 21, a2 = 1 << 21;
  for (int t = 0; t < n; ++t) {
    const int kt = k[t];
    a0 += (int)row[3 * t + 0] * kt;
    a1 += (int)row[3 * t + 1] * kt;
    a2 += (int)row[3 * t + 2] * kt;
  }
  uint8_t* d = p.tmp + ((size_t)y * p.out_w + x) * 3;
  d[0] = (uint8_t)pil_clip8(a0);
  d[1] = (uint8_t)pil_clip8(a1);
  d[2] = (uint8_t)pil_clip8(a2);
}
__global__ __launch_bounds__(256) void pil_resample_v_kernel(const PilResampleParams p) {
  const int x = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y;
  if (x >= p.out_w) return;
  const int lo = p.vb[2 * y], n = p.vb[2 * y + 1];
  const int* k = p.vk + (size_t)y * p.vks;
  const uint8_t* col = p.tmp + ((size_t)lo * p.out_w + x) * 3;
  int a[3] = {1 << 21, 1 << 21, 1 << 21};
  for (int t = 0; t < n; ++t) {
    const int kt = k[t];
    const uint8_t* s = col + (size_t)t * p.out_w * 3;
    a[0] += (int)s[0] * kt;
    a[1] += (int)s[1] * kt;
    a[2] += (int)s[2] * kt;
  }
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    const int sc = p.src_c[c];
    const int q = pil_clip8(sc == 0 ? a[0] : (sc == 1 ? a[1] : a[2]));
    const float t = unit_from_u8(q, p.norm_form);   // default 0: torchvision to_tensor, u8 -> fp32 / 255
    p.out[((size_t)c * p.out_h + y) * p.out_w + x] = __fdiv_rn(__fsub_rn(t, p.mean[c]), p.stdv[c]);
  }
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
  for (int i = 0; i < 8; ++i) acc[i] = p.act == ACT_GELU ? gelu_exact(acc[i]) : apply_act(acc[i], p.act);
  store8(p.out, (size_t)pix * p.out.C + cg * 8, acc);
}

// Round 5: the matvec that builds the one-channel H x W map (scene networks: context_layer_2 + sigmoid; AutoDrive CTX: exp0 + SiLU twice) FUSED with
// the 3x3 convolution 1 -> C that reads it (context_layer_3 / ctx0): the map was a launch boundary and 0.8 - 131 KB through HBM for a tensor a
// workgroup can rebuild where it needs it.  A workgroup owns a T x T pixel patch: (1) x -- the previous layer's vector, or the average pool rebuilt
// from slab partials in their fixed order -- goes to LDS; (2) the (T + 2)^2 map values under the patch: row n of the matrix against x, G lanes per
// row (each a strided quarter-vector walk, then an xor-shuffle tree: a fixed order), bias, activation, zero outside the map (the convolution pads the
// MAP); (3) the nine taps per (pixel, channel octet), ctx_conv1_kernel's arithmetic term for term.  Neighbouring workgroups recompute the halo rows
// (1.27x at T = 16, 1.56x at T = 8: a few thousand multiply-adds).
template <int T>
__global__ __launch_bounds__(256) void ctx_exp_conv1_kernel(const CtxExpConv1Params q) {
  extern __shared__ __attribute__((aligned(16))) float smem_f[];
  const FcParams& p = q.fc;
  const CtxConv1Params& c = q.cv;
  constexpr int HT = T + 2;
  float* const xs = smem_f;                       // [K]
  float* const mp = smem_f + ((p.K + 3) & ~3);    // [HT][HT]
  for (int k = threadIdx.x; k < p.K; k += 256) {
    float xv;
    if (p.partial) {
      xv = 0.f;
#pragma unroll 8
      for (int s = 0; s < p.nslab; ++s) xv += p.partial[(size_t)s * p.Kstride + k];
      xv *= p.inv_hw;
    } else {
      xv = p.x[k];
    }
    xs[k] = xv;
  }
  __syncthreads();
  const int tiles_x = (c.W + T - 1) / T;
  const int y0 = (blockIdx.x / tiles_x) * T, x0 = (blockIdx.x % tiles_x) * T;
  const int G = q.glanes, rows_per_pass = 256 / G;
  const int g = threadIdx.x / G, l = threadIdx.x % G;
  const f32x4_t* x4 = reinterpret_cast<const f32x4_t*>(xs);
  const int K4 = p.K >> 2;
  for (int r0 = 0; r0 < HT * HT; r0 += rows_per_pass) {   // uniform trip count: every lane takes part in the shuffle tree of every pass
    const int r = r0 + g;
    const bool live = r < HT * HT;
    const int hy = r / HT, hx = r - hy * HT;
    const int iy = y0 - 1 + hy, ix = x0 - 1 + hx;
    const bool inside = live && (unsigned)iy < (unsigned)c.H && (unsigned)ix < (unsigned)c.W;
    const int n = inside ? iy * c.W + ix : 0;
    float s = 0.f;
    if (p.w8) {
      const unsigned* wr8 = reinterpret_cast<const unsigned*>(p.w8 + (size_t)n * p.K);
#pragma unroll 16
      for (int k = l; k < K4; k += G) {   // (unrolled: a lane's pieces of the row are independent loads -- one or two round trips, not one per piece)
        const unsigned c4 = wr8[k];
        const f32x4_t m = x4[k];
        s += (e4m3_to_float(c4 & 0xffu) * m[0] + e4m3_to_float((c4 >> 8) & 0xffu) * m[1]) + (e4m3_to_float((c4 >> 16) & 0xffu) * m[2] + e4m3_to_float(c4 >> 24) * m[3]);
      }
    } else {
      const f32x4_t* wr = reinterpret_cast<const f32x4_t*>(p.w + (size_t)n * p.K);
#pragma unroll 16
      for (int k = l; k < K4; k += G) {
        const f32x4_t a = wr[k], m = x4[k];
        s += (a[0] * m[0] + a[1] * m[1]) + (a[2] * m[2] + a[3] * m[3]);
      }
    }
    for (int o = G >> 1; o > 0; o >>= 1) s += __shfl_xor(s, o);
    if (l == 0 && live) mp[r] = inside ? apply_act((p.w8 ? s * p.wscale8[n] : s) + p.b[n], p.act) : 0.f;
  }
  __syncthreads();
  const int CG = c.out.C >> 3;
  for (int t = threadIdx.x; t < T * T * CG; t += 256) {
    const int cg = t % CG, lp = t / CG;
    const int ly = lp / T, lx = lp - ly * T;
    const int y = y0 + ly, x = x0 + lx;
    if (y >= c.H || x >= c.W) continue;
    float acc[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[i] = c.b[cg * 8 + i];
#pragma unroll
    for (int ky = 0; ky < 3; ++ky)
#pragma unroll
      for (int kx = 0; kx < 3; ++kx) {
        const float v = mp[(ly + ky) * HT + lx + kx];
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[i] = fmaf(v, c.w[(ky * 3 + kx) * c.out.C + cg * 8 + i], acc[i]);
      }
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[i] = c.act == ACT_GELU ? gelu_exact(acc[i]) : apply_act(acc[i], c.act);
    store8(c.out, (size_t)(y * c.W + x) * c.out.C + cg * 8, acc);
  }
}

// --------------------------------------------------------------------------------- EgoLanes feature fusion
// MaxPool2x2 applied n times == max over a 2^n x 2^n window; concat along C (backbone_feature_fusion.py:13-38).
// One thread per output element, walking its window two bytes at a time: the fallback for channel counts that are not octet multiples
// (and, until round 4, the only form: 115 us of an EgoLanes frame -- the 6400 threads of the stride-2 tap each chained 512 loads).
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
// Round 4: one workgroup per output pixel.  A work unit = (tap, channel octet, slice of the tap's window): up to 16 window elements fetched as
// independent 16-byte pieces, their maximum parked in LDS; a second pass takes the maximum over a (tap, octet)'s slices and stores 8 channels.
// Same values as the kernel above (a maximum has no rounding), 16 us instead of 115 (measured).
__global__ __launch_bounds__(256) void fusion_octet_kernel(const FusionParams p) {
  __shared__ float part[512 * 8];
  const int pix = blockIdx.x, tid = threadIdx.x;
  const int y = pix / p.out.W, x = pix - y * p.out.W;
  // unit ranges per tap: octets * slices, slices = min(window elements, 16)
  int ubase[6], nsl[5];
  ubase[0] = 0;
#pragma unroll
  for (int l = 0; l < 5; ++l) {
    const int n2 = 1 << (2 * p.shift[l]);
    nsl[l] = n2 < 16 ? n2 : 16;
    ubase[l + 1] = ubase[l] + (p.creal[l] >> 3) * nsl[l];
  }
  for (int u = tid; u < ubase[5]; u += 256) {
    int l = 0;
    while (u >= ubase[l + 1]) ++l;
    const int r = u - ubase[l], oct = r / nsl[l], sl = r - oct * nsl[l];
    const ActView& a = p.f[l];
    const int n = 1 << p.shift[l], n2 = n * n, per = n2 / nsl[l];
    float best[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) best[i] = -3.0e38f;
#pragma unroll 4
    for (int e = sl * per; e < (sl + 1) * per; ++e) {
      const int dy = e >> p.shift[l], dx = e & (n - 1);
      float v[8];
      load8(a, ((size_t)(y * n + dy) * a.W + (x * n + dx)) * a.C + oct * 8, v);
#pragma unroll
      for (int i = 0; i < 8; ++i) best[i] = fmaxf(best[i], v[i]);
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) part[u * 8 + i] = best[i];
  }
  __syncthreads();
  const int n_oct = p.out.C >> 3;
  for (int o = tid; o < n_oct; o += 256) {
    float best[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) best[i] = 0.0f;   // pad channels stay zero
    if (o * 8 < p.Creal_out) {
      int l = 0, oc = o;
      while (oc >= (p.creal[l] >> 3)) { oc -= p.creal[l] >> 3; ++l; }
      const float* src = part + (ubase[l] + oc * nsl[l]) * 8;
#pragma unroll
      for (int i = 0; i < 8; ++i) best[i] = src[i];
      for (int s = 1; s < nsl[l]; ++s)
#pragma unroll
        for (int i = 0; i < 8; ++i) best[i] = fmaxf(best[i], src[s * 8 + i]);
    }
    store8(p.out, (size_t)pix * p.out.C + o * 8, best);
  }
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

// MasksVisualizationEngine::visualize (common/visualizers/masks_visualization_engine.cpp:11-58): colour LUT on the mask
// (createColorMask :41-58; label values not listed stay black), cv::resize INTER_NEAREST to the frame size (:21-25),
// cv::addWeighted(color, 0.5, frame, 0.5, 0) (:29) = saturate_cast<uchar>(cvRound(0.5*a + 0.5*b)): both halves are exact
// in float, cvRound rounds half to even -> (a+b)>>1, plus 1 when the sum is odd and that half is odd.
__global__ __launch_bounds__(256) void viz_blend_kernel(const uint8_t* mask, int mw, const int* ytab, const int* xtab, const uint8_t* frame,
                                                        int stride, int oh, int ow, const uint8_t* lut /* [256][3] BGR */, int frame_is_rgb,
                                                        uint8_t* dst) {
  const int x = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y;
  if (x >= ow) return;
  const uint8_t label = mask[(size_t)ytab[y] * mw + xtab[x]];
  const uint8_t* f = frame + (size_t)y * stride + 3 * x;
  uint8_t* d = dst + ((size_t)y * ow + x) * 3;
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    const int a = lut[label * 3 + c], b = f[frame_is_rgb ? 2 - c : c];
    const int s2 = a + b, h = s2 >> 1;
    d[c] = (uint8_t)((s2 & 1) ? h + (h & 1) : h);
  }
}

// cv::resize INTER_NEAREST (run_model_node.cpp:176-177); index tables from the host.
__global__ __launch_bounds__(256) void resize_nearest_kernel(const uint8_t* src, int sw, const int* ytab, const int* xtab,
                                                             int oh, int ow, uint8_t* dst) {
  const int x = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y;
  if (x >= ow) return;
  dst[(size_t)y * ow + x] = src[(size_t)ytab[y] * sw + xtab[x]];
}

// float bilinear up-resize of the depth plane (run_model_node.cpp:100-104); taps from the host.
__global__ __launch_bounds__(256) void resize_bilinear_f32_kernel(const float* src, int sw, const int* yi, const float* yf,
                                                                  const int* xi, const float* xf, int oh, int ow, float* dst) {
  // No FMA contraction here (the definition rounds every product and sum).  ROCm's __fmul_rn/__fadd_rn are plain
  // '*' and '+' compiled with contraction allowed inside their own bodies, so the arithmetic is spelled out with
  // operators under a lexically scoped contract(off).
#pragma clang fp contract(off)
  const int x = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y;
  if (x >= ow) return;
  const int x0 = xi[2 * x], x1 = xi[2 * x + 1], y0 = yi[2 * y], y1 = yi[2 * y + 1];
  const float a0 = xf[2 * x], a1 = xf[2 * x + 1], b0 = yf[2 * y], b1 = yf[2 * y + 1];
  const float p00 = src[(size_t)y0 * sw + x0] * a0, p01 = src[(size_t)y0 * sw + x1] * a1;
  const float p10 = src[(size_t)y1 * sw + x0] * a0, p11 = src[(size_t)y1 * sw + x1] * a1;
  const float h0 = p00 + p01, h1 = p10 + p11;
  const float q0 = h0 * b0, q1 = h1 * b1;
  dst[(size_t)y * ow + x] = q0 + q1;
}

// ------------------------------------------------------------------------------ depth visualisation
// DepthVisualizationEngine::visualize (depth_visualization_engine.cpp:9-26): minMaxLoc -> convertTo(CV_8U, 255/(max-min),
// -min*255/(max-min)) -> applyColorMap(VIRIDIS).  Two launches over the frame-size fp32 depth plane.
// Order-preserving float -> unsigned key, so that min / max are integer atomics (exact, order-independent).
__device__ inline unsigned f32_order_key(float f) {
  const unsigned b = __float_as_uint(f);
  return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}
__device__ inline float f32_from_order_key(unsigned k) { return __uint_as_float((k & 0x80000000u) ? (k & 0x7FFFFFFFu) : ~k); }

// Range probe on a network's output: flags[blockIdx.x] = 1 if any value this workgroup scanned is inf / NaN, else 0.  The parity mode carries
// fp16 EXPONENT range (hi = fp16(x) overflows to inf above 65504); an overflow anywhere in the network reaches the logits as inf or NaN
// (inf - inf in the next convolution), so one pass over the 2.4 MB of logits (reads served by the L2 right behind the head's stores)
// catches it.  Every pass OVERWRITES all VP_PROBE_BLOCKS words (round 5): nothing is sticky, so no clear -- neither a memset node in the
// captured graph nor a host-side one -- stands between a bad frame and the good frame behind it; the host ORs the words it fetched.
__global__ __launch_bounds__(256) void finite_probe_kernel(const float* src, size_t n, unsigned* flags) {
  bool bad = false;
  const size_t n4 = n >> 2;
  const f32x4_t* s4 = reinterpret_cast<const f32x4_t*>(src);
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
    const f32x4_t v = s4[i];
#pragma unroll
    for (int r = 0; r < 4; ++r) bad |= (__float_as_uint(v[r]) & 0x7F800000u) == 0x7F800000u;
  }
  if (blockIdx.x == 0 && threadIdx.x < (n & 3)) bad |= (__float_as_uint(src[(n4 << 2) + threadIdx.x]) & 0x7F800000u) == 0x7F800000u;
  __shared__ unsigned s_any;
  if (threadIdx.x == 0) s_any = 0u;
  __syncthreads();
  if (bad) s_any = 1u;   // every writer stores the same value
  __syncthreads();
  if (threadIdx.x == 0) flags[blockIdx.x] = s_any;
}

__global__ __launch_bounds__(256) void minmax_f32_kernel(const float* src, size_t n, unsigned* mm) {  // mm[0] = min key, mm[1] = max key
  __shared__ unsigned smin[4], smax[4];
  unsigned lo = 0xFFFFFFFFu, hi = 0u;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
    const unsigned k = f32_order_key(src[i]);
    lo = k < lo ? k : lo;
    hi = k > hi ? k : hi;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const unsigned l2 = __shfl_xor(lo, o), h2 = __shfl_xor(hi, o);
    lo = l2 < lo ? l2 : lo;
    hi = h2 > hi ? h2 : hi;
  }
  if ((threadIdx.x & 63) == 0) {
    smin[threadIdx.x >> 6] = lo;
    smax[threadIdx.x >> 6] = hi;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int q = 1; q < 4; ++q) {
      lo = smin[q] < lo ? smin[q] : lo;
      hi = smax[q] > hi ? smax[q] : hi;
    }
    atomicMin(mm, lo);
    atomicMax(mm + 1, hi);
  }
}

// u8 = saturate(cvRound(x * alpha + beta)) with alpha, beta formed in double and applied in float, as cv::Mat::convertTo does
// (fused multiply-add, the AVX2 / NEON build's v_fma; cvRound = round-half-even), then the 256 x BGR table.
__global__ __launch_bounds__(256) void depth_colorize_kernel(const float* src, size_t n, const unsigned* mm, const uint8_t* lut, uint8_t* dst) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const float mn = f32_from_order_key(mm[0]), mx = f32_from_order_key(mm[1]);
  int v = 0;
  if (mx > mn) {
    const double range = (double)mx - (double)mn;
    const float a = (float)(255.0 / range), b = (float)(-(double)mn * 255.0 / range);
    const int r = __float2int_rn(__builtin_fmaf(src[i], a, b));
    v = r < 0 ? 0 : (r > 255 ? 255 : r);
  }
  dst[3 * i] = lut[3 * v];
  dst[3 * i + 1] = lut[3 * v + 1];
  dst[3 * i + 2] = lut[3 * v + 2];
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
hipError_t launch_preprocess(const PreprocessParams& p, hipStream_t st) {
  VP_LAUNCH(preprocess_kernel, dim3(nblk(p.out_w), p.out_h), dim3(256), 0, st, p);
}
hipError_t launch_pil_resample(const PilResampleParams& p, hipStream_t st) {
  hipLaunchKernelGGL(pil_resample_h_kernel, dim3(nblk(p.out_w), p.in_h), dim3(256), 0, st, p);
  if (hipError_t e = hipGetLastError(); e != hipSuccess) return e;
  VP_LAUNCH(pil_resample_v_kernel, dim3(nblk(p.out_w), p.out_h), dim3(256), 0, st, p);
}
bool ctx_exp_conv1_ok(const CtxExpConv1Params& q) {
  const FcParams& p = q.fc;
  const bool g_ok = q.glanes >= 1 && q.glanes <= 64 && (q.glanes & (q.glanes - 1)) == 0;
  return (q.tile == 2 || q.tile == 8 || q.tile == 16) && g_ok && p.K >= 4 && (p.K & 3) == 0 && p.N == q.cv.H * q.cv.W && (p.w != nullptr) != (p.w8 != nullptr) && p.b && q.cv.w && q.cv.b &&
         (q.cv.out.C & 7) == 0 && p.act_rows == 0 && (p.partial != nullptr || p.x != nullptr) && (size_t)(((p.K + 3) & ~3) + (q.tile + 2) * (q.tile + 2)) * 4 <= 48 * 1024;
}
hipError_t launch_ctx_exp_conv1(const CtxExpConv1Params& q, hipStream_t st) {
  if (!ctx_exp_conv1_ok(q)) return hipErrorInvalidValue;
  const int T = q.tile;
  const unsigned grid = (unsigned)(((q.cv.H + T - 1) / T) * ((q.cv.W + T - 1) / T));
  const size_t lds = (size_t)(((q.fc.K + 3) & ~3) + (T + 2) * (T + 2)) * sizeof(float);
  if (T == 16) {
    VP_LAUNCH(ctx_exp_conv1_kernel<16>, dim3(grid), dim3(256), lds, st, q);
  }
  if (T == 2) {
    VP_LAUNCH(ctx_exp_conv1_kernel<2>, dim3(grid), dim3(256), lds, st, q);
  }
  VP_LAUNCH(ctx_exp_conv1_kernel<8>, dim3(grid), dim3(256), lds, st, q);
}
hipError_t launch_ctx_conv1(const CtxConv1Params& p, hipStream_t st) {
  VP_LAUNCH(ctx_conv1_kernel, dim3(nblk((long long)p.H * p.W * (p.out.C >> 3))), dim3(256), 0, st, p);
}
bool fusion_octets_ok(const FusionParams& p) {
  int units = 0, creal = 0;
  for (int l = 0; l < 5; ++l) {
    if ((p.creal[l] & 7) || p.shift[l] < 0 || p.shift[l] > 6) return false;
    const int n2 = 1 << (2 * p.shift[l]);
    units += (p.creal[l] >> 3) * (n2 < 16 ? n2 : 16);
    creal += p.creal[l];
  }
  return units <= 512 && creal == p.Creal_out && (p.Creal_out & 7) == 0 && (p.out.C & 7) == 0;
}
hipError_t launch_fusion(const FusionParams& p, hipStream_t st) {
  if (p.octets) {
    if (!fusion_octets_ok(p)) return hipErrorInvalidValue;
    VP_LAUNCH(fusion_octet_kernel, dim3(p.out.H * p.out.W), dim3(256), 0, st, p);
  }
  VP_LAUNCH(fusion_kernel, dim3(nblk((long long)p.out.H * p.out.W * p.out.C)), dim3(256), 0, st, p);
}
hipError_t launch_decode_mask(const float* logits, int C, int HW, int mode, uint8_t* out, hipStream_t st) {
  VP_LAUNCH(decode_mask_kernel, dim3(nblk(HW)), dim3(256), 0, st, logits, C, HW, mode, out);
}
hipError_t launch_resize_nearest(const uint8_t* src, int sw, const int* ytab, const int* xtab, int oh, int ow, uint8_t* dst,
                                 hipStream_t st) {
  VP_LAUNCH(resize_nearest_kernel, dim3(nblk(ow), oh), dim3(256), 0, st, src, sw, ytab, xtab, oh, ow, dst);
}
hipError_t launch_viz_blend(const uint8_t* mask, int mw, const int* ytab, const int* xtab, const uint8_t* frame, int stride, int oh, int ow,
                            const uint8_t* lut, int frame_is_rgb, uint8_t* dst, hipStream_t st) {
  VP_LAUNCH(viz_blend_kernel, dim3(nblk(ow), oh), dim3(256), 0, st, mask, mw, ytab, xtab, frame, stride, oh, ow, lut, frame_is_rgb, dst);
}
hipError_t launch_minmax_f32(const float* src, size_t n, unsigned* mm, hipStream_t st) {
  const unsigned blocks = (unsigned)((n + 255) / 256);
  VP_LAUNCH(minmax_f32_kernel, dim3(blocks < 1024u ? (blocks ? blocks : 1u) : 1024u), dim3(256), 0, st, src, n, mm);
}
hipError_t launch_finite_probe(const float* src, size_t n, unsigned* flags, hipStream_t st) {
  VP_LAUNCH(finite_probe_kernel, dim3(VP_PROBE_BLOCKS), dim3(256), 0, st, src, n, flags);  // always the full grid: every word is rewritten
}
hipError_t launch_depth_colorize(const float* src, size_t n, const unsigned* mm, const uint8_t* lut, uint8_t* dst, hipStream_t st) {
  VP_LAUNCH(depth_colorize_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, src, n, mm, lut, dst);
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
