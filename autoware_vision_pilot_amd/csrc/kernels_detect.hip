// AutoSpeed detector pre / post-processing on the device (second half).  The detector network itself is a
// different model family and not part of this library; these are the two stages the reference's engines wrap around it on the CPU
// (VisionPilot/middleware_recipes/common/backends/autospeed/onnxruntime_engine.cpp; tensorrt_engine.cpp holds the same code):
//   letterbox_kernel           preprocessAutoSpeed (:71-113): resize (aspect kept) into a 114-grey canvas, / 255, planes R, G, B
//   detect_decode_nms_kernel   postProcess + applyNMS (:170-290): strict-'>' class argmax from 0, confidence threshold, xywh -> xyxy, letterbox
//                              space -> image, clamp, sort by confidence, greedy same-class IoU suppression
// Integer / index work is bit-exact against oracle/autospeed.py; the fp32 arithmetic is the reference's, expression for expression, with
// IEEE division and without contraction (__f*_rn), so the kept set and its coordinates are bit-identical too.
#include "act_io.hpp"
#include "conv_epilogue.hpp"

namespace vp {

__global__ __launch_bounds__(256) void letterbox_kernel(const LetterboxParams p) {
  const int x = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y;
  if (x >= p.out_w) return;
  const int rx = x - p.pad_x, ry = y - p.pad_y;
  int q[3] = {114, 114, 114};  // cv::Scalar(114, 114, 114)
  if ((unsigned)rx < (unsigned)p.new_w && (unsigned)ry < (unsigned)p.new_h) {
    const int4 xt = *reinterpret_cast<const int4*>(p.xtab + 4 * rx);
    const int4 yt = *reinterpret_cast<const int4*>(p.ytab + 4 * ry);
    const uint8_t* r0 = p.frame + (size_t)yt.x * p.stride;
    const uint8_t* r1 = p.frame + (size_t)yt.y * p.stride;
#pragma unroll
    for (int c = 0; c < 3; ++c) {  // the integer bilinear of preprocess_kernel (kernels_misc.hip)
      const int s0 = (int)r0[xt.x * 3 + c] * xt.z + (int)r0[xt.y * 3 + c] * xt.w;
      const int s1 = (int)r1[xt.x * 3 + c] * xt.z + (int)r1[xt.y * 3 + c] * xt.w;
      q[c] = min(max((((yt.z * (s0 >> 4)) >> 16) + ((yt.w * (s1 >> 4)) >> 16) + 2) >> 2, 0), 255);
    }
  }
#pragma unroll
  for (int c = 0; c < 3; ++c)  // plane c = source byte 2 - c (BGR -> RGB); convertTo(CV_32F, 1 / 255)
    p.out[((size_t)c * p.out_h + y) * p.out_w + x] = __fmul_rn((float)q[2 - c], (float)(1.0 / 255.0));
}

hipError_t launch_letterbox(const LetterboxParams& p, hipStream_t st) {
  if (!p.frame || !p.xtab || !p.ytab || !p.out || p.new_w < 1 || p.new_h < 1 || p.pad_x < 0 || p.pad_y < 0 || p.pad_x + p.new_w > p.out_w ||
      p.pad_y + p.new_h > p.out_h)
    return hipErrorInvalidValue;
  VP_LAUNCH(letterbox_kernel, dim3((p.out_w + 255) / 256, p.out_h), dim3(256), 0, st, p);
}

// ------------------------------------------------------------------------------------------------------------- decode + NMS
// One workgroup of 1024 threads: the stage is a few thousand boxes of 8 floats and inherently sequential at its end (greedy NMS).
//   A  candidates: per box the class argmax and the threshold; survivors get a 64-bit key (confidence bits << 32 | ~box index) in LDS.
//      Confidences are >= +0 (the argmax starts from 0), so their bit patterns order like the floats; ~index makes equal confidences
//      keep the box order (the reference's std::sort leaves ties unspecified).
//   B  bitonic sort of the keys in LDS, descending (<= 16384 keys = 128 KB).
//   C  the sorted candidates' boxes in image coordinates -> scratch.
//   D  greedy suppression: a kept box i suppresses later same-class boxes with IoU > threshold, all threads striding over j; boxes that
//      are already suppressed cost nothing (the flag is uniform after the last barrier), so the barriers number the KEPT boxes.
//   E  compaction in order (per-thread chunks, scan in LDS).
constexpr int kDetThreads = 1024;

__device__ __forceinline__ float det_iou(const f32x4_t a, const f32x4_t b) {  // computeIoU, onnxruntime_engine.cpp:239-255
  const float iw = fmaxf(0.0f, __fsub_rn(fminf(a[2], b[2]), fmaxf(a[0], b[0])));
  const float ih = fmaxf(0.0f, __fsub_rn(fminf(a[3], b[3]), fmaxf(a[1], b[1])));
  const float inter = __fmul_rn(iw, ih);
  const float area_a = __fmul_rn(__fsub_rn(a[2], a[0]), __fsub_rn(a[3], a[1]));
  const float area_b = __fmul_rn(__fsub_rn(b[2], b[0]), __fsub_rn(b[3], b[1]));
  const float uni = __fsub_rn(__fadd_rn(area_a, area_b), inter);
  return uni > 0.0f ? __fdiv_rn(inter, uni) : 0.0f;
}

__global__ __launch_bounds__(kDetThreads) void detect_decode_nms_kernel(const DetectParams p, const int P_max) {
  extern __shared__ __attribute__((aligned(16))) unsigned char det_smem[];
  unsigned long long* const keys = reinterpret_cast<unsigned long long*>(det_smem);      // [P_max]
  unsigned char* const flags = det_smem + (size_t)P_max * 8;                              // [P_max] 1 = suppressed
  __shared__ int n_cand;
  __shared__ int scan[kDetThreads];
  const int tid = threadIdx.x, nb = p.num_boxes;
  if (tid == 0) n_cand = 0;
  __syncthreads();
  // ---- A
  for (int i = tid; i < nb; i += kDetThreads) {
    float best = 0.0f;
    int cls = -1;
    for (int c = 4; c < p.num_attrs; ++c) {
      const float s = p.raw[(size_t)c * nb + i];
      if (s > best) {
        best = s;
        cls = c - 4;
      }
    }
    if (best < p.conf_thresh) continue;
    const int slot = atomicAdd(&n_cand, 1);
    keys[slot] = ((unsigned long long)__float_as_uint(best) << 32) | (unsigned long long)(0xFFFFFFFFu - (unsigned)i);
    p.cls[i] = cls;
  }
  __syncthreads();
  const int n = n_cand;
  int P = 1;
  while (P < n) P <<= 1;
  for (int t = n + tid; t < P; t += kDetThreads) keys[t] = 0ull;
  __syncthreads();
  // ---- B: descending bitonic sort
  for (int k = 2; k <= P; k <<= 1)
    for (int j = k >> 1; j > 0; j >>= 1) {
      for (int t = tid; t < P; t += kDetThreads) {
        const int u = t ^ j;
        if (u > t) {
          const unsigned long long a = keys[t], b = keys[u];
          if (((t & k) == 0) ? a < b : a > b) {
            keys[t] = b;
            keys[u] = a;
          }
        }
      }
      __syncthreads();
    }
  // ---- C: boxes of the sorted candidates (postProcess :198-222)
  const float px = (float)p.pad_x, py = (float)p.pad_y, ow = (float)p.orig_w, oh = (float)p.orig_h;
  for (int r = tid; r < n; r += kDetThreads) {
    const int i = (int)(0xFFFFFFFFu - (unsigned)(keys[r] & 0xFFFFFFFFull));
    const float cx = p.raw[i], cy = p.raw[(size_t)nb + i], w = p.raw[(size_t)2 * nb + i], h = p.raw[(size_t)3 * nb + i];
    const float hw = __fdiv_rn(w, 2.0f), hh = __fdiv_rn(h, 2.0f);
    f32x4_t b;
    b[0] = fmaxf(0.0f, fminf(ow, __fdiv_rn(__fsub_rn(__fsub_rn(cx, hw), px), p.scale)));
    b[1] = fmaxf(0.0f, fminf(oh, __fdiv_rn(__fsub_rn(__fsub_rn(cy, hh), py), p.scale)));
    b[2] = fmaxf(0.0f, fminf(ow, __fdiv_rn(__fsub_rn(__fadd_rn(cx, hw), px), p.scale)));
    b[3] = fmaxf(0.0f, fminf(oh, __fdiv_rn(__fsub_rn(__fadd_rn(cy, hh), py), p.scale)));
    *reinterpret_cast<f32x4_t*>(p.boxes + (size_t)r * 4) = b;
    flags[r] = 0;
  }
  __threadfence_block();
  __syncthreads();
  // ---- D: greedy same-class suppression (applyNMS :257-290)
  for (int i = 0; i < n; ++i) {
    if (flags[i]) continue;  // uniform: nothing was written since the last barrier
    const f32x4_t bi = *reinterpret_cast<const f32x4_t*>(p.boxes + (size_t)i * 4);
    const int ci = p.cls[(int)(0xFFFFFFFFu - (unsigned)(keys[i] & 0xFFFFFFFFull))];
    for (int j = i + 1 + tid; j < n; j += kDetThreads) {
      if (flags[j]) continue;
      if (p.cls[(int)(0xFFFFFFFFu - (unsigned)(keys[j] & 0xFFFFFFFFull))] != ci) continue;
      if (det_iou(bi, *reinterpret_cast<const f32x4_t*>(p.boxes + (size_t)j * 4)) > p.iou_thresh) flags[j] = 1;
    }
    __syncthreads();
  }
  // ---- E: the kept boxes, in order
  const int per = (n + kDetThreads - 1) / kDetThreads, r0 = min(n, tid * per), r1 = min(n, r0 + per);
  int mine = 0;
  for (int r = r0; r < r1; ++r) mine += flags[r] ? 0 : 1;
  scan[tid] = mine;
  __syncthreads();
  if (tid == 0) {
    int run = 0;
    for (int t = 0; t < kDetThreads; ++t) {
      const int v = scan[t];
      scan[t] = run;
      run += v;
    }
    p.count[0] = run;
    p.count[1] = n;
  }
  __syncthreads();
  int pos = scan[tid];
  for (int r = r0; r < r1; ++r) {
    if (flags[r]) continue;
    if (pos < p.out_cap) {
      const f32x4_t b = *reinterpret_cast<const f32x4_t*>(p.boxes + (size_t)r * 4);
      const int i = (int)(0xFFFFFFFFu - (unsigned)(keys[r] & 0xFFFFFFFFull));
      Detection d;
      d.x1 = b[0];
      d.y1 = b[1];
      d.x2 = b[2];
      d.y2 = b[3];
      d.confidence = __uint_as_float((unsigned)(keys[r] >> 32));
      d.class_id = p.cls[i];
      p.out[pos] = d;
    }
    ++pos;
  }
}

hipError_t launch_detect_decode_nms(const DetectParams& p, hipStream_t st) {
  if (!p.raw || !p.boxes || !p.cls || !p.out || !p.count || p.num_attrs < 5 || p.num_boxes < 1 || p.num_boxes > kDetectMaxBoxes || p.out_cap < 0 ||
      !(p.scale > 0.0f))
    return hipErrorInvalidValue;
  int P = 1;
  while (P < p.num_boxes) P <<= 1;
  const size_t lds = (size_t)P * 9;
  static LdsAttrOnce once;
  if (hipError_t e = set_max_dynamic_lds(once, reinterpret_cast<const void*>(detect_decode_nms_kernel), 9 * kDetectMaxBoxes); e != hipSuccess) return e;
  hipLaunchKernelGGL(detect_decode_nms_kernel, dim3(1), dim3(kDetThreads), lds, st, p, P);
  return hipGetLastError();
}

}  // namespace vp
