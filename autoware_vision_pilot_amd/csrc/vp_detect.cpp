// vp_detect_*: the AutoSpeed detector's letterbox preprocess and decode + NMS on the device behind the C ABI (include/vp_hip.h; SURVEY.md N4;
// VisionPilot/middleware_recipes/common/backends/autospeed/onnxruntime_engine.cpp:71-113, :170-290).  The detector network itself is not
// part of this library: a host runs it with its own runtime between the two calls (or hands its output tensor over on the device).
#include <algorithm>
#include <cstring>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/vp_hip.h"
#include "engine.hpp"
#include "kernels.hpp"

static_assert(sizeof(vp_detection) == sizeof(vp::Detection) && sizeof(vp_detection) == 24, "vp_detection is the reference's Detection");

struct vp_detect {
  int gpu = 0, net_h = 0, net_w = 0, max_boxes = 0, max_attrs = 0;
  hipStream_t stream = nullptr;
  uint8_t* d_frame = nullptr;
  size_t frame_cap = 0;
  uint8_t* h_frame = nullptr;  // pinned staging
  size_t h_frame_cap = 0;
  int* d_xtab = nullptr;
  int* d_ytab = nullptr;
  int tab_w = 0, tab_h = 0, tab_cap_w = 0, tab_cap_h = 0;
  float* d_input = nullptr;  // [3][net_h][net_w]
  float* d_raw = nullptr;    // [max_attrs][max_boxes]
  float* d_boxes = nullptr;
  int* d_cls = nullptr;
  vp::Detection* d_out = nullptr;
  int* d_count = nullptr;
  vp::Detection* h_out = nullptr;  // pinned [max_boxes]
  int* h_count = nullptr;          // pinned [2]
  // letterbox geometry of the last preprocessed frame (scale_, pad_x_, pad_y_, orig_width_, orig_height_ of the reference engine)
  float scale = 0.0f;
  int pad_x = 0, pad_y = 0, orig_w = 0, orig_h = 0;
  std::string err;

  ~vp_detect() {
    (void)hipSetDevice(gpu);
    if (stream) (void)hipStreamSynchronize(stream);
    for (void* q : {(void*)d_frame, (void*)d_xtab, (void*)d_ytab, (void*)d_input, (void*)d_raw, (void*)d_boxes, (void*)d_cls, (void*)d_out, (void*)d_count})
      if (q) (void)hipFree(q);
    for (void* q : {(void*)h_frame, (void*)h_out, (void*)h_count})
      if (q) (void)hipHostFree(q);
    if (stream) (void)hipStreamDestroy(stream);
  }
};

namespace {

void set_err(char* err, size_t n, const std::string& msg) {
  if (err && n) {
    std::strncpy(err, msg.c_str(), n - 1);
    err[n - 1] = 0;
  }
}

template <class F>
int guarded(vp_detect* d, F&& f) {
  if (!d) return VP_ERR_ARG;
  try {
    VP_HIP_CHECK(hipSetDevice(d->gpu));
    return f();
  } catch (const std::invalid_argument& ex) {
    d->err = ex.what();
    return VP_ERR_ARG;
  } catch (const std::exception& ex) {
    d->err = ex.what();
    return VP_ERR_HIP;
  }
}

}  // namespace

extern "C" {

int vp_detect_create(vp_detect** out, int gpu_id, int net_h, int net_w, int max_boxes, int max_attrs, char* err, size_t err_len) {
  if (!out || net_h < 1 || net_w < 1 || net_h > 8192 || net_w > 8192 || max_boxes < 1 || max_boxes > vp::kDetectMaxBoxes || max_attrs < 5 || max_attrs > 4096) {
    set_err(err, err_len, "vp_detect_create: bad argument (max_boxes <= 16384, max_attrs >= 5)");
    return VP_ERR_ARG;
  }
  try {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n < 1) throw std::runtime_error("no HIP device visible: the detector stages have no CPU fallback");
    if (gpu_id < 0 || gpu_id >= n) throw std::invalid_argument("gpu_id out of range");
    auto d = std::make_unique<vp_detect>();
    d->gpu = gpu_id;
    d->net_h = net_h;
    d->net_w = net_w;
    d->max_boxes = max_boxes;
    d->max_attrs = max_attrs;
    VP_HIP_CHECK(hipSetDevice(gpu_id));
    VP_HIP_CHECK(hipStreamCreateWithFlags(&d->stream, hipStreamNonBlocking));
    VP_HIP_CHECK(hipMalloc(&d->d_input, (size_t)3 * net_h * net_w * sizeof(float)));
    VP_HIP_CHECK(hipMalloc(&d->d_raw, (size_t)max_attrs * max_boxes * sizeof(float)));
    VP_HIP_CHECK(hipMalloc(&d->d_boxes, (size_t)max_boxes * 4 * sizeof(float)));
    VP_HIP_CHECK(hipMalloc(&d->d_cls, (size_t)max_boxes * sizeof(int)));
    VP_HIP_CHECK(hipMalloc(&d->d_out, (size_t)max_boxes * sizeof(vp::Detection)));
    VP_HIP_CHECK(hipMalloc(&d->d_count, 2 * sizeof(int)));
    VP_HIP_CHECK(hipHostMalloc(&d->h_out, (size_t)max_boxes * sizeof(vp::Detection)));
    VP_HIP_CHECK(hipHostMalloc(&d->h_count, 2 * sizeof(int)));
    *out = d.release();
    return VP_OK;
  } catch (const std::invalid_argument& ex) {
    set_err(err, err_len, ex.what());
    return VP_ERR_ARG;
  } catch (const std::exception& ex) {
    set_err(err, err_len, ex.what());
    return VP_ERR_HIP;
  }
}

void vp_detect_destroy(vp_detect* d) { delete d; }

const char* vp_detect_last_error(const vp_detect* d) { return d ? d->err.c_str() : "null handle"; }

int vp_detect_preprocess(vp_detect* d, const uint8_t* bgr, int h, int w, int stride_bytes, float* dst_host_chw) {
  return guarded(d, [&]() -> int {
    if (!bgr || h < 1 || w < 1 || h > 16384 || w > 16384 || stride_bytes < 3 * w) throw std::invalid_argument("vp_detect_preprocess: bad frame geometry");
    // preprocessAutoSpeed :78-98 -- fp32 scale, truncated new size, centred paste
    const float scale = std::min((float)d->net_w / (float)w, (float)d->net_h / (float)h);
    const int new_w = (int)((float)w * scale), new_h = (int)((float)h * scale);
    if (new_w < 1 || new_h < 1 || new_w > d->net_w || new_h > d->net_h) throw std::invalid_argument("vp_detect_preprocess: the frame collapses under the letterbox scale");
    const size_t bytes = (size_t)(h - 1) * stride_bytes + (size_t)3 * w;   // a strided view may END with its last pixel: never read a full last stride
    if (bytes > d->frame_cap) {
      if (d->d_frame) VP_HIP_CHECK(hipFree(d->d_frame));
      d->d_frame = nullptr;
      d->frame_cap = 0;
      VP_HIP_CHECK(hipMalloc(&d->d_frame, bytes));
      d->frame_cap = bytes;
    }
    if (bytes > d->h_frame_cap) {
      if (d->h_frame) VP_HIP_CHECK(hipHostFree(d->h_frame));
      d->h_frame = nullptr;
      d->h_frame_cap = 0;
      VP_HIP_CHECK(hipHostMalloc(&d->h_frame, bytes));
      d->h_frame_cap = bytes;
    }
    VP_HIP_CHECK(hipStreamSynchronize(d->stream));  // the staging buffer and the tables may still be read by the previous call
    std::memcpy(d->h_frame, bgr, bytes);
    VP_HIP_CHECK(hipMemcpyAsync(d->d_frame, d->h_frame, bytes, hipMemcpyHostToDevice, d->stream));
    if (d->tab_w != w || d->tab_h != h) {
      std::vector<int> xt, yt;
      vp::linear_taps_u8(w, new_w, &xt);
      vp::linear_taps_u8(h, new_h, &yt);
      if (new_w > d->tab_cap_w) {
        if (d->d_xtab) VP_HIP_CHECK(hipFree(d->d_xtab));
        d->d_xtab = nullptr;
        VP_HIP_CHECK(hipMalloc(&d->d_xtab, (size_t)d->net_w * 4 * sizeof(int)));
        d->tab_cap_w = d->net_w;
      }
      if (new_h > d->tab_cap_h) {
        if (d->d_ytab) VP_HIP_CHECK(hipFree(d->d_ytab));
        d->d_ytab = nullptr;
        VP_HIP_CHECK(hipMalloc(&d->d_ytab, (size_t)d->net_h * 4 * sizeof(int)));
        d->tab_cap_h = d->net_h;
      }
      // on the detector's own stream, then waited for: hipMemcpy would use the legacy stream, which the runtime refuses while any thread captures a graph
      VP_HIP_CHECK(hipMemcpyAsync(d->d_xtab, xt.data(), xt.size() * sizeof(int), hipMemcpyHostToDevice, d->stream));
      VP_HIP_CHECK(hipMemcpyAsync(d->d_ytab, yt.data(), yt.size() * sizeof(int), hipMemcpyHostToDevice, d->stream));
      VP_HIP_CHECK(hipStreamSynchronize(d->stream));
      d->tab_w = w;
      d->tab_h = h;
    }
    vp::LetterboxParams p{};
    p.frame = d->d_frame;
    p.stride = stride_bytes;
    p.xtab = d->d_xtab;
    p.ytab = d->d_ytab;
    p.new_w = new_w;
    p.new_h = new_h;
    p.pad_x = (d->net_w - new_w) / 2;
    p.pad_y = (d->net_h - new_h) / 2;
    p.out_h = d->net_h;
    p.out_w = d->net_w;
    p.out = d->d_input;
    VP_HIP_CHECK(vp::launch_letterbox(p, d->stream));
    d->scale = scale;
    d->pad_x = p.pad_x;
    d->pad_y = p.pad_y;
    d->orig_w = w;
    d->orig_h = h;
    if (dst_host_chw) VP_HIP_CHECK(hipMemcpyAsync(dst_host_chw, d->d_input, (size_t)3 * d->net_h * d->net_w * sizeof(float), hipMemcpyDeviceToHost, d->stream));
    VP_HIP_CHECK(hipStreamSynchronize(d->stream));
    return VP_OK;
  });
}

int vp_detect_input_device(const vp_detect* d, void** dev_f32_chw) {
  if (!d || !dev_f32_chw) return VP_ERR_ARG;
  *dev_f32_chw = d->d_input;
  return VP_OK;
}

int vp_detect_letterbox(const vp_detect* d, float* scale, int* pad_x, int* pad_y) {
  if (!d || d->orig_w == 0) return VP_ERR_ARG;
  if (scale) *scale = d->scale;
  if (pad_x) *pad_x = d->pad_x;
  if (pad_y) *pad_y = d->pad_y;
  return VP_OK;
}

int vp_detect_set_letterbox(vp_detect* d, float scale, int pad_x, int pad_y, int orig_w, int orig_h) {
  if (!d || !(scale > 0.0f) || orig_w < 1 || orig_h < 1) return VP_ERR_ARG;
  d->scale = scale;
  d->pad_x = pad_x;
  d->pad_y = pad_y;
  d->orig_w = orig_w;
  d->orig_h = orig_h;
  return VP_OK;
}

int vp_detect_postprocess(vp_detect* d, const float* raw, int raw_on_device, int num_attrs, int num_boxes, float conf_thresh, float iou_thresh,
                          vp_detection* out, int out_cap, int* count) {
  return guarded(d, [&]() -> int {
    if (!raw || !count || out_cap < 0 || (out_cap > 0 && !out) || num_attrs < 5 || num_attrs > d->max_attrs || num_boxes < 1 || num_boxes > d->max_boxes)
      throw std::invalid_argument("vp_detect_postprocess: bad argument (num_attrs / num_boxes within the sizes given to vp_detect_create)");
    if (d->orig_w == 0) throw std::invalid_argument("vp_detect_postprocess: no letterbox geometry yet (vp_detect_preprocess or vp_detect_set_letterbox first)");
    const float* src = raw;
    if (!raw_on_device) {
      VP_HIP_CHECK(hipMemcpyAsync(d->d_raw, raw, (size_t)num_attrs * num_boxes * sizeof(float), hipMemcpyHostToDevice, d->stream));
      src = d->d_raw;
    }
    vp::DetectParams p{};
    p.raw = src;
    p.num_attrs = num_attrs;
    p.num_boxes = num_boxes;
    p.conf_thresh = conf_thresh;
    p.iou_thresh = iou_thresh;
    p.scale = d->scale;
    p.pad_x = d->pad_x;
    p.pad_y = d->pad_y;
    p.orig_w = d->orig_w;
    p.orig_h = d->orig_h;
    p.boxes = d->d_boxes;
    p.cls = d->d_cls;
    p.out = d->d_out;
    p.out_cap = std::min(out_cap, d->max_boxes);
    p.count = d->d_count;
    VP_HIP_CHECK(vp::launch_detect_decode_nms(p, d->stream));
    VP_HIP_CHECK(hipMemcpyAsync(d->h_count, d->d_count, 2 * sizeof(int), hipMemcpyDeviceToHost, d->stream));
    VP_HIP_CHECK(hipStreamSynchronize(d->stream));
    const int kept = d->h_count[0], n = std::min(kept, p.out_cap);
    if (n > 0) {
      VP_HIP_CHECK(hipMemcpyAsync(d->h_out, d->d_out, (size_t)n * sizeof(vp::Detection), hipMemcpyDeviceToHost, d->stream));
      VP_HIP_CHECK(hipStreamSynchronize(d->stream));
      std::memcpy(out, d->h_out, (size_t)n * sizeof(vp::Detection));
    }
    *count = kept;
    return VP_OK;
  });
}

}  // extern "C"
