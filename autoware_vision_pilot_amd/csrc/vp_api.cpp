// C ABI of libvp_hip.so (declarations + reference citations: include/vp_hip.h).  No exception crosses it.
#include <cctype>
#include <cstdio>
#include <cstring>
#include <fstream>

#include "vp_handle.hpp"
#include "engine_internal.hpp"

namespace {

void set_err(char* err, size_t n, const std::string& msg) {
  if (err && n) {
    std::snprintf(err, n, "%s", msg.c_str());
  }
}

template <class F>
int guarded(vp_engine* e, F&& f) {
  if (!e || !e->impl) return VP_ERR_ARG;
  try {
    f(*e->impl);
    return VP_OK;
  } catch (const std::invalid_argument& ex) {
    e->err = ex.what();
    return VP_ERR_ARG;
  } catch (const vp::RangeError& ex) {
    e->err = ex.what();
    return VP_ERR_RANGE;
  } catch (const std::exception& ex) {
    e->err = ex.what();
    return std::strncmp(ex.what(), "HIP error", 9) == 0 ? VP_ERR_HIP : VP_ERR_STATE;
  }
}

int create_impl(vp_engine** out, int kind, const void* blob, size_t bytes, int precision, int gpu_id, char* err, size_t err_len,
                vp::Engine* base = nullptr, int frames = 1, int frame_index = 0) {
  if (!out) return VP_ERR_ARG;
  *out = nullptr;
  try {
    vp::WeightBlob wb;
    try {
      wb.parse(blob, bytes);
    } catch (const std::exception& ex) {
      set_err(err, err_len, ex.what());
      return VP_ERR_WEIGHTS;
    }
    auto h = std::make_unique<vp_engine>();
    try {
      h->impl = std::make_unique<vp::Engine>(kind, &wb, precision, gpu_id, base, frames, frame_index);
    } catch (const std::invalid_argument& ex) {
      set_err(err, err_len, ex.what());
      return VP_ERR_ARG;
    } catch (const vp::RangeError& ex) {   // a folded weight beyond the fp16 range
      set_err(err, err_len, ex.what());
      return VP_ERR_RANGE;
    } catch (const std::exception& ex) {
      set_err(err, err_len, ex.what());
      return std::strstr(ex.what(), "weight") ? VP_ERR_WEIGHTS : VP_ERR_HIP;
    }
    *out = h.release();
    return VP_OK;
  } catch (...) {
    set_err(err, err_len, "unknown failure");
    return VP_ERR_STATE;
  }
}

// Weight file -> VPW1 bytes.  `*.onnx` (the reference's `model_path`, ROS2/models/config/autoseg.yaml:3) goes through
// the native ONNX reader; anything else is read as a VPW1 blob.
int read_weights(const char* path, std::vector<char>& buf, char* err, size_t err_len) {
  const std::string p(path);
  auto ends_with = [&](const char* suf) {
    const size_t n = std::strlen(suf);
    if (p.size() < n) return false;
    for (size_t i = 0; i < n; ++i)
      if (std::tolower((unsigned char)p[p.size() - n + i]) != suf[i]) return false;
    return true;
  };
  if (ends_with(".onnx")) {
    try {
      buf = vp::onnx_to_blob(p);
      return VP_OK;
    } catch (const std::exception& ex) {
      set_err(err, err_len, ex.what());
      return VP_ERR_WEIGHTS;
    }
  }
  std::ifstream f(path, std::ios::binary | std::ios::ate);
  if (!f) {
    set_err(err, err_len, std::string("cannot open weight file: ") + path);
    return VP_ERR_WEIGHTS;
  }
  const std::streamsize n = f.tellg();
  f.seekg(0);
  buf.resize((size_t)n);
  if (n && !f.read(buf.data(), n)) {
    set_err(err, err_len, "short read on weight file");
    return VP_ERR_WEIGHTS;
  }
  return VP_OK;
}

}  // namespace

extern "C" {

// Stateless twin of MasksVisualizationKernels::createMaskFromTensor{CUDA,HIP} / createEgoLanesMaskFromTensorCUDA
// (common/include/masks_visualization_kernels.hpp:14-45, masks_viz.hip.cpp:41-97): HOST logits up, decode kernel, mask down.
// Same shape of work as the reference helper (allocate, copy, launch, copy, free); the engine-resident path (vp_mask_u8)
// never re-uploads the logits.
int vp_decode_logits_host(int gpu_id, const float* logits_nchw, int channels, int h, int w, int decode_mode, uint8_t* mask_out) {
  if (!logits_nchw || !mask_out || channels < 1 || h < 1 || w < 1 || decode_mode < 0 || decode_mode > 2) return VP_ERR_ARG;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0 || gpu_id < 0 || gpu_id >= ndev) return VP_ERR_HIP;
  if (hipSetDevice(gpu_id) != hipSuccess) return VP_ERR_HIP;
  const size_t hw = (size_t)h * w, in_bytes = hw * channels * sizeof(float);
  float* d_in = nullptr;
  uint8_t* d_out = nullptr;
  hipStream_t st = nullptr;
  int rc = VP_ERR_HIP;
  if (hipMalloc(reinterpret_cast<void**>(&d_in), in_bytes) == hipSuccess && hipMalloc(reinterpret_cast<void**>(&d_out), hw) == hipSuccess &&
      hipStreamCreateWithFlags(&st, hipStreamNonBlocking) == hipSuccess &&
      hipMemcpyAsync(d_in, logits_nchw, in_bytes, hipMemcpyHostToDevice, st) == hipSuccess &&
      vp::launch_decode_mask(d_in, channels, (int)hw, decode_mode, d_out, st) == hipSuccess &&
      hipMemcpyAsync(mask_out, d_out, hw, hipMemcpyDeviceToHost, st) == hipSuccess && hipStreamSynchronize(st) == hipSuccess)
    rc = VP_OK;
  if (st) hipStreamDestroy(st);
  if (d_in) hipFree(d_in);
  if (d_out) hipFree(d_out);
  return rc;
}


int vp_create_from_memory(vp_engine** out, int model_kind, const void* blob, size_t blob_bytes, int precision, int gpu_id, char* err,
                          size_t err_len) {
  if (!blob) {
    set_err(err, err_len, "null weight blob");
    return VP_ERR_ARG;
  }
  return create_impl(out, model_kind, blob, blob_bytes, precision, gpu_id, err, err_len);
}

int vp_create(vp_engine** out, int model_kind, const char* weights_path, int precision, int gpu_id, char* err, size_t err_len) {
  if (!weights_path || !*weights_path) {
    set_err(err, err_len, "No path to weight file provided");  // scene_seg_infer.py:33
    return VP_ERR_ARG;
  }
  std::vector<char> buf;
  if (const int rc = read_weights(weights_path, buf, err, err_len)) return rc;
  return create_impl(out, model_kind, buf.data(), buf.size(), precision, gpu_id, err, err_len);
}

int vp_create_shared_from_memory(vp_engine** out, vp_engine* base, int model_kind, const void* blob, size_t blob_bytes, int precision,
                                 int gpu_id, char* err, size_t err_len) {
  if (!blob || !base || !base->impl) {
    set_err(err, err_len, !blob ? "null weight blob" : "null base engine");
    return VP_ERR_ARG;
  }
  return create_impl(out, model_kind, blob, blob_bytes, precision, gpu_id, err, err_len, base->impl.get());
}

int vp_create_shared(vp_engine** out, vp_engine* base, int model_kind, const char* weights_path, int precision, int gpu_id, char* err,
                     size_t err_len) {
  if (!weights_path || !*weights_path) {
    set_err(err, err_len, "No path to weight file provided");
    return VP_ERR_ARG;
  }
  std::vector<char> buf;
  if (const int rc = read_weights(weights_path, buf, err, err_len)) return rc;
  return vp_create_shared_from_memory(out, base, model_kind, buf.data(), buf.size(), precision, gpu_id, err, err_len);
}

int vp_create_batched_from_memory(vp_engine** out, int model_kind, const void* blob, size_t blob_bytes, int precision, int gpu_id, int frames,
                                  char* err, size_t err_len) {
  if (!blob) {
    set_err(err, err_len, "null weight blob");
    return VP_ERR_ARG;
  }
  return create_impl(out, model_kind, blob, blob_bytes, precision, gpu_id, err, err_len, nullptr, frames, 0);
}

int vp_create_batched(vp_engine** out, int model_kind, const char* weights_path, int precision, int gpu_id, int frames, char* err,
                      size_t err_len) {
  if (!weights_path || !*weights_path) {
    set_err(err, err_len, "No path to weight file provided");
    return VP_ERR_ARG;
  }
  std::vector<char> buf;
  if (const int rc = read_weights(weights_path, buf, err, err_len)) return rc;
  return create_impl(out, model_kind, buf.data(), buf.size(), precision, gpu_id, err, err_len, nullptr, frames, 0);
}

int vp_create_shared_frame_from_memory(vp_engine** out, vp_engine* base, int frame_index, int model_kind, const void* blob, size_t blob_bytes,
                                       int precision, int gpu_id, char* err, size_t err_len) {
  if (!blob || !base || !base->impl) {
    set_err(err, err_len, !blob ? "null weight blob" : "null base engine");
    return VP_ERR_ARG;
  }
  return create_impl(out, model_kind, blob, blob_bytes, precision, gpu_id, err, err_len, base->impl.get(), 1, frame_index);
}

int vp_create_shared_frame(vp_engine** out, vp_engine* base, int frame_index, int model_kind, const char* weights_path, int precision,
                           int gpu_id, char* err, size_t err_len) {
  if (!weights_path || !*weights_path) {
    set_err(err, err_len, "No path to weight file provided");
    return VP_ERR_ARG;
  }
  std::vector<char> buf;
  if (const int rc = read_weights(weights_path, buf, err, err_len)) return rc;
  return vp_create_shared_frame_from_memory(out, base, frame_index, model_kind, buf.data(), buf.size(), precision, gpu_id, err, err_len);
}

int vp_frames(const vp_engine* e) { return (e && e->impl) ? e->impl->frames() : -1; }

int vp_upload_frame_n(vp_engine* e, int index, const uint8_t* frame, int h, int w, int stride_bytes) {
  return guarded(e, [&](vp::Engine& g) { g.upload_frame(frame, h, w, stride_bytes, index); });
}

int vp_convert_onnx(const char* onnx_path, const char* vpw_path, char* err, size_t err_len) {
  if (!onnx_path || !*onnx_path || !vpw_path || !*vpw_path) {
    set_err(err, err_len, "No path to weight file provided");
    return VP_ERR_ARG;
  }
  try {
    const std::vector<char> blob = vp::onnx_to_blob(onnx_path);
    std::ofstream o(vpw_path, std::ios::binary | std::ios::trunc);
    if (!o || !o.write(blob.data(), (std::streamsize)blob.size())) {
      set_err(err, err_len, std::string("cannot write ") + vpw_path);
      return VP_ERR_WEIGHTS;
    }
    return VP_OK;
  } catch (const std::exception& ex) {
    set_err(err, err_len, ex.what());
    return VP_ERR_WEIGHTS;
  }
}

int vp_shared_level(const vp_engine* e) { return (e && e->impl) ? e->impl->shared_level() : -1; }

int vp_infer_shared(vp_engine* e) {
  return guarded(e, [&](vp::Engine& g) {
    if (g.shared_level() == 0) throw std::invalid_argument("vp_infer_shared: not a shared-prefix engine");
    g.enqueue();
    g.fetch_outputs();
  });
}

void vp_destroy(vp_engine* e) { delete e; }

const char* vp_last_error(const vp_engine* e) { return e ? e->err.c_str() : "null engine"; }

int vp_set_input_format(vp_engine* e, int pixel_format, int plane_order) {
  return guarded(e, [&](vp::Engine& g) { g.set_input_format(pixel_format, plane_order); });
}
int vp_set_decode_mode(vp_engine* e, int mode) {
  return guarded(e, [&](vp::Engine& g) { g.set_decode_mode(mode); });
}
int vp_set_resize_mode(vp_engine* e, int mode) {
  return guarded(e, [&](vp::Engine& g) { g.set_resize_mode(mode); });
}
// host only: the tap tables the VP_RESIZE_PIL_* modes use for one axis (tests pin them against Pillow's through the oracle)
int vp_resample_coeffs(int in_size, int out_size, int resize_mode, int* bounds, int* coeffs, int coeffs_cap) {
  if (in_size < 1 || out_size < 1 || (resize_mode != 1 && resize_mode != 2) || !bounds || !coeffs) return VP_ERR_ARG;
  std::vector<int> b, k;
  const int ksize = vp::pil_coeffs(in_size, out_size, resize_mode, &b, &k);
  if ((long long)ksize * out_size > coeffs_cap) return VP_ERR_ARG;
  std::copy(b.begin(), b.end(), bounds);
  std::copy(k.begin(), k.end(), coeffs);
  return ksize;
}
int vp_set_norm_form(vp_engine* e, int form) {
  return guarded(e, [&](vp::Engine& g) { g.set_norm_form(form); });
}
int vp_get_norm_form(const vp_engine* e) { return (e && e->impl) ? e->impl->norm_form() : VP_ERR_ARG; }
int vp_set_lane_ring(vp_engine* e, int enable) {
  return guarded(e, [&](vp::Engine& g) { g.set_lane_ring(enable != 0); });
}
int vp_lane_ring_device(const vp_engine* e, void** dev_f32, int* frames_valid) {
  if (!e || !e->impl || !e->impl->lane_ring()) return VP_ERR_STATE;
  if (dev_f32) *dev_f32 = e->impl->dev_lane_ring();
  if (frames_valid) *frames_valid = e->impl->lane_ring_frames();
  return VP_OK;
}
int vp_lane_ring_fetch(vp_engine* e, float* dst_host, int* frames_valid) {
  const int rc = guarded(e, [&](vp::Engine& g) { g.fetch_lane_ring(dst_host); });
  if (rc == VP_OK && frames_valid) *frames_valid = e->impl->lane_ring_frames();
  return rc;
}
int vp_get_resize_mode(const vp_engine* e) { return (e && e->impl) ? e->impl->resize_mode() : VP_ERR_ARG; }
int vp_device_count(void) {
  int n = 0;
  return hipGetDeviceCount(&n) == hipSuccess ? n : 0;
}
int vp_get_decode_mode(const vp_engine* e) { return (e && e->impl) ? e->impl->decode_mode() : VP_ERR_ARG; }
int vp_gpu_id(const vp_engine* e) { return (e && e->impl) ? e->impl->gpu() : VP_ERR_ARG; }
int vp_host_logits_current(const vp_engine* e) { return (e && e->impl) ? (e->impl->host_logits_current() ? 1 : 0) : VP_ERR_ARG; }
int vp_input_hw(const vp_engine* e, int* h, int* w) {
  if (!e || !e->impl || !h || !w) return VP_ERR_ARG;
  *h = e->impl->net_h();
  *w = e->impl->net_w();
  return VP_OK;
}

int vp_infer(vp_engine* e, const uint8_t* frame, int h, int w, int stride_bytes) {
  return guarded(e, [&](vp::Engine& g) {
    g.upload_frame(frame, h, w, stride_bytes);
    g.enqueue();
    g.fetch_outputs();
  });
}
int vp_set_outputs(vp_engine* e, int outputs) {
  return guarded(e, [&](vp::Engine& g) { g.set_outputs(outputs); });
}
int vp_set_finite_check(vp_engine* e, int enable) {
  return guarded(e, [&](vp::Engine& g) { g.set_finite_check(enable != 0); });
}
int vp_set_pinned_staging(vp_engine* e, int enable) {
  return guarded(e, [&](vp::Engine& g) { g.set_pinned_staging(enable != 0); });
}
int vp_register_frames(const void* pool, size_t bytes) {
  try {
    return vp::register_frame_range(pool, bytes);
  } catch (...) {
    return VP_ERR_HIP;
  }
}
int vp_unregister_frames(const void* pool) {
  try {
    return vp::unregister_frame_range(pool);
  } catch (...) {
    return VP_ERR_HIP;
  }
}
int vp_output_shape(const vp_engine* e, int64_t shape[4]) {
  if (!e || !e->impl || !shape) return VP_ERR_ARG;
  if (!e->impl->have_outputs()) {
    const_cast<vp_engine*>(e)->err = "Inference has not been run yet. Call vp_infer() first.";  // onnx_runtime_backend.cpp:86-88
    return VP_ERR_STATE;
  }
  shape[0] = 1;
  shape[1] = e->impl->out_c();
  shape[2] = e->impl->out_h();
  shape[3] = e->impl->out_w();
  return VP_OK;
}
int vp_frame_hw(const vp_engine* e, int* h, int* w) {
  if (!e || !e->impl || !h || !w) return VP_ERR_ARG;
  *h = e->impl->frame_h();
  *w = e->impl->frame_w();
  return VP_OK;
}
// One camera frame through a base engine and its shared-prefix heads: ONE H2D, every network enqueued back to back on the
// base engine's stream, the selected outputs of every engine copied D2H behind them, ONE host synchronisation.
int vp_infer_multi(vp_engine* base, vp_engine* const* shared, int n_shared, const uint8_t* frame, int h, int w, int stride_bytes) {
  if (n_shared < 0 || (n_shared > 0 && !shared)) return VP_ERR_ARG;
  for (int i = 0; i < n_shared; ++i)
    if (!shared[i] || !shared[i]->impl) return VP_ERR_ARG;
  return guarded(base, [&](vp::Engine& g) {
    for (int i = 0; i < n_shared; ++i)
      if (shared[i]->impl->shared_level() == 0 || shared[i]->impl->stream() != g.stream())
        throw std::invalid_argument("vp_infer_multi: every head must be a shared-prefix engine of this base");
    g.upload_frame(frame, h, w, stride_bytes);
    std::vector<vp::Engine*> heads;
    for (int i = 0; i < n_shared; ++i) heads.push_back(shared[i]->impl.get());
    g.enqueue_multi(heads);
    g.enqueue_fetch();
    for (int i = 0; i < n_shared; ++i) shared[i]->impl->enqueue_fetch();
    // one synchronisation, then EVERY member's verdict is collected (and thereby consumed) before anything is thrown: a bad frame is
    // reported once, and no member keeps a stale flag for the next frame
    VP_HIP_CHECK(hipStreamSynchronize(g.stream()));
    bool bad = g.poll_status();
    for (int i = 0; i < n_shared; ++i) bad = shared[i]->impl->poll_status() || bad;
    if (bad)
      throw vp::RangeError("non-finite value (inf / NaN) in a network output of this frame: an activation left the fp16 range of the matrix pipe "
                           "(|x| > 65504) or the input / weights were not finite; outputs of this frame are invalid");
  });
}
int vp_enqueue_multi(vp_engine* base, vp_engine* const* shared, int n_shared) {
  if (n_shared < 0 || (n_shared > 0 && !shared)) return VP_ERR_ARG;
  for (int i = 0; i < n_shared; ++i)
    if (!shared[i] || !shared[i]->impl) return VP_ERR_ARG;
  return guarded(base, [&](vp::Engine& g) {
    std::vector<vp::Engine*> heads;
    for (int i = 0; i < n_shared; ++i) heads.push_back(shared[i]->impl.get());
    g.enqueue_multi(heads);
  });
}
int vp_set_multi_fork(vp_engine* base, int enable) {
  return guarded(base, [&](vp::Engine& g) { g.set_multi_fork(enable != 0); });
}
int vp_infer_pair(vp_engine* e, const uint8_t* prev, const uint8_t* curr, int h, int w, int stride_bytes) {
  return guarded(e, [&](vp::Engine& g) {
    g.upload_frame(prev, h, w, stride_bytes);
    g.prime_previous();
    g.upload_frame(curr, h, w, stride_bytes);
    g.enqueue();
    g.fetch_outputs();
  });
}
int vp_infer_tensor(vp_engine* e, const float* nchw) {
  return guarded(e, [&](vp::Engine& g) {
    g.upload_tensor(nchw);
    g.enqueue();
    g.fetch_outputs();
  });
}
int vp_logits(const vp_engine* e, const float** data, int64_t shape[4]) {
  if (!e || !e->impl || !data || !shape) return VP_ERR_ARG;
  if (!e->impl->have_outputs()) {
    const_cast<vp_engine*>(e)->err = "Inference has not been run yet. Call vp_infer() first.";  // onnx_runtime_backend.cpp:86-88
    return VP_ERR_STATE;
  }
  try {
    *data = e->impl->host_logits();
  } catch (const std::exception& ex) {
    const_cast<vp_engine*>(e)->err = ex.what();
    return VP_ERR_HIP;
  }
  shape[0] = 1;
  shape[1] = e->impl->out_c();
  shape[2] = e->impl->out_h();
  shape[3] = e->impl->out_w();
  return VP_OK;
}
int vp_mask_u8(const vp_engine* e, const uint8_t** data, int* h, int* w) {
  if (!e || !e->impl || !data || !h || !w) return VP_ERR_ARG;
  if (!e->impl->have_outputs()) {
    const_cast<vp_engine*>(e)->err = "Inference has not been run yet. Call vp_infer() first.";
    return VP_ERR_STATE;
  }
  try {
    *data = e->impl->host_mask();
  } catch (const std::exception& ex) {
    const_cast<vp_engine*>(e)->err = ex.what();
    return VP_ERR_HIP;
  }
  *h = e->impl->out_h();
  *w = e->impl->out_w();
  return VP_OK;
}
int vp_mask_resized_u8(vp_engine* e, uint8_t* dst, int h, int w) {
  return guarded(e, [&](vp::Engine& g) { g.mask_resized(dst, h, w); });
}
int vp_depth_resized_f32(vp_engine* e, float* dst, int h, int w) {
  return guarded(e, [&](vp::Engine& g) { g.depth_resized(dst, h, w); });
}
int vp_visualize_depth_bgr8(vp_engine* e, uint8_t* dst, int h, int w) {
  return guarded(e, [&](vp::Engine& g) { g.visualize_depth(dst, h, w); });
}
int vp_visualize_mask_bgr8(vp_engine* e, int viz_type, uint8_t* dst, int h, int w) {
  return guarded(e, [&](vp::Engine& g) { g.visualize_mask(viz_type, dst, h, w); });
}
int vp_input_tensor(vp_engine* e, float* dst) {
  return guarded(e, [&](vp::Engine& g) { g.read_input_tensor(dst); });
}

int vp_upload_frame(vp_engine* e, const uint8_t* frame, int h, int w, int stride_bytes) {
  return guarded(e, [&](vp::Engine& g) { g.upload_frame(frame, h, w, stride_bytes); });
}
int vp_enqueue(vp_engine* e) {
  return guarded(e, [&](vp::Engine& g) { g.enqueue(); });
}
int vp_sync(vp_engine* e) {
  return guarded(e, [&](vp::Engine& g) { g.sync(); });
}
int vp_fetch_outputs(vp_engine* e) {
  return guarded(e, [&](vp::Engine& g) { g.fetch_outputs(); });
}
int vp_device_outputs(const vp_engine* e, void** logits, void** mask) {
  if (!e || !e->impl) return VP_ERR_ARG;
  if (logits) *logits = e->impl->dev_logits();
  if (mask) *mask = e->impl->dev_mask();
  return VP_OK;
}
int vp_use_graph(vp_engine* e, int enable) {
  return guarded(e, [&](vp::Engine& g) { g.use_graph(enable != 0); });
}
int vp_timer_begin(vp_engine* e) {
  return guarded(e, [&](vp::Engine& g) { g.timer_begin(); });
}
int vp_timer_end(vp_engine* e, float* ms) {
  return guarded(e, [&](vp::Engine& g) {
    const float t = g.timer_end();
    if (ms) *ms = t;
  });
}

int vp_layer_count(const vp_engine* e) { return (e && e->impl) ? (int)e->impl->ops().size() : VP_ERR_ARG; }
int vp_layer_info(const vp_engine* e, int i, const char** name, double* flops, double* bytes) {
  if (!e || !e->impl || i < 0 || i >= (int)e->impl->ops().size()) return VP_ERR_ARG;
  const vp::Op& op = e->impl->ops()[i];
  if (name) *name = op.name.c_str();
  if (flops) *flops = op.flops;
  if (bytes) *bytes = op.bytes;
  return VP_OK;
}
int vp_layer_flops_executed(const vp_engine* e, int i, double* flops) {
  if (!e || !e->impl || !flops || i < 0 || i >= (int)e->impl->ops().size()) return VP_ERR_ARG;
  const vp::Op& op = e->impl->ops()[i];
  *flops = op.flops_executed > 0 ? op.flops_executed : op.flops;
  return VP_OK;
}
int vp_layer_kernel(const vp_engine* e, int i, const char** kernel) {
  if (!e || !e->impl || !kernel || i < 0 || i >= (int)e->impl->ops().size()) return VP_ERR_ARG;
  *kernel = e->impl->ops()[i].kernel.c_str();
  return VP_OK;
}
// launch geometry beyond the tag ("nsplit=4", "groups=32", "nsplit=2 wgs=128"; "" when the tag says it all): what vp_plan_hash mixes in third
// AutoSteerOnnxEngine::postProcess (production_release/src/inference/autosteer_engine.cpp:160-185): first maximum wins, angle = class - 30
float vp_autosteer_angle(const float* logits, int classes) {
  if (!logits || classes < 1) return 0.0f;
  int best = 0;
  float best_v = logits[0];
  for (int i = 1; i < classes; ++i)
    if (logits[i] > best_v) {
      best_v = logits[i];
      best = i;
    }
  return static_cast<float>(best - 30);
}
int vp_layer_launch(const vp_engine* e, int i, const char** launch) {
  if (!e || !e->impl || !launch || i < 0 || i >= (int)e->impl->ops().size()) return VP_ERR_ARG;
  *launch = e->impl->ops()[i].launch.c_str();
  return VP_OK;
}
// FNV-1a over (launch name, kernel tag, launch geometry) of every launch of the plan: two engines with equal hashes run the same kernels with the same
// split factors / group counts in the same order
unsigned long long vp_plan_hash(const vp_engine* e) {
  if (!e || !e->impl) return 0;
  unsigned long long h = 1469598103934665603ull;
  auto mix = [&h](const std::string& t) {
    for (unsigned char c : t) {
      h ^= c;
      h *= 1099511628211ull;
    }
    h ^= 0xffu;
    h *= 1099511628211ull;
  };
  for (const vp::Op& op : e->impl->ops()) {
    mix(op.name);
    mix(op.kernel);
    if (!op.launch.empty()) mix(op.launch);
  }
  return h ? h : 1;
}
int vp_weight_bytes(const vp_engine* e, unsigned long long* fp8_bytes, unsigned long long* fp16_bytes, unsigned long long* fp32_bytes) {
  if (!e || !e->impl) return VP_ERR_ARG;
  const unsigned long long* w = e->impl->weight_bytes();
  if (fp8_bytes) *fp8_bytes = w[0];
  if (fp16_bytes) *fp16_bytes = w[1];
  if (fp32_bytes) *fp32_bytes = w[2];
  return VP_OK;
}
// host only: the (hi, lo) fp16 planes and the per-row 2^-s the engine makes of a weight matrix (engine.cpp row_prescale + split_half)
int vp_split_weight_rows(const float* w, int rows, int per_row, uint16_t* hi, uint16_t* lo, float* post_scale) {
  if (!w || rows < 1 || per_row < 1 || !hi || !lo || !post_scale) return VP_ERR_ARG;
  try {
    const vp::RowScale rs = vp::row_prescale(w, (size_t)rows, (size_t)per_row, (size_t)rows);
    for (int r = 0; r < rows; ++r) {
      post_scale[r] = rs.post[r];
      for (int i = 0; i < per_row; ++i) {
        vp::half_t h, l;
        vp::split_half(w[(size_t)r * per_row + i], rs.pre[r], &h, &l);
        std::memcpy(hi + (size_t)r * per_row + i, &h, 2);
        std::memcpy(lo + (size_t)r * per_row + i, &l, 2);
      }
    }
  } catch (const vp::RangeError&) {
    return VP_ERR_RANGE;
  }
  return VP_OK;
}
// host only: the e4m3 codes and row scales VP_WEIGHTS_FP8 storage makes of a (quantised, possibly re-scaled) weight matrix
int vp_fp8_encode_rows(const float* w, int rows, int per_row, uint8_t* codes, float* row_scale) {
  if (!w || rows < 1 || per_row < 1 || !codes || !row_scale) return VP_ERR_ARG;
  for (int r = 0; r < rows; ++r) {
    float amax = 0.0f;
    for (int i = 0; i < per_row; ++i) amax = std::max(amax, std::fabs(w[(size_t)r * per_row + i]));
    row_scale[r] = vp::fp8_row_scale(amax);
    for (int i = 0; i < per_row; ++i) codes[(size_t)r * per_row + i] = vp::e4m3_encode(w[(size_t)r * per_row + i] / row_scale[r]);
  }
  return VP_OK;
}
int vp_copy_outputs_device(vp_engine* e, void* logits_dst, void* mask_dst) {
  return guarded(e, [&](vp::Engine& g) { g.copy_outputs_device(logits_dst, mask_dst); });
}
int vp_profile_layers(vp_engine* e, int iters, float* ms, int capacity) {
  int n = 0;
  const int rc = guarded(e, [&](vp::Engine& g) { n = g.profile_layers(iters, ms, capacity); });
  return rc == VP_OK ? n : rc;
}
int vp_tensor_count(const vp_engine* e) { return (e && e->impl) ? (int)e->impl->acts().size() : VP_ERR_ARG; }
int vp_tensor_info(const vp_engine* e, int i, const char** name, int* c, int* h, int* w) {
  if (!e || !e->impl || i < 0 || i >= (int)e->impl->acts().size()) return VP_ERR_ARG;
  const vp::Act& a = *e->impl->acts()[i];
  if (name) *name = a.name.c_str();
  if (c) *c = a.Creal;
  if (h) *h = a.H;
  if (w) *w = a.W;
  return VP_OK;
}
int vp_tensor_read(vp_engine* e, int i, float* dst) {
  return guarded(e, [&](vp::Engine& g) { g.read_act(i, dst); });
}

int vp_op_conv2d(int gpu_id, int precision, int mode, const float* in, int cin, int h, int w, const float* weight, const float* bias,
                 int cout, int ks, int act, int res_mode, const float* res, int tile, int bk, int nsplit, float* out, char* err,
                 size_t err_len) {
  if (!in || !weight || !bias || !out || cin < 1 || cout < 1 || h < 1 || w < 1) {
    set_err(err, err_len, "bad argument");
    return VP_ERR_ARG;
  }
  try {
    vp::Engine g(-1, nullptr, precision, gpu_id);
    vp::Act* a = g.new_act("in", cin, h, w);
    g.upload_act(a, in);
    const int oh = mode >= 1 ? 2 * h : h, ow = mode >= 1 ? 2 * w : w;
    if (mode == 2) {  // ConvTranspose2d(k2, s2)(in) + Conv2d 1x1(skip): `res` is the skip INPUT [res_mode channels][2h][2w]
      const int cs = res_mode;
      if (!res || cs < 1) throw std::invalid_argument("mode 2: res = skip tensor, res_mode = its channel count");
      vp::Act* sk = g.new_act("skip", cs, oh, ow);
      g.upload_act(sk, res);
      const size_t wt_n = (size_t)cin * cout * 4;
      std::vector<float> wt(weight, weight + wt_n), ws(weight + wt_n, weight + wt_n + (size_t)cout * cs), bt(bias, bias + cout),
          bs(bias + cout, bias + 2 * cout);
      vp::Act* y = g.add_convT_skip("op", "op.skip", a, sk, wt, bt, ws, bs, cout);
      g.run_eager();
      g.sync();
      for (size_t i = 0; i < g.acts().size(); ++i)
        if (g.acts()[i].get() == y) g.read_act((int)i, out);
      return VP_OK;
    }
    if (mode == 3) {  // a head's logits convolution: 3x3, fp32 NCHW output straight from the kernel (STORE_NCHW_F32)
      if (ks != 3 || act != 0 || res_mode != 0) throw std::invalid_argument("mode 3: 3x3, no activation, no residual");
      float* d_out = g.alloc_f32((size_t)cout * h * w);
      vp::ConvOpts lo;
      lo.tile = tile;
      lo.logits_out = d_out;
      const size_t wn3 = (size_t)cin * cout * 9;
      g.add_conv("op", a, std::vector<float>(weight, weight + wn3), std::vector<float>(bias, bias + cout), cout, 3, lo);
      g.run_eager();
      g.sync();
      g.copy_d2h(out, d_out, (size_t)cout * h * w * sizeof(float));
      return VP_OK;
    }
    vp::ConvOpts o;
    o.act = act;
    o.tile = tile;
    o.bk = bk;
    o.nsplit = nsplit;
    vp::Act* r = nullptr;
    if (res_mode != 0) {
      if (!res) throw std::invalid_argument("res_mode set but res is null");
      r = g.new_act("res", cout, oh, ow);
      g.upload_act(r, res);
      o.res_mode = res_mode;
      o.res = r;
    }
    const size_t wn = (size_t)cin * cout * (mode == 1 ? 4 : ks * ks);
    std::vector<float> wv(weight, weight + wn), bv(bias, bias + cout);
    vp::Act* y = mode == 1 ? g.add_convT("op", a, wv, bv, cout, o) : g.add_conv("op", a, wv, bv, cout, ks, o);
    g.run_eager();
    g.sync();
    int idx = -1;
    for (size_t i = 0; i < g.acts().size(); ++i)
      if (g.acts()[i].get() == y) idx = (int)i;
    g.read_act(idx, out);
    return VP_OK;
  } catch (const std::invalid_argument& ex) {
    set_err(err, err_len, ex.what());
    return VP_ERR_ARG;
  } catch (const std::exception& ex) {
    set_err(err, err_len, ex.what());
    return VP_ERR_HIP;
  }
}

// ---- composed up-sampling stage (round 6): the load-time weight composition alone, and the whole stage as one operator
int vp_compose_upconv(int gpu_id, const float* wt, const float* bt, const float* ws, const float* bs, const float* w3, const float* b3, int cin, int cm,
                      int cout, int cs, double* wx, double* wsk, double* bias, char* err, size_t err_len) {
  if (!wt || !bt || !w3 || !b3 || !wx || !bias || cin < 1 || cm < 1 || cout < 1 || cs < 0 || (cs > 0 && (!ws || !bs || !wsk))) {
    set_err(err, err_len, "bad argument");
    return VP_ERR_ARG;
  }
  try {
    vp::Engine g(-1, nullptr, VP_FP16X3, gpu_id);
    vp::UpconvComposed c;
    g.compose_upconv(wt, bt, ws, bs, w3, b3, cin, cm, cout, cs, &c);
    std::memcpy(wx, c.wx.data(), c.wx.size() * sizeof(double));
    if (cs > 0) std::memcpy(wsk, c.ws.data(), c.ws.size() * sizeof(double));
    std::memcpy(bias, c.bias.data(), c.bias.size() * sizeof(double));
    return VP_OK;
  } catch (const std::invalid_argument& ex) {
    set_err(err, err_len, ex.what());
    return VP_ERR_ARG;
  } catch (const std::exception& ex) {
    set_err(err, err_len, ex.what());
    return VP_ERR_HIP;
  }
}

int vp_op_upconv(int gpu_id, const float* in, int cin, int h, int w, const float* skip, int cs, const float* wt, const float* bt, const float* ws,
                 const float* bs, const float* w3, const float* b3, int cm, int cout, int act, int shape, int nsplit, int precision, float* out, char* err,
                 size_t err_len) {
  if (precision != VP_FP16 && precision != VP_FP16X3) {
    set_err(err, err_len, "precision must be VP_FP16 or VP_FP16X3");
    return VP_ERR_ARG;
  }
  if (!in || !wt || !bt || !w3 || !b3 || !out || cin < 1 || cm < 1 || cout < 1 || h < 1 || w < 1 || cs < 0 || (cs > 0 && (!skip || !ws || !bs))) {
    set_err(err, err_len, "bad argument");
    return VP_ERR_ARG;
  }
  try {
    vp::Engine g(-1, nullptr, precision, gpu_id);
    vp::Act* a = g.new_act("in", cin, h, w);
    g.upload_act(a, in);
    vp::Act* sk = nullptr;
    if (cs > 0) {
      sk = g.new_act("skip", cs, 2 * h, 2 * w);
      g.upload_act(sk, skip);
    }
    const std::vector<float> vwt(wt, wt + (size_t)cin * cm * 4), vbt(bt, bt + cm), vw3(w3, w3 + (size_t)cout * cm * 9), vb3(b3, b3 + cout);
    const std::vector<float> vws(cs > 0 ? ws : nullptr, cs > 0 ? ws + (size_t)cm * cs : nullptr), vbs(cs > 0 ? bs : nullptr, cs > 0 ? bs + cm : nullptr);
    vp::Act* y = g.add_upconv("op", a, sk, vwt, vbt, vws, vbs, vw3, vb3, cm, cout, act, shape, nsplit);
    g.run_eager();
    g.sync();
    for (size_t i = 0; i < g.acts().size(); ++i)
      if (g.acts()[i].get() == y) g.read_act((int)i, out);
    return VP_OK;
  } catch (const vp::RangeError& ex) {
    set_err(err, err_len, ex.what());
    return VP_ERR_RANGE;
  } catch (const std::invalid_argument& ex) {
    set_err(err, err_len, ex.what());
    return VP_ERR_ARG;
  } catch (const std::exception& ex) {
    set_err(err, err_len, ex.what());
    return VP_ERR_HIP;
  }
}

}  // extern "C"
