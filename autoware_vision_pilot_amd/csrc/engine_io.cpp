// libvp_hip engine, part 3 of 3: the per-frame path around the plan -- input format and resize tables, frame upload, eager run / hipGraph
// capture and replay (single engine and base + shared heads), output fetch and the range probe's verdict, resizes, visualisation, timers.
#include "engine_internal.hpp"

#include <cstdint>
#include <mutex>
#include <utility>
#include <vector>

#include "../../include/vp_hip.h"

namespace vp {

// ------------------------------------------------------------------------------------------ frame handling
void Engine::set_input_format(int pixel_format, int plane_order) {
  if (pixel_format < 0 || pixel_format > 1 || plane_order < 0 || plane_order > 1) throw std::invalid_argument("bad input format");
  if (pixel_format != pixel_format_ || plane_order != plane_order_) { graph_valid_ = false; ++plan_epoch_; }
  pixel_format_ = pixel_format;
  plane_order_ = plane_order;
}
void Engine::set_norm_form(int form) {
  if (form < 0 || form > 1) throw std::invalid_argument("norm form: 0 = q / 255 (torchvision to_tensor), 1 = q * fl(1/255) (cv::Mat::convertTo)");
  if (base_) throw std::invalid_argument("shared engine: the base engine owns the frame path");
  if (form != norm_form_) { graph_valid_ = false; ++plan_epoch_; }
  norm_form_ = form;
}
void Engine::set_lane_ring(bool on) {
  if (kind_ != 3) throw std::invalid_argument("lane ring: VP_EGOLANES engines only (the AutoSteer hand-over of the EgoLanes logits)");
  if (on == lane_ring_) return;
  VP_HIP_CHECK(hipSetDevice(gpu_));
  VP_HIP_CHECK(hipStreamSynchronize(stream_));
  if (on && !d_lane_ring_) d_lane_ring_ = static_cast<float*>(dalloc((size_t)2 * out_c_ * out_h_ * out_w_ * sizeof(float), true));
  if (on) fill_zero(d_lane_ring_, (size_t)2 * out_c_ * out_h_ * out_w_ * sizeof(float));
  lane_ring_ = on;
  ring_frames_ = 0;
  graph_valid_ = false;
  ++plan_epoch_;
}
void Engine::fetch_lane_ring(float* dst) {
  if (!lane_ring_ || !d_lane_ring_) throw std::runtime_error("lane ring is not enabled (vp_set_lane_ring)");
  if (!dst) throw std::invalid_argument("null destination");
  VP_HIP_CHECK(hipSetDevice(gpu_));
  VP_HIP_CHECK(hipMemcpyAsync(dst, d_lane_ring_, (size_t)2 * out_c_ * out_h_ * out_w_ * sizeof(float), hipMemcpyDeviceToHost, stream_));
  VP_HIP_CHECK(hipStreamSynchronize(stream_));
}
void Engine::set_decode_mode(int mode) {
  if (mode < 0 || mode > 2) throw std::invalid_argument("bad decode mode");
  if (mode != decode_mode_) { graph_valid_ = false; ++plan_epoch_; }
  decode_mode_ = mode;
}

// 11-bit fixed-point bilinear taps -- must stay bit-identical to oracle/pre_post.py linear_taps_u8.
void linear_taps_u8(int src, int dst, std::vector<int>* tab) {
  tab->resize((size_t)dst * 4);
  const double scale = (double)src / (double)dst;
  for (int d = 0; d < dst; ++d) {
    float f = (float)(((double)d + 0.5) * scale - 0.5);
    int s = (int)std::floor(f);
    f -= (float)s;
    if (s < 0) {
      f = 0.0f;
      s = 0;
    }
    if (s >= src - 1) {
      f = 0.0f;
      s = src - 1;
    }
    (*tab)[4 * d + 0] = s;
    (*tab)[4 * d + 1] = std::min(s + 1, src - 1);
    (*tab)[4 * d + 2] = (int)std::nearbyint((1.0f - f) * 2048.0f);
    (*tab)[4 * d + 3] = (int)std::nearbyint(f * 2048.0f);
  }
}

// Pillow's precompute_coeffs + normalize_coeffs_8bpc (src/libImaging/Resample.c) in the same double arithmetic, operation for
// operation (restated and pinned against PIL in oracle/pre_post.py pil_resample_coeffs): filter support scaled by the
// down-scaling factor, taps normalised to sum 1, quantised to 22 fractional bits.  filter: 1 = BILINEAR, 2 = BICUBIC (a = -0.5).
#pragma clang fp contract(off)
int pil_coeffs(int in_size, int out_size, int filter, std::vector<int>* bounds, std::vector<int>* kk) {
  auto weight = [filter](double x) -> double {
    if (x < 0.0) x = -x;
    if (filter == 1) return x < 1.0 ? 1.0 - x : 0.0;
    const double a = -0.5;
    if (x < 1.0) return ((a + 2.0) * x - (a + 3.0)) * x * x + 1;
    if (x < 2.0) return (((x - 5) * x + 8) * x - 4) * a;
    return 0.0;
  };
  double filterscale = (double)in_size / out_size;
  const double scale = filterscale;
  if (filterscale < 1.0) filterscale = 1.0;
  const double support = (filter == 1 ? 1.0 : 2.0) * filterscale;
  const int ksize = (int)std::ceil(support) * 2 + 1;
  bounds->assign((size_t)out_size * 2, 0);
  kk->assign((size_t)out_size * ksize, 0);
  std::vector<double> k(ksize);
  const double ss = 1.0 / filterscale;
  for (int xx = 0; xx < out_size; ++xx) {
    const double center = (xx + 0.5) * scale;
    double ww = 0.0;
    int xmin = (int)(center - support + 0.5);
    if (xmin < 0) xmin = 0;
    int xmax = (int)(center + support + 0.5);
    if (xmax > in_size) xmax = in_size;
    xmax -= xmin;
    for (int x = 0; x < xmax; ++x) {
      const double w = weight((x + xmin - center + 0.5) * ss);
      k[x] = w;
      ww += w;
    }
    for (int x = 0; x < xmax; ++x) {
      if (ww != 0.0) k[x] /= ww;
      (*kk)[(size_t)xx * ksize + x] = k[x] < 0 ? (int)(-0.5 + k[x] * (1 << 22)) : (int)(0.5 + k[x] * (1 << 22));
    }
    (*bounds)[2 * xx] = xmin;
    (*bounds)[2 * xx + 1] = xmax;
  }
  return ksize;
}

PilResampleParams Engine::pil_params(const PreprocessParams& pp) const {
  PilResampleParams q{};
  q.frame = pp.frame;
  q.stride = pp.stride;
  q.in_h = frame_h_;
  q.in_w = frame_w_;
  q.out_h = pp.out_h;
  q.out_w = pp.out_w;
  q.hb = d_pil_hb_;
  q.hk = d_pil_hk_;
  q.hks = pil_hks_;
  q.vb = d_pil_vb_;
  q.vk = d_pil_vk_;
  q.vks = pil_vks_;
  q.tmp = d_pil_tmp_;
  for (int c = 0; c < 3; ++c) {
    q.src_c[c] = pp.src_c[c];
    q.mean[c] = pp.mean[c];
    q.stdv[c] = pp.stdv[c];
  }
  q.out = pp.out;
  q.norm_form = pp.norm_form;
  return q;
}

void Engine::set_resize_mode(int mode) {
  if (mode < 0 || mode > 2) throw std::invalid_argument("resize mode: 0 = cv::resize INTER_LINEAR model, 1 = PIL BILINEAR, 2 = PIL BICUBIC");
  if (base_) throw std::invalid_argument("shared engine: the base engine owns the frame path");
  if (mode != resize_mode_) {
    resize_mode_ = mode;
    tab_h_ = tab_w_ = 0;
    graph_valid_ = false;
    ++plan_epoch_;
  }
}

void Engine::ensure_tables(int h, int w) {
  if (h == tab_h_ && w == tab_w_) return;
  if (resize_mode_ != 0) {  // Pillow's resample: per-output tap tables for both passes + the u8 image between them
    std::vector<int> hb, hk, vb, vk;
    pil_hks_ = pil_coeffs(w, net_w(), resize_mode_, &hb, &hk);
    pil_vks_ = pil_coeffs(h, net_h(), resize_mode_, &vb, &vk);
    VP_HIP_CHECK(hipStreamSynchronize(stream_));
    // capacity-tracked: a host alternating between frame geometries re-uses (or re-grows and FREES) these five buffers instead of
    // leaking a set per change
    upload_grow(d_pil_hb_, pil_cap_[0], hb);
    upload_grow(d_pil_hk_, pil_cap_[1], hk);
    upload_grow(d_pil_vb_, pil_cap_[2], vb);
    upload_grow(d_pil_vk_, pil_cap_[3], vk);
    const size_t tmp_need = (size_t)h * net_w() * 3;
    if (tmp_need > pil_cap_[4]) {
      dfree(d_pil_tmp_);
      d_pil_tmp_ = static_cast<uint8_t*>(dalloc(tmp_need, false));
      pil_cap_[4] = tmp_need;
    }
    tab_h_ = h;
    tab_w_ = w;
    return;
  }
  std::vector<int> xt, yt;
  linear_taps_u8(w, net_w(), &xt);
  linear_taps_u8(h, net_h(), &yt);
  if (!d_xtab_) {
    d_xtab_ = static_cast<int*>(dalloc(net_w() * 4 * sizeof(int)));
    d_ytab_ = static_cast<int*>(dalloc(net_h() * 4 * sizeof(int)));
  }
  VP_HIP_CHECK(hipStreamSynchronize(stream_));
  copy_h2d(d_xtab_, xt.data(), xt.size() * sizeof(int));
  copy_h2d(d_ytab_, yt.data(), yt.size() * sizeof(int));
  tab_h_ = h;
  tab_w_ = w;
}

// ---- frame pools registered for DMA (vp_register_frames): a short sorted table behind a mutex; the lookup is two compares per range
namespace {
std::mutex g_pool_mu;
std::vector<std::pair<uintptr_t, size_t>> g_pools;   // [begin, bytes), disjoint
}  // namespace
int register_frame_range(const void* pool, size_t bytes) {
  if (!pool || bytes == 0) return VP_ERR_ARG;
  const uintptr_t b = reinterpret_cast<uintptr_t>(pool);
  std::lock_guard<std::mutex> lk(g_pool_mu);
  for (const auto& r : g_pools)
    if (b < r.first + r.second && r.first < b + bytes) return VP_ERR_ARG;   // overlaps a registered range
  if (hipHostRegister(const_cast<void*>(pool), bytes, hipHostRegisterPortable) != hipSuccess) {
    (void)hipGetLastError();
    return VP_ERR_HIP;
  }
  g_pools.emplace_back(b, bytes);
  return VP_OK;
}
int unregister_frame_range(const void* pool) {
  if (!pool) return VP_ERR_ARG;
  const uintptr_t b = reinterpret_cast<uintptr_t>(pool);
  std::lock_guard<std::mutex> lk(g_pool_mu);
  for (size_t i = 0; i < g_pools.size(); ++i)
    if (g_pools[i].first == b) {
      // unlock first, forget only on success: a refused unlock (a DMA still in flight) leaves the range page-locked AND tracked, so frames from it keep
      // the one-DMA path and a later retry finds it (erased first, a failure left it locked but unknown)
      if (hipHostUnregister(const_cast<void*>(pool)) != hipSuccess) {
        (void)hipGetLastError();
        return VP_ERR_HIP;
      }
      g_pools.erase(g_pools.begin() + i);
      return VP_OK;
    }
  return VP_ERR_ARG;
}
bool frame_range_registered(const void* p, size_t bytes) {
  const uintptr_t b = reinterpret_cast<uintptr_t>(p);
  std::lock_guard<std::mutex> lk(g_pool_mu);
  for (const auto& r : g_pools)
    if (b >= r.first && b + bytes <= r.first + r.second) return true;
  return false;
}

void Engine::upload_frame(const uint8_t* frame, int h, int w, int stride, int index) {
  if (base_) throw std::invalid_argument("shared engine: frames go to the base engine (vp_infer on the base, then vp_infer_shared)");
  if (!frame || h < 2 || w < 2 || stride < 3 * w) throw std::invalid_argument("bad frame geometry");
  if (index < 0 || index >= frames_) throw std::invalid_argument("frame index out of range");
  VP_HIP_CHECK(hipSetDevice(gpu_));
  const size_t need = (size_t)h * stride;
  if ((h != frame_h_ || w != frame_w_ || stride != frame_stride_) && frames_ > 1 && index != 0 && frame_h_ != 0)
    throw std::invalid_argument("batched encoder: all frames of a pass share one geometry (upload slot 0 first to change it)");
  if (need * frames_ > frame_cap_) {
    VP_HIP_CHECK(hipStreamSynchronize(stream_));
    dfree(d_frame_);
    d_frame_ = static_cast<uint8_t*>(dalloc(need * frames_, true));
    frame_cap_ = need * frames_;
    { graph_valid_ = false; ++plan_epoch_; }
  }
  if (h != frame_h_ || w != frame_w_ || stride != frame_stride_) { graph_valid_ = false; ++plan_epoch_; }
  ensure_tables(h, w);
  frame_h_ = h;
  frame_w_ = w;
  frame_stride_ = stride;
  // A strided view (cv::Mat ROI, numpy slice) guarantees only (h-1)*stride + 3*w readable bytes: the tail of the last row
  // belongs to the parent image or to nobody.  Packed frames go as one copy, views row by row (hipMemcpy2D).
  // The caller's buffer is pageable (cv::Mat); the copy is staged through this engine's pinned buffer so the transfer
  // itself is one DMA that overlaps other engines' kernels (the reference does the same H2D: tensorrt_backend.cpp:184-186).
  uint8_t* dst = d_frame_ + (size_t)index * need;
  const size_t packed = (size_t)(h - 1) * stride + (size_t)3 * w;
  // a frame inside a pool the host registered (vp_register_frames) is page-locked already: one DMA straight from the caller's memory
  const bool staged = pinned_staging_ && !frame_range_registered(frame, packed);
  if (staged) {
    if (need > h_frame_cap_) {
      VP_HIP_CHECK(hipStreamSynchronize(stream_));
      if (h_frame_) hipHostFree(h_frame_);
      h_frame_ = nullptr;
      VP_HIP_CHECK(hipHostMalloc(reinterpret_cast<void**>(&h_frame_), need * 2, hipHostMallocDefault));  // two slots: frame n+1 is staged while n flies
      h_frame_cap_ = need;
    }
    h_frame_slot_ ^= 1;
    uint8_t* slot = h_frame_ + (size_t)h_frame_slot_ * h_frame_cap_;
    // A pinned source makes the H2D below truly asynchronous: the copy that last read this slot (two uploads ago) may still be
    // queued behind earlier frames' graphs.  Its event orders this host write behind it.
    if (!h_frame_ev_[h_frame_slot_]) VP_HIP_CHECK(hipEventCreateWithFlags(&h_frame_ev_[h_frame_slot_], hipEventDisableTiming));
    else VP_HIP_CHECK(hipEventSynchronize(h_frame_ev_[h_frame_slot_]));
    if (stride == 3 * w) {
      std::memcpy(slot, frame, packed);
    } else {
      for (int y = 0; y < h; ++y) std::memcpy(slot + (size_t)y * stride, frame + (size_t)y * stride, (size_t)3 * w);
    }
    VP_HIP_CHECK(hipMemcpyAsync(dst, slot, packed, hipMemcpyHostToDevice, stream_));
    VP_HIP_CHECK(hipEventRecord(h_frame_ev_[h_frame_slot_], stream_));
  } else if (stride == 3 * w) {
    VP_HIP_CHECK(hipMemcpyAsync(dst, frame, packed, hipMemcpyHostToDevice, stream_));
  } else {
    VP_HIP_CHECK(hipMemcpy2DAsync(dst, stride, frame, stride, (size_t)3 * w, h, hipMemcpyHostToDevice, stream_));
  }
  if (input_is_tensor_) { graph_valid_ = false; ++plan_epoch_; }
  input_is_tensor_ = false;
}

void Engine::upload_tensor(const float* nchw) {
  if (base_) throw std::invalid_argument("shared engine: tensors go to the base engine");
  if (frames_ > 1) throw std::invalid_argument("batched encoder: frames only (vp_upload_frame_n)");
  if (!nchw) throw std::invalid_argument("null tensor");
  VP_HIP_CHECK(hipSetDevice(gpu_));
  VP_HIP_CHECK(hipMemcpyAsync(d_input_, nchw, (size_t)3 * net_h() * net_w() * sizeof(float), hipMemcpyHostToDevice, stream_));
  if (!input_is_tensor_) { graph_valid_ = false; ++plan_epoch_; }
  input_is_tensor_ = true;
}

void Engine::run_ops(hipStream_t st, size_t begin, size_t end) {
  for (size_t i = begin; i < end; ++i) {
    hipError_t e = ops_[i].run(st);
    if (e != hipSuccess) throw std::runtime_error("launch failed in layer '" + ops_[i].name + "': " + hipGetErrorString(e));
  }
}
void Engine::run_eager() { run_ops(stream_, input_is_tensor_ ? first_net_op_ : 0, ops_.size()); }

// One frame through this engine AND its shared-prefix heads as ONE graph launch on this engine's stream (so every stream-order
// guarantee of the separate vp_enqueue calls holds).  Inside the graph the heads that consume only the backbone (shared level 1:
// Scene3D, EgoLanes on a SceneSeg base) are forked onto side streams right behind the backbone and joined at the end: a single
// frame's two or three decoders overlap (the small-map neck layers and the 200-tile big layers leave CUs idle on their own).
// Same kernels, same arguments, same results as base.enqueue() followed by head.enqueue().
void Engine::enqueue_multi(const std::vector<Engine*>& heads) {
  VP_HIP_CHECK(hipSetDevice(gpu_));
  if (base_) throw std::invalid_argument("enqueue_multi: call it on the engine that owns the encoder");
  for (Engine* h : heads)
    if (!h || h->base_ != this || h->stream_ != stream_) throw std::invalid_argument("enqueue_multi: every head must be a shared-prefix engine of this engine");
  bool plain = !multi_fork_ || !use_graph_ || !warmed_ || kind_ == 4 || frames_ > 1 || n_fork_ops_ == 0 || heads.empty();
  for (Engine* h : heads) plain = plain || !h->warmed_ || !h->use_graph_;
  if (plain) {  // first frames (eager warm-up), graph replay switched off, or nothing to fork
    enqueue();
    for (Engine* h : heads) h->enqueue();
    return;
  }
  if (!input_is_tensor_ && !d_frame_) throw std::runtime_error("no frame resident: call vp_upload_frame / vp_infer first");
  std::vector<std::pair<const Engine*, unsigned long long>> key{{this, plan_epoch_}};
  for (Engine* h : heads) key.emplace_back(h, h->plan_epoch_);
  if (!multi_exec_ || key != multi_key_) {
    if (multi_exec_) hipGraphExecDestroy(multi_exec_);
    if (multi_graph_) hipGraphDestroy(multi_graph_);
    multi_exec_ = nullptr;
    multi_graph_ = nullptr;
    size_t n_side = 0;
    for (Engine* h : heads) n_side += h->shared_level_ == 1 ? 1 : 0;
    while (side_streams_.size() < n_side) {
      hipStream_t s = nullptr;
      VP_HIP_CHECK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
      side_streams_.push_back(s);
    }
    while (side_events_.size() < n_side + 1) {
      hipEvent_t e = nullptr;
      VP_HIP_CHECK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
      side_events_.push_back(e);
    }
    VP_HIP_CHECK(hipStreamBeginCapture(stream_, hipStreamCaptureModeThreadLocal));
    try {
      const size_t first = input_is_tensor_ ? first_net_op_ : 0;
      run_ops(stream_, first, n_fork_ops_);
      VP_HIP_CHECK(hipEventRecord(side_events_[0], stream_));
      size_t si = 0;
      for (Engine* h : heads)
        if (h->shared_level_ == 1) {
          VP_HIP_CHECK(hipStreamWaitEvent(side_streams_[si], side_events_[0], 0));
          h->run_ops(side_streams_[si], 0, h->ops_.size());
          VP_HIP_CHECK(hipEventRecord(side_events_[1 + si], side_streams_[si]));
          ++si;
        }
      run_ops(stream_, n_fork_ops_, ops_.size());
      for (Engine* h : heads)
        if (h->shared_level_ != 1) h->run_ops(stream_, 0, h->ops_.size());  // needs this engine's context + neck: behind them, in order
      for (size_t i = 0; i < si; ++i) VP_HIP_CHECK(hipStreamWaitEvent(stream_, side_events_[1 + i], 0));
    } catch (...) {
      hipGraph_t g = nullptr;
      hipStreamEndCapture(stream_, &g);
      if (g) hipGraphDestroy(g);
      throw;
    }
    VP_HIP_CHECK(hipStreamEndCapture(stream_, &multi_graph_));
    VP_HIP_CHECK(hipGraphInstantiate(&multi_exec_, multi_graph_, nullptr, nullptr, 0));
    multi_key_ = key;
  }
  VP_HIP_CHECK(hipGraphLaunch(multi_exec_, stream_));
  have_outputs_ = true;
  host_logits_valid_ = host_mask_valid_ = false;
  note_pass();
  for (Engine* h : heads) {
    h->have_outputs_ = true;
    h->host_logits_valid_ = h->host_mask_valid_ = false;
    h->note_pass();
  }
}

void Engine::capture_graph() {
  if (graph_exec_) {
    hipGraphExecDestroy(graph_exec_);
    graph_exec_ = nullptr;
  }
  if (graph_) {
    hipGraphDestroy(graph_);
    graph_ = nullptr;
  }
  VP_HIP_CHECK(hipStreamBeginCapture(stream_, hipStreamCaptureModeThreadLocal));
  try {
    run_eager();
  } catch (...) {
    hipGraph_t g = nullptr;
    hipStreamEndCapture(stream_, &g);
    if (g) hipGraphDestroy(g);
    throw;
  }
  VP_HIP_CHECK(hipStreamEndCapture(stream_, &graph_));
  VP_HIP_CHECK(hipGraphInstantiate(&graph_exec_, graph_, nullptr, nullptr, 0));
  graph_valid_ = true;
}

void Engine::enqueue() {
  VP_HIP_CHECK(hipSetDevice(gpu_));
  if (base_) {
    if (!base_->have_outputs_) throw std::runtime_error("shared engine: run the base engine on a frame first");
  } else if (!input_is_tensor_ && !d_frame_) {
    throw std::runtime_error("no frame resident: call vp_upload_frame / vp_infer first");
  }
  if (kind_ == 4 && !ad_primed_) prime_previous();  // first frame of a stream: previous := current
  if (!warmed_) {  // first pass is eager: sets kernel attributes and surfaces launch errors with layer names
    run_eager();
    VP_HIP_CHECK(hipStreamSynchronize(stream_));
    warmed_ = true;
    have_outputs_ = true;
    host_logits_valid_ = host_mask_valid_ = false;
    if (!use_graph_ || kind_ == 4 || lane_ring_) {  // AutoDrive (feature shift) and the lane ring carry state: a frame must run exactly once
      note_pass();
      return;
    }
  }
  if (use_graph_) {
    if (!graph_valid_) capture_graph();
    VP_HIP_CHECK(hipGraphLaunch(graph_exec_, stream_));
  } else {
    run_eager();
  }
  have_outputs_ = true;
  host_logits_valid_ = host_mask_valid_ = false;
  note_pass();
}

void Engine::sync() {
  VP_HIP_CHECK(hipStreamSynchronize(stream_));
  check_status();
}

// The probe's verdict on the pass whose outputs were last fetched (enqueue_fetch copies the flag behind them).  The flag is PER PASS: the
// probe kernel overwrites every word of it on every pass (kernels_misc.hip; nothing to clear, nothing sticky), so a bad frame is reported once, by the call that fetches THAT frame, and a
// pass that was enqueued but never fetched leaves nothing behind for a later frame.
bool Engine::poll_status() {
  if (!status_pending_ || !h_status_) return false;
  status_pending_ = false;
  unsigned bad = 0;
  for (int i = 0; i < VP_PROBE_BLOCKS; ++i) bad |= h_status_[i];
  std::memset(h_status_, 0, VP_PROBE_BLOCKS * sizeof(unsigned));
  return bad != 0;
}
void Engine::check_status() {
  if (poll_status())
    throw RangeError("non-finite value (inf / NaN) in the network output: an activation left the fp16 range of the matrix pipe (|x| > 65504) "
                     "or the input / weights were not finite; outputs of this frame are invalid");
}

void Engine::fetch_outputs() {
  if (!d_logits_) throw std::runtime_error("this engine has no outputs (batched encoder: fetch from its shared-prefix engines)");
  enqueue_fetch();
  VP_HIP_CHECK(hipStreamSynchronize(stream_));
  check_status();
}

// D2H of the outputs the caller selected (vp_set_outputs), asynchronous on the engine stream, into pinned host memory.
void Engine::enqueue_fetch() {
  if (!d_logits_) throw std::runtime_error("this engine has no outputs (batched encoder: fetch from its shared-prefix engines)");
  if (outputs_ & 1)
    VP_HIP_CHECK(hipMemcpyAsync(h_logits_, d_logits_, (size_t)out_c_ * out_h_ * out_w_ * sizeof(float), hipMemcpyDeviceToHost, stream_));
  if ((outputs_ & 2) && d_mask_) VP_HIP_CHECK(hipMemcpyAsync(h_mask_, d_mask_, (size_t)out_h_ * out_w_, hipMemcpyDeviceToHost, stream_));
  host_logits_valid_ = (outputs_ & 1) != 0;
  host_mask_valid_ = (outputs_ & 2) != 0;
  if (finite_check_ && d_status_) {
    VP_HIP_CHECK(hipMemcpyAsync(h_status_, d_status_, VP_PROBE_BLOCKS * sizeof(unsigned), hipMemcpyDeviceToHost, stream_));
    status_pending_ = true;
  }
}

// Lazy variants behind vp_logits / vp_mask_u8: an output de-selected with vp_set_outputs is fetched on first use.
const float* Engine::host_logits() {
  if (!host_logits_valid_ && d_logits_ && have_outputs_) {
    VP_HIP_CHECK(hipSetDevice(gpu_));
    VP_HIP_CHECK(hipMemcpyAsync(h_logits_, d_logits_, (size_t)out_c_ * out_h_ * out_w_ * sizeof(float), hipMemcpyDeviceToHost, stream_));
    VP_HIP_CHECK(hipStreamSynchronize(stream_));
    host_logits_valid_ = true;
  }
  return h_logits_;
}
const uint8_t* Engine::host_mask() {
  if (!host_mask_valid_ && d_mask_ && have_outputs_) {
    VP_HIP_CHECK(hipSetDevice(gpu_));
    VP_HIP_CHECK(hipMemcpyAsync(h_mask_, d_mask_, (size_t)out_h_ * out_w_, hipMemcpyDeviceToHost, stream_));
    VP_HIP_CHECK(hipStreamSynchronize(stream_));
    host_mask_valid_ = true;
  }
  return h_mask_;
}

void Engine::copy_outputs_device(void* logits_dst, void* mask_dst) {
  if (logits_dst)
    VP_HIP_CHECK(hipMemcpyAsync(logits_dst, d_logits_, (size_t)out_c_ * out_h_ * out_w_ * sizeof(float), hipMemcpyDeviceToDevice, stream_));
  if (mask_dst) VP_HIP_CHECK(hipMemcpyAsync(mask_dst, d_mask_, (size_t)out_h_ * out_w_, hipMemcpyDeviceToDevice, stream_));
}

void Engine::read_input_tensor(float* dst) {
  VP_HIP_CHECK(hipStreamSynchronize(stream_));
  copy_d2h(dst, d_input_, (size_t)3 * net_h() * net_w() * sizeof(float));
}

// OpenCV resizeNN index table (oracle/pre_post.py nearest_index)
static void nearest_tab(int src, int dst, int* tab) {
  const double inv = (double)dst / (double)src;
  const double ifx = 1.0 / inv;
  for (int d = 0; d < dst; ++d) tab[d] = std::min((int)std::floor(d * ifx), src - 1);
}
static void linear_taps_f32(int src, int dst, int* idx, float* wgt) {
  const double scale = (double)src / (double)dst;
  for (int d = 0; d < dst; ++d) {
    float f = (float)(((double)d + 0.5) * scale - 0.5);
    int s = (int)std::floor(f);
    f -= (float)s;
    if (s < 0) {
      f = 0.0f;
      s = 0;
    }
    if (s >= src - 1) {
      f = 0.0f;
      s = src - 1;
    }
    idx[2 * d] = s;
    idx[2 * d + 1] = std::min(s + 1, src - 1);
    wgt[2 * d] = 1.0f - f;
    wgt[2 * d + 1] = f;
  }
}

void Engine::mask_resized(uint8_t* dst, int h, int w) {
  if (!have_outputs_) throw std::runtime_error("Inference has not been run yet");
  if (!dst || h < 1 || w < 1) throw std::invalid_argument("bad resize target");
  const size_t need = (size_t)h * w, tabn = (size_t)(h + w);
  if (need > resize_cap_) {
    dfree(d_resize_out_);   // only this family of calls uses it, and each returns synchronised
    d_resize_out_ = dalloc(std::max(need, (size_t)4 * h * w), false);
    resize_cap_ = std::max(need, (size_t)4 * h * w);
  }
  if (tabn * 4 > rs_tab_cap_) {
    dfree(d_rs_tab_);
    d_rs_tab_ = static_cast<int*>(dalloc(tabn * 4 * sizeof(int), false));
    rs_tab_cap_ = tabn * 4;
  }
  std::vector<int> tab(h + w);
  nearest_tab(out_h_, h, tab.data());
  nearest_tab(out_w_, w, tab.data() + h);
  VP_HIP_CHECK(hipMemcpyAsync(d_rs_tab_, tab.data(), tab.size() * sizeof(int), hipMemcpyHostToDevice, stream_));
  VP_HIP_CHECK(launch_resize_nearest(d_mask_, out_w_, d_rs_tab_, d_rs_tab_ + h, h, w, static_cast<uint8_t*>(d_resize_out_), stream_));
  VP_HIP_CHECK(hipMemcpyAsync(dst, d_resize_out_, need, hipMemcpyDeviceToHost, stream_));
  VP_HIP_CHECK(hipStreamSynchronize(stream_));
}

// MasksVisualizationEngine::visualize on the device: the mask of the LAST inference, coloured, nearest-resized to the frame
// that produced it and blended 50/50 with that (still resident) frame; BGR8 out, frame size.
void Engine::visualize_mask(int viz_type, uint8_t* dst, int dst_h, int dst_w) {
  if (!have_outputs_) throw std::runtime_error("Inference has not been run yet");
  if (!dst || viz_type < 0 || viz_type > 2) throw std::invalid_argument("bad visualisation request");
  if (!d_frame_ || input_is_tensor_ || base_) throw std::runtime_error("visualize_mask needs the frame path (vp_infer) on a base engine");
  const int h = frame_h_, w = frame_w_;
  // The blend writes h*w*3 bytes: the caller's buffer must have the geometry of the frame that was inferred last (the
  // reference takes the size from original_image itself, masks_visualization_engine.cpp:19-27, so it cannot mismatch).
  if (dst_h != h || dst_w != w)
    throw std::invalid_argument("visualize_mask: destination is " + std::to_string(dst_w) + "x" + std::to_string(dst_h) +
                                " but the last inferred frame was " + std::to_string(w) + "x" + std::to_string(h));
  if (!d_viz_lut_) {
    // createColorMask (masks_visualization_engine.cpp:41-58), BGR
    std::vector<uint8_t> lut(3 * 256 * 3, 0);
    for (int v = 1; v < 256; ++v) { lut[(0 * 256 + v) * 3 + 2] = 255; }                          // "scene": 1..255 -> (0,0,255)
    const uint8_t dom0[3] = {255, 93, 61}, dom255[3] = {145, 28, 255};                            // "domain"
    for (int c = 0; c < 3; ++c) { lut[(1 * 256 + 0) * 3 + c] = dom0[c]; lut[(1 * 256 + 255) * 3 + c] = dom255[c]; }
    const uint8_t ego[3][3] = {{255, 0, 0}, {255, 0, 200}, {0, 153, 0}};                          // "egolanes": labels 0,1,2
    for (int v = 0; v < 3; ++v)
      for (int c = 0; c < 3; ++c) lut[(2 * 256 + v) * 3 + c] = ego[v][c];
    d_viz_lut_ = dupload(lut);
  }
  const size_t need = (size_t)3 * h * w, tabn = (size_t)(h + w);
  if (need > resize_cap_) {
    dfree(d_resize_out_);   // only this family of calls uses it, and each returns synchronised
    d_resize_out_ = dalloc(std::max(need, (size_t)4 * h * w), false);
    resize_cap_ = std::max(need, (size_t)4 * h * w);
  }
  if (tabn * 4 > rs_tab_cap_) {
    dfree(d_rs_tab_);
    d_rs_tab_ = static_cast<int*>(dalloc(tabn * 4 * sizeof(int), false));
    rs_tab_cap_ = tabn * 4;
  }
  std::vector<int> tab(h + w);
  nearest_tab(out_h_, h, tab.data());
  nearest_tab(out_w_, w, tab.data() + h);
  VP_HIP_CHECK(hipMemcpyAsync(d_rs_tab_, tab.data(), tab.size() * sizeof(int), hipMemcpyHostToDevice, stream_));
  VP_HIP_CHECK(launch_viz_blend(d_mask_, out_w_, d_rs_tab_, d_rs_tab_ + h, d_frame_, frame_stride_, h, w, d_viz_lut_ + (size_t)viz_type * 768,
                                pixel_format_ == 1, static_cast<uint8_t*>(d_resize_out_), stream_));
  VP_HIP_CHECK(hipMemcpyAsync(dst, d_resize_out_, need, hipMemcpyDeviceToHost, stream_));
  VP_HIP_CHECK(hipStreamSynchronize(stream_));
}

void Engine::resize_depth_on_device(int h, int w) {
  const size_t need = (size_t)4 * h * w, tabn = (size_t)4 * (h + w);
  if (need > resize_cap_) {
    dfree(d_resize_out_);   // only this family of calls uses it, and each returns synchronised
    d_resize_out_ = dalloc(need, false);
    resize_cap_ = need;
  }
  if (tabn > rs_tab_cap_) {
    dfree(d_rs_tab_);
    d_rs_tab_ = static_cast<int*>(dalloc(tabn * sizeof(int), false));
    rs_tab_cap_ = tabn;
  }
  // host tap tables are members: they must outlive the asynchronous copies (every caller syncs the stream before returning)
  std::vector<int>& idx = rs_idx_host_;
  std::vector<float>& wgt = rs_wgt_host_;
  idx.assign(2 * (h + w), 0);
  wgt.assign(2 * (h + w), 0.f);
  linear_taps_f32(out_h_, h, idx.data(), wgt.data());
  linear_taps_f32(out_w_, w, idx.data() + 2 * h, wgt.data() + 2 * h);
  int* d_idx = d_rs_tab_;
  float* d_wgt = reinterpret_cast<float*>(d_rs_tab_ + 2 * (h + w));
  VP_HIP_CHECK(hipMemcpyAsync(d_idx, idx.data(), idx.size() * sizeof(int), hipMemcpyHostToDevice, stream_));
  VP_HIP_CHECK(hipMemcpyAsync(d_wgt, wgt.data(), wgt.size() * sizeof(float), hipMemcpyHostToDevice, stream_));
  VP_HIP_CHECK(launch_resize_bilinear_f32(d_logits_, out_w_, d_idx, d_wgt, d_idx + 2 * h, d_wgt + 2 * h, h, w,
                                          static_cast<float*>(d_resize_out_), stream_));
}

void Engine::depth_resized(float* dst, int h, int w) {
  if (!have_outputs_) throw std::runtime_error("Inference has not been run yet");
  if (!dst || h < 1 || w < 1) throw std::invalid_argument("bad resize target");
  resize_depth_on_device(h, w);
  VP_HIP_CHECK(hipMemcpyAsync(dst, d_resize_out_, (size_t)4 * h * w, hipMemcpyDeviceToHost, stream_));
  VP_HIP_CHECK(hipStreamSynchronize(stream_));
}

// DepthVisualizationEngine::visualize (depth_visualization_engine.cpp:9-26) on the device: plane 0 of the logits,
// bilinear-resized to h x w (what the depth topic carries, run_model_node.cpp:100-104), min-max normalised to u8 and
// mapped through COLORMAP_VIRIDIS; BGR8 out.
void Engine::visualize_depth(uint8_t* dst, int h, int w) {
  if (!have_outputs_) throw std::runtime_error("Inference has not been run yet");
  if (!dst || h < 1 || w < 1) throw std::invalid_argument("bad visualisation target");
  static const uint8_t kViridisBgr[256 * 3] = {
#include "viridis_lut.inc"
  };
  if (!d_viridis_) {
    d_viridis_ = dupload(std::vector<uint8_t>(kViridisBgr, kViridisBgr + sizeof(kViridisBgr)));
    d_minmax_ = static_cast<unsigned*>(dalloc(2 * sizeof(unsigned), false));
  }
  const size_t n = (size_t)h * w;
  if (3 * n > depth_viz_cap_) {
    dfree(d_depth_viz_);
    d_depth_viz_ = dalloc(3 * n, false);
    depth_viz_cap_ = 3 * n;
  }
  resize_depth_on_device(h, w);
  static const unsigned kInit[2] = {0xFFFFFFFFu, 0u};
  VP_HIP_CHECK(hipMemcpyAsync(d_minmax_, kInit, sizeof(kInit), hipMemcpyHostToDevice, stream_));
  VP_HIP_CHECK(launch_minmax_f32(static_cast<const float*>(d_resize_out_), n, d_minmax_, stream_));
  VP_HIP_CHECK(launch_depth_colorize(static_cast<const float*>(d_resize_out_), n, d_minmax_, d_viridis_, static_cast<uint8_t*>(d_depth_viz_), stream_));
  VP_HIP_CHECK(hipMemcpyAsync(dst, d_depth_viz_, 3 * n, hipMemcpyDeviceToHost, stream_));
  VP_HIP_CHECK(hipStreamSynchronize(stream_));
}

// -------------------------------------------------------------------------------------------------- timing
void Engine::timer_begin() { VP_HIP_CHECK(hipEventRecord(ev0_, stream_)); }
float Engine::timer_end() {
  VP_HIP_CHECK(hipEventRecord(ev1_, stream_));
  VP_HIP_CHECK(hipEventSynchronize(ev1_));
  float ms = 0.f;
  VP_HIP_CHECK(hipEventElapsedTime(&ms, ev0_, ev1_));
  return ms;
}

int Engine::profile_layers(int iters, float* ms, int cap) {
  const size_t first = input_is_tensor_ ? first_net_op_ : 0;
  const int n = (int)ops_.size();
  if (cap < n) throw std::invalid_argument("profile buffer too small");
  if (base_) {
    if (!base_->have_outputs_) throw std::runtime_error("shared engine: run the base engine on a frame first");
  } else if (!input_is_tensor_ && !d_frame_) {
    throw std::runtime_error("no frame resident");
  }
  std::vector<hipEvent_t> ev(n + 1);
  for (auto& e : ev) VP_HIP_CHECK(hipEventCreate(&e));
  std::vector<double> acc(n, 0.0);
  for (int it = 0; it < iters + 1; ++it) {  // iteration 0 is a warm-up
    for (int i = (int)first; i < n; ++i) {
      VP_HIP_CHECK(hipEventRecord(ev[i], stream_));
      hipError_t e = ops_[i].run(stream_);
      if (e != hipSuccess) throw std::runtime_error("launch failed in layer '" + ops_[i].name + "'");
    }
    VP_HIP_CHECK(hipEventRecord(ev[n], stream_));
    VP_HIP_CHECK(hipStreamSynchronize(stream_));
    if (it == 0) continue;
    for (int i = (int)first; i < n; ++i) {
      float t = 0.f;
      VP_HIP_CHECK(hipEventElapsedTime(&t, ev[i], ev[i + 1]));
      acc[i] += t;
    }
  }
  for (int i = 0; i < n; ++i) ms[i] = (float)(acc[i] / std::max(1, iters));
  for (auto& e : ev) hipEventDestroy(e);
  warmed_ = true;
  have_outputs_ = true;
  return n;
}

void Engine::read_act(int i, float* dst) {
  if (i < 0 || i >= (int)acts_.size()) throw std::invalid_argument("tensor index out of range");
  const Act& a = *acts_[i];
  const size_t n = (size_t)a.Creal * a.H * a.W;
  float* d = nullptr;
  VP_HIP_CHECK(hipStreamSynchronize(stream_));
  VP_HIP_CHECK(hipMalloc(reinterpret_cast<void**>(&d), n * sizeof(float)));
  hipError_t e = launch_act_to_nchw(a.view(), a.Creal, d, stream_);
  if (e == hipSuccess) e = hipStreamSynchronize(stream_);
  if (e == hipSuccess) e = hipMemcpyAsync(dst, d, n * sizeof(float), hipMemcpyDeviceToHost, stream_);   // (not the legacy stream: engine.hpp copy_h2d)
  if (e == hipSuccess) e = hipStreamSynchronize(stream_);
  hipFree(d);
  VP_HIP_CHECK(e);
}

}  // namespace vp
