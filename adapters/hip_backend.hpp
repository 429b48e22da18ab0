// HipBackend -- drop-in for autoware_pov::vision::InferenceBackend
// (VisionPilot/middleware_recipes/common/include/inference_backend_base.hpp:14-27), built on the C ABI of
// libvp_hip.so (include/vp_hip.h).  Same construction / call / error conventions as OnnxRuntimeBackend
// (common/backends/onnx_runtime_backend.cpp:9-101) and TensorRTBackend (tensorrt_backend.cpp:35-217):
//   * ctor(model_path, precision, gpu_id) throws std::runtime_error / std::invalid_argument on failure;
//   * doInference(const cv::Mat& bgr8) returns false after logging on a runtime failure;
//   * getRawTensorData() returns HOST fp32 NCHW logits owned by the backend, valid until the next doInference;
//     getters throw std::runtime_error before the first inference (onnx_runtime_backend.cpp:86-91);
//   * one instance per node / thread, not re-entrant, no global state.
// `model_path` is the reference's `.onnx` file (read natively by libvp_hip, csrc/onnx_reader.cpp) or a VPW1 weight blob
// (autoware_vision_pilot_amd/weights.py export_checkpoint converts the reference .pth).  precision: "fp16" (fast) or "fp32" (fp16x3 parity mode).  Header-only; include it from the node exactly
// where onnx_runtime_backend.hpp is included.
#ifndef HIP_BACKEND_HPP_
#define HIP_BACKEND_HPP_

#include <map>
#include <mutex>
#include <stdexcept>
#include <string>
#include <vector>

#include "inference_backend_base.hpp"  // the reference's abstract base
#include "vp_hip.h"

#ifndef LOG_ERROR
#include <cstdio>
#define VP_ADAPTER_LOG_ERROR(...) (std::fprintf(stderr, __VA_ARGS__), std::fprintf(stderr, "\n"))
#else
#define VP_ADAPTER_LOG_ERROR(...) LOG_ERROR(__VA_ARGS__)
#endif

namespace autoware_pov::vision
{

// Host logits pointers handed out by live HipBackend instances -> their engines.  Lets the reference's STATIC mask helper
// (MasksVisualizationKernels::createMaskFromTensorHIP(const float*, shape, cv::Mat&), see masks_visualization_kernels_hip.hpp)
// recognise "this tensor is an engine's own output" and return the mask that engine already decoded on the device.
class HipTensorRegistry
{
public:
  static HipTensorRegistry & instance()
  {
    static HipTensorRegistry r;
    return r;
  }
  void add(const float * p, vp_engine * e)
  {
    std::lock_guard<std::mutex> g(m_);
    map_[p] = e;
  }
  void remove(vp_engine * e)
  {
    std::lock_guard<std::mutex> g(m_);
    for (auto it = map_.begin(); it != map_.end();) it = it->second == e ? map_.erase(it) : std::next(it);
  }
  vp_engine * find(const float * p)
  {
    std::lock_guard<std::mutex> g(m_);
    auto it = map_.find(p);
    return it == map_.end() ? nullptr : it->second;
  }

private:
  std::mutex m_;
  std::map<const float *, vp_engine *> map_;
};

class HipBackend : public InferenceBackend
{
public:
  // model_type: "segmentation" (SceneSeg), "depth" (Scene3D), "domain" (DomainSeg), "egolanes" -- the node's
  // `model_type` parameter (run_model_node.cpp:35) selects the network the same way it selects the post-process.
  HipBackend(const std::string & model_path, const std::string & precision, int gpu_id,
             const std::string & model_type = "segmentation", const std::string & plan = "latency")
  {
    int kind;
    if (model_type == "segmentation") kind = VP_SCENESEG;
    else if (model_type == "depth") kind = VP_SCENE3D;
    else if (model_type == "domain") kind = VP_DOMAINSEG;
    else if (model_type == "egolanes") kind = VP_EGOLANES;
    else throw std::invalid_argument("HipBackend: unknown model_type '" + model_type + "'");
    int prec;
    if (precision == "fp16") prec = VP_FP16;
    else if (precision == "fp32" || precision == "fp16x3") prec = VP_FP16X3;
    else throw std::invalid_argument("HipBackend: precision must be fp16 or fp32, got '" + precision + "'");
    // One backend instance = one network on one camera, one frame at a time (run_model_node.cpp:79-86): nothing runs beside a layer on the CUs the
    // default (several-cameras / forked-heads) kernel plan frees, so THIS engine is created with the latency plan (a creation flag since round 6: no
    // process-wide state is touched; a process that packs several cameras onto one GPU passes plan = "throughput").  SceneSeg alone: p50 1.91 -> 1.80 ms
    // (profiles/r05_plan_target_ab.txt).
    if (plan == "latency") prec |= VP_PLAN_LATENCY;
    else if (plan != "throughput") throw std::invalid_argument("HipBackend: plan must be latency or throughput, got '" + plan + "'");
    char err[512] = {0};
    const int rc = vp_create(&engine_, kind, model_path.c_str(), prec, gpu_id, err, sizeof(err));
    if (rc != VP_OK) {
      if (rc == VP_ERR_ARG) throw std::invalid_argument(std::string("HipBackend: ") + err);
      throw std::runtime_error(std::string("HipBackend: ") + err);
    }
    // middleware 'common' convention: BGR8 frame in, B,G,R planes with BGR-ordered constants
    // (onnx_runtime_backend.cpp:47-57); the EgoLanes engines use RGB planes (onnxruntime_engine.cpp:80-100)
    vp_set_input_format(engine_, VP_BGR8, kind == VP_EGOLANES ? VP_PLANES_RGB : VP_PLANES_BGR);
    // the C++ front-end's own float expression: convertTo(CV_32FC3, 1.0 / 255.0) = q * fl(1/255), then subtract / divide
    // (onnx_runtime_backend.cpp:45-49, tensorrt_backend.cpp:164-168) -- not torchvision's q / 255
    vp_set_norm_form(engine_, VP_NORM_OPENCV);
    // What doInference copies to the host every frame (round 5): the decoded class map only.  The node consumes the mask when its GPU helper
    // succeeds (run_model_node.cpp:144-177; createMask / createMaskFromTensorHIP below) and, for depth, the map createDepth resizes on the
    // device; the 2.4 MB (0.8 MB) fp32 tensor stays in HBM and is fetched by the first getRawTensorData() that asks for it -- the pointer's
    // contents and lifetime are as before.  setCopyLogitsEveryFrame(true) restores the copy inside doInference.
    vp_set_outputs(engine_, kind == VP_SCENE3D ? 0 : VP_OUT_MASK);
    vp_input_hw(engine_, &in_h_, &in_w_);
  }
  ~HipBackend() override
  {
    HipTensorRegistry::instance().remove(engine_);
    vp_destroy(engine_);
  }
  HipBackend(const HipBackend &) = delete;
  HipBackend & operator=(const HipBackend &) = delete;

  // FNV-1a over the engine's kernel plan (vp_plan_hash): tells the latency plan from the throughput plan
  unsigned long long planHash() const { return vp_plan_hash(engine_); }

  bool doInference(const cv::Mat & input_image) override
  {
    if (input_image.empty() || input_image.type() != CV_8UC3) {
      VP_ADAPTER_LOG_ERROR("HipBackend: expected a non-empty CV_8UC3 (BGR8) image");
      return false;
    }
    const int rc = vp_infer(engine_, input_image.data, input_image.rows, input_image.cols, static_cast<int>(input_image.step));
    if (rc != VP_OK) {
      VP_ADAPTER_LOG_ERROR("HIP inference failed: %s", vp_last_error(engine_));
      return false;
    }
    ran_ = true;
    return true;
  }

  const float * getRawTensorData() const override
  {
    const float * data = nullptr;
    int64_t shape[4];
    if (!ran_ || vp_logits(engine_, &data, shape) != VP_OK)
      throw std::runtime_error("Inference has not been run yet. Call doInference() first.");
    HipTensorRegistry::instance().add(data, engine_);
    return data;
  }
  std::vector<int64_t> getTensorShape() const override
  {
    int64_t shape[4];
    if (!ran_ || vp_output_shape(engine_, shape) != VP_OK)   // the shape alone: no fetch of a tensor nobody asked for
      throw std::runtime_error("Inference has not been run yet. Call doInference() first.");
    return {shape[0], shape[1], shape[2], shape[3]};
  }
  int getModelInputHeight() const override { return in_h_; }
  int getModelInputWidth() const override { return in_w_; }

  // ---- beyond the base interface: decode + resize done on the GPU, replacing the node's CPU loops
  // (run_model_node.cpp:144-177) and MasksVisualizationKernels::createMaskFromTensor{CUDA,HIP}.
  // Returns false (caller falls back to its CPU loop) on failure, like the reference helpers.
  bool createMask(cv::Mat & output_mask, const cv::Size & frame_size)
  {
    if (!ran_) return false;
    output_mask.create(frame_size, CV_8UC1);
    if (!output_mask.isContinuous()) return false;
    return vp_mask_resized_u8(engine_, output_mask.data, frame_size.height, frame_size.width) == VP_OK;
  }
  bool createDepth(cv::Mat & output_depth, const cv::Size & frame_size)
  {
    if (!ran_) return false;
    output_depth.create(frame_size, CV_32FC1);
    if (!output_depth.isContinuous()) return false;
    return vp_depth_resized_f32(engine_, reinterpret_cast<float *>(output_depth.data), frame_size.height, frame_size.width) == VP_OK;
  }

  // MasksVisualizationEngine::visualize (common/visualizers/masks_visualization_engine.cpp:11-38) on the GPU: colour
  // LUT + nearest resize + 50/50 blend with the frame of the last doInference; viz_type "scene" | "domain" | "egolanes".
  bool visualizeMask(const std::string & viz_type, cv::Mat & blended, const cv::Size & frame_size)
  {
    if (!ran_) return false;
    const int t = viz_type == "scene" ? VP_VIZ_SCENE : viz_type == "domain" ? VP_VIZ_DOMAIN : viz_type == "egolanes" ? VP_VIZ_EGOLANES : -1;
    if (t < 0) return false;
    int fh = 0, fw = 0;  // the blend is defined on the frame of the last doInference: refuse any other geometry
    if (vp_frame_hw(engine_, &fh, &fw) != VP_OK || fh != frame_size.height || fw != frame_size.width) return false;
    blended.create(frame_size, CV_8UC3);
    if (!blended.isContinuous()) return false;
    return vp_visualize_mask_bgr8(engine_, t, blended.data, frame_size.height, frame_size.width) == VP_OK;
  }

  // Default (round 5): the logits stay in HBM until getRawTensorData() asks for them; a host that reads the raw tensor every frame
  // (the unpatched depth path, run_model_node.cpp:94-104) saves the extra synchronisation by copying it inside doInference again.
  void setCopyLogitsEveryFrame(bool on) { vp_set_outputs(engine_, (on ? VP_OUT_LOGITS : 0) | VP_OUT_MASK); }
  // The node's frame pool (cv::Mat buffers it reuses): page-locked once, then doInference DMAs straight from the cv::Mat with no staging copy
  // (vp_register_frames).  The pool must outlive the registration.
  static bool registerFramePool(const void * pool, size_t bytes) { return vp_register_frames(pool, bytes) == VP_OK; }
  static bool unregisterFramePool(const void * pool) { return vp_unregister_frames(pool) == VP_OK; }
  vp_engine * handle() { return engine_; }

private:
  vp_engine * engine_ = nullptr;
  int in_h_ = 0, in_w_ = 0;
  bool ran_ = false;
};

}  // namespace autoware_pov::vision

#endif  // HIP_BACKEND_HPP_
