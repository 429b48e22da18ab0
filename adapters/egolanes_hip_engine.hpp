// EgoLanesHipEngine -- drop-in for autoware_pov::vision::egolanes::EgoLanesOnnxEngine / EgoLanesTensorRTEngine
// (VisionPilot/production_release/include/inference/onnxruntime_engine.hpp:27-125, tensorrt_engine.hpp:40-148;
// implementation onnxruntime_engine.cpp:13-210), built on libvp_hip.so.  Same public methods, argument meaning and
// failure behaviour: ctor throws std::runtime_error; inference() returns an empty LaneSegmentation{} on failure
// (onnxruntime_engine.cpp:141-145); getRawTensorData()/getTensorShape() throw before the first inference.
// Called from the lateral thread (production_release/main.cpp:505,515); several engines may coexist in one process.
#ifndef EGOLANES_HIP_ENGINE_HPP_
#define EGOLANES_HIP_ENGINE_HPP_

#include <stdexcept>
#include <string>
#include <vector>

#include "inference/lane_segmentation.hpp"  // the reference's LaneSegmentation
#include "vp_hip.h"

namespace autoware_pov::vision::egolanes
{

class EgoLanesHipEngine
{
public:
  // provider / cache_dir are accepted for signature compatibility (the ONNX/TensorRT engines use them); precision:
  // "fp16" or "fp32" (parity mode).  model_path: the EgoLanes `.onnx` file or a VPW1 blob exported from the checkpoint.
  EgoLanesHipEngine(const std::string & model_path, const std::string & provider = "hip", const std::string & precision = "fp16",
                    int device_id = 0, const std::string & cache_dir = "", bool latency_plan = false)
  {
    (void)provider;
    (void)cache_dir;
    char err[512] = {0};
    // The production app runs this engine on its lateral thread BESIDE the AutoSpeed / AutoSteer engines' threads on the same GPU
    // (production_release/main.cpp:505-535): the default (throughput) kernel plan leaves them CUs.  latency_plan = true (a creation flag of THIS engine,
    // no process-wide state) is for a host that runs it alone: EgoLanes p50 1.43 -> 1.32 ms.
    const int prec = ((precision == "fp32" || precision == "fp16x3") ? VP_FP16X3 : VP_FP16) | (latency_plan ? VP_PLAN_LATENCY : 0);
    if (vp_create(&engine_, VP_EGOLANES, model_path.c_str(), prec, device_id, err, sizeof(err)) != VP_OK)
      throw std::runtime_error(std::string("[hip_engine] ") + err);
    vp_set_input_format(engine_, VP_BGR8, VP_PLANES_RGB);  // resize, BGR->RGB, ImageNet norm: onnxruntime_engine.cpp:72-102
    vp_set_norm_form(engine_, VP_NORM_OPENCV);             // convertTo(CV_32FC3, 1.0 / 255.0), then (x - MEAN[c]) / STD[c]: :85, :94-100
    vp_input_hw(engine_, &in_h_, &in_w_);
  }
  ~EgoLanesHipEngine() { vp_destroy(engine_); }
  EgoLanesHipEngine(const EgoLanesHipEngine &) = delete;
  EgoLanesHipEngine & operator=(const EgoLanesHipEngine &) = delete;

  unsigned long long planHash() const { return vp_plan_hash(engine_); }

  LaneSegmentation inference(const cv::Mat & input_image, float threshold = 0.0f)
  {
    if (input_image.empty() || input_image.type() != CV_8UC3 ||
        vp_infer(engine_, input_image.data, input_image.rows, input_image.cols, static_cast<int>(input_image.step)) != VP_OK) {
      return LaneSegmentation{};
    }
    const float * raw = nullptr;
    int64_t shape[4];
    if (vp_logits(engine_, &raw, shape) != VP_OK) return LaneSegmentation{};
    ran_ = true;
    out_h_ = static_cast<int>(shape[2]);
    out_w_ = static_cast<int>(shape[3]);
    // postProcess (onnxruntime_engine.cpp:151-192): three CV_32FC1 0/1 planes, '> threshold'
    LaneSegmentation result;
    result.height = out_h_;
    result.width = out_w_;
    const int n = out_h_ * out_w_;
    cv::Mat * planes[3] = {&result.ego_left, &result.ego_right, &result.other_lanes};
    for (int c = 0; c < 3; ++c) {
      planes[c]->create(out_h_, out_w_, CV_32FC1);
      float * dst = reinterpret_cast<float *>(planes[c]->data);
      const float * src = raw + static_cast<size_t>(c) * n;
      for (int i = 0; i < n; ++i) dst[i] = src[i] > threshold ? 1.0f : 0.0f;
    }
    return result;
  }

  const float * getRawTensorData() const
  {
    const float * raw = nullptr;
    int64_t shape[4];
    if (!ran_ || vp_logits(engine_, &raw, shape) != VP_OK)
      throw std::runtime_error("Inference has not been run yet. Call inference() first.");
    return raw;
  }
  std::vector<int64_t> getTensorShape() const
  {
    const float * raw = nullptr;
    int64_t shape[4];
    if (!ran_ || vp_logits(engine_, &raw, shape) != VP_OK)
      throw std::runtime_error("Inference has not been run yet. Call inference() first.");
    return {shape[0], shape[1], shape[2], shape[3]};
  }
  // AutoSteer hand-over (main.cpp:472-534): the raw logits of the last two inference() calls, concatenated [t-1 | t] = {1, 6, 80, 160}, kept on
  // the DEVICE by the engine (vp_set_lane_ring) instead of the app's circular buffer of host copies + two memcpys per frame.
  //   engine.enableAutoSteerInput();                                   // once, after construction
  //   if (engine.autoSteerInput(autosteer_input_buffer))              // false until two frames are in (the reference's `buffer.full()` test, :521)
  //     steering = autosteer_engine->inference(autosteer_input_buffer);
  // autoSteerInputDevice() hands a device-resident AutoSteer runtime the same 6 x 80 x 160 floats without the D2H.
  void enableAutoSteerInput()
  {
    if (vp_set_lane_ring(engine_, 1) != VP_OK) throw std::runtime_error(std::string("[hip_engine] ") + vp_last_error(engine_));
  }
  bool autoSteerInput(std::vector<float> & buffer)
  {
    int frames = 0;
    buffer.resize(static_cast<size_t>(6) * out_h_ * out_w_);
    if (vp_lane_ring_fetch(engine_, buffer.data(), &frames) != VP_OK) return false;
    return frames >= 2;
  }
  const float * autoSteerInputDevice(int * frames_valid = nullptr) const
  {
    void * p = nullptr;
    return vp_lane_ring_device(engine_, &p, frames_valid) == VP_OK ? static_cast<const float *>(p) : nullptr;
  }
  int getInputWidth() const { return in_w_; }
  int getInputHeight() const { return in_h_; }
  int getOutputWidth() const { return out_w_; }
  int getOutputHeight() const { return out_h_; }

private:
  vp_engine * engine_ = nullptr;
  int in_h_ = 0, in_w_ = 0, out_h_ = 80, out_w_ = 160;
  bool ran_ = false;
};

}  // namespace autoware_pov::vision::egolanes

#endif  // EGOLANES_HIP_ENGINE_HPP_
