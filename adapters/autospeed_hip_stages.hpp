// AutoSpeedHipStages -- the two CPU stages of autoware_pov::vision::autospeed::AutoSpeedOnnxEngine / AutoSpeedTensorRTEngine on the
// MI355X, built on libvp_hip.so (vp_detect_*): preprocessAutoSpeed (VisionPilot/middleware_recipes/common/backends/autospeed/
// onnxruntime_engine.cpp:71-113) and postProcess + computeIoU + applyNMS (:170-290).  The detector network stays with the engine's own
// runtime; a maintainer swaps the two private methods for calls into this class (INTEGRATION.md shows the patch):
//     void   preprocessAutoSpeed(const cv::Mat & image, float * buffer)         -> stages_.preprocessAutoSpeed(image, buffer)
//     std::vector<Detection> postProcess(float conf_thresh, float iou_thresh)   -> stages_.postProcess(raw, channels, predictions, conf, iou)
// Same results as the CPU code, bit for bit (equal confidences keep the box order, which std::sort leaves open); Detection is the
// reference's struct (detection.hpp:8-12), vp_detection has its layout.
#ifndef AUTOSPEED_HIP_STAGES_HPP_
#define AUTOSPEED_HIP_STAGES_HPP_

#include <cstring>
#include <cstdio>
#include <stdexcept>
#include <string>
#include <vector>

#include <opencv2/opencv.hpp>

#include "vp_hip.h"

namespace autoware_pov::vision::autospeed
{

#ifndef AUTOSPEED_DETECTION_HPP_
#define AUTOSPEED_DETECTION_HPP_
struct Detection {  // detection.hpp:8-12 (used when the reference header is not on the include path)
  float x1, y1, x2, y2;
  float confidence;
  int class_id;
};
#endif
static_assert(sizeof(Detection) == sizeof(vp_detection), "Detection and vp_detection share one layout");

class AutoSpeedHipStages
{
public:
  // input_width x input_height: the detector's input (640 x 640); max_predictions / max_channels bound its output tensor [channels][predictions]
  AutoSpeedHipStages(int input_width = 640, int input_height = 640, int max_predictions = 8400, int max_channels = 84, int device_id = 0)
  {
    char err[512] = {0};
    if (vp_detect_create(&d_, device_id, input_height, input_width, max_predictions, max_channels, err, sizeof(err)) != VP_OK)
      throw std::runtime_error(std::string("[hip_stages] ") + err);
  }
  ~AutoSpeedHipStages() { vp_detect_destroy(d_); }
  AutoSpeedHipStages(const AutoSpeedHipStages &) = delete;
  AutoSpeedHipStages & operator=(const AutoSpeedHipStages &) = delete;

  // letterbox to the detector's input, / 255, planes R, G, B into `buffer` (3 * H * W floats, host); remembers scale_, pad_x_, pad_y_, orig size
  void preprocessAutoSpeed(const cv::Mat & input_image, float * buffer)
  {
    if (input_image.empty() || input_image.type() != CV_8UC3)   // the C call is never made: its last-error text would be a stale one
      throw std::runtime_error("[hip_stages] preprocess: input image is empty or not CV_8UC3");
    if (vp_detect_preprocess(d_, input_image.data, input_image.rows, input_image.cols, static_cast<int>(input_image.step), buffer) != VP_OK)
      throw std::runtime_error(std::string("[hip_stages] preprocess: ") + vp_detect_last_error(d_));
  }
  // the tensor the last preprocessAutoSpeed produced, on the device (a runtime that binds device memory skips the host copy)
  const float * inputDevice() const
  {
    void * p = nullptr;
    vp_detect_input_device(d_, &p);
    return static_cast<const float *>(p);
  }
  // raw_output: [num_attrs][num_boxes] fp32 on the host (output_tensors_[0].GetTensorData<float>()).  {} for "no tensor" (null pointer /
  // no boxes), as the reference returns {} when it has no output tensor; a DEVICE failure (VP_ERR_HIP) throws instead of looking like
  // "no detections", and a bad argument is logged through lastError() before the empty return
  std::vector<Detection> postProcess(const float * raw_output, int num_attrs, int num_boxes, float conf_thresh, float iou_thresh, bool raw_on_device = false)
  {
    if (raw_output == nullptr || num_boxes <= 0) return {};
    std::vector<Detection> out(static_cast<size_t>(num_boxes));
    int n = 0;
    const int rc = vp_detect_postprocess(d_, raw_output, raw_on_device ? 1 : 0, num_attrs, num_boxes, conf_thresh, iou_thresh,
                                         reinterpret_cast<vp_detection *>(out.data()), static_cast<int>(out.size()), &n);
    if (rc == VP_ERR_HIP) throw std::runtime_error(std::string("[hip_stages] postProcess: ") + vp_detect_last_error(d_));
    if (rc != VP_OK) {
      std::fprintf(stderr, "[hip_stages] postProcess: %s\n", vp_detect_last_error(d_));
      return {};
    }
    out.resize(static_cast<size_t>(n));
    return out;
  }
  const char * lastError() const { return vp_detect_last_error(d_); }

private:
  vp_detect * d_ = nullptr;
};

}  // namespace autoware_pov::vision::autospeed

#endif  // AUTOSPEED_HIP_STAGES_HPP_
