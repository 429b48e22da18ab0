// Compile-check stand-in for VisionPilot/middleware_recipes/common/include/inference_backend_base.hpp:14-27
// (the abstract interface HipBackend derives from; a real build includes the reference's own header).
#pragma once
#include <opencv2/opencv.hpp>
#include <cstdint>
#include <vector>

namespace autoware_pov::vision
{
class InferenceBackend
{
public:
  virtual ~InferenceBackend() = default;
  virtual bool doInference(const cv::Mat & input_image) = 0;
  virtual const float * getRawTensorData() const = 0;
  virtual std::vector<int64_t> getTensorShape() const = 0;
  virtual int getModelInputHeight() const = 0;
  virtual int getModelInputWidth() const = 0;
};
}  // namespace autoware_pov::vision
