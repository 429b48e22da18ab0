// Stand-in for the reference's common/include/masks_visualization_kernels.hpp, used ONLY where the reference tree is absent
// (the GPU box): just the declaration the HIP shim defines.  With /root/reference present the adapter check compiles against
// the reference's own header instead (adapters/Makefile).
#pragma once
#include <cstdint>
#include <opencv2/opencv.hpp>
#include <vector>

namespace autoware_pov::common
{
class MasksVisualizationKernels
{
public:
#ifdef HIP_FOUND
  static bool createMaskFromTensorHIP(const float * tensor_data, const std::vector<int64_t> & tensor_shape, cv::Mat & output_mask);
#endif
};
}  // namespace autoware_pov::common
