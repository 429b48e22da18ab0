// MasksVisualizationKernels::createMaskFromTensorHIP -- the reference's static helper
// (VisionPilot/middleware_recipes/common/include/masks_visualization_kernels.hpp:37-45, defined for ROCm builds in
// common/visualizers/masks_viz.hip.cpp:41-97 and called by the Zenoh model runner) with the SAME signature, on libvp_hip.
// Compile this header INSTEAD of masks_viz.hip.cpp in a HIP_FOUND build (exactly one translation unit must include it with
// VP_HIP_DEFINE_MASK_KERNELS defined; it provides the member definitions the reference header declares).
//
//   * tensor_data is the pointer a live HipBackend::getRawTensorData() returned, the engine decodes with the reference's rule
//     (VP_DECODE_SEG_MASK) and the host tensor holds the LAST frame's logits: the engine decoded them at the end of that frame --
//     the mask (0.2 MB) is copied down, nothing is uploaded;
//   * anything else (another host tensor; an engine in another decode mode, e.g. an EgoLanes backend's lane labels; a pointer kept
//     from an earlier frame while logits copies are switched off): vp_decode_logits_host = upload + decode kernel + download on
//     the tensor's own engine's GPU (GPU 0 for foreign tensors), the reference helper's own shape.
// Decode rule (bit-exact contract): C > 1 -> 255 where the first maximum is class 1, else 0; C == 1 -> 255 where value > 0.
#ifndef MASKS_VISUALIZATION_KERNELS_HIP_HPP_
#define MASKS_VISUALIZATION_KERNELS_HIP_HPP_

#ifndef HIP_FOUND
#define HIP_FOUND 1
#endif
#include "masks_visualization_kernels.hpp"  // the reference's declaration

#include <cstring>

#include "hip_backend.hpp"

#ifdef VP_HIP_DEFINE_MASK_KERNELS
namespace autoware_pov::common
{

bool MasksVisualizationKernels::createMaskFromTensorHIP(
  const float * tensor_data, const std::vector<int64_t> & tensor_shape, cv::Mat & output_mask)
{
  if (!tensor_data || tensor_shape.size() != 4) return false;
  const int channels = static_cast<int>(tensor_shape[1]);
  const int rows = static_cast<int>(tensor_shape[2]), cols = static_cast<int>(tensor_shape[3]);
  if (channels < 1 || rows < 1 || cols < 1) return false;
  output_mask.create(rows, cols, CV_8UC1);
  int gpu = 0;
  if (vp_engine * e = autoware_pov::vision::HipTensorRegistry::instance().find(tensor_data)) {
    gpu = vp_gpu_id(e) >= 0 ? vp_gpu_id(e) : 0;
    const uint8_t * mask = nullptr;
    int h = 0, w = 0;
    if (vp_get_decode_mode(e) == VP_DECODE_SEG_MASK && vp_host_logits_current(e) == 1 &&
        vp_mask_u8(e, &mask, &h, &w) == VP_OK && h == rows && w == cols) {
      for (int y = 0; y < rows; ++y) std::memcpy(output_mask.data + (size_t)y * output_mask.step, mask + (size_t)y * cols, (size_t)cols);
      return true;
    }
  }
  if (!output_mask.isContinuous()) return false;
  return vp_decode_logits_host(gpu, tensor_data, channels, rows, cols, VP_DECODE_SEG_MASK, output_mask.data) == VP_OK;
}

}  // namespace autoware_pov::common
#endif  // VP_HIP_DEFINE_MASK_KERNELS

#endif  // MASKS_VISUALIZATION_KERNELS_HIP_HPP_
