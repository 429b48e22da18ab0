// Compile-check stand-in for VisionPilot/production_release/include/inference/lane_segmentation.hpp:16-44
// (only the members EgoLanesHipEngine fills; a real build includes the reference's own header).
#pragma once
#include <opencv2/opencv.hpp>

namespace autoware_pov::vision::egolanes
{
struct LaneSegmentation
{
  cv::Mat ego_left, ego_right, other_lanes;
  int height = 0;
  int width = 0;
};
}  // namespace autoware_pov::vision::egolanes
