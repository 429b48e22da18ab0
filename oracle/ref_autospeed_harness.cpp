// TEST INFRASTRUCTURE (see oracle/__init__.py).  Drives the REFERENCE'S OWN AutoSpeedOnnxEngine -- compiled from its source where it lies
// under /root/reference (oracle/Makefile; never copied into this repository) on top of the stand-in headers of oracle/ref_stubs -- so that
// oracle/autospeed.py is pinned against the real thing:
//   post  RAW.bin ATTRS BOXES CONF IOU SCALE PAD_X PAD_Y ORIG_W ORIG_H OUT.bin
//         the reference's postProcess (+ computeIoU + applyNMS, onnxruntime_engine.cpp:170-290) on a detector tensor [ATTRS][BOXES];
//         OUT.bin = int32 n, then n x {x1, y1, x2, y2, confidence (fp32), class_id (int32)}
//   pre   FRAME.bin H W OUT.bin
//         the reference's preprocessAutoSpeed (:71-113) on a BGR8 frame; OUT.bin = fp32 scale_, int32 pad_x_, pad_y_, then the [3][640][640]
//         fp32 tensor.  cv::resize is the placeholder of oracle/ref_stubs (nearest neighbour): geometry, canvas, /255 and plane order are
//         the reference's, the resize arithmetic is not OpenCV's.
#define private public  // the two stages are private members of the reference class
#include "onnxruntime_engine.hpp"
#include "onnxruntime_session.hpp"
#undef private

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>

namespace autoware_pov::vision::autospeed {
// the factory the reference engine's constructor calls (its real implementation, onnxruntime_session.cpp, needs ONNX Runtime)
int OnnxRuntimeSessionFactory::num_threads_ = 1;
std::unique_ptr<Ort::Session> OnnxRuntimeSessionFactory::createSession(const std::string&, const std::string&, const std::string&, int, const std::string&) {
  return std::make_unique<Ort::Session>();
}
}  // namespace autoware_pov::vision::autospeed

using autoware_pov::vision::autospeed::AutoSpeedOnnxEngine;
using autoware_pov::vision::autospeed::Detection;

static std::vector<char> slurp(const char* path) {
  std::ifstream f(path, std::ios::binary);
  return std::vector<char>((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
}

int main(int argc, char** argv) {
  if (argc < 2) return 2;
  AutoSpeedOnnxEngine eng("none.onnx", "cpu", "fp32", 0, "");
  if (!std::strcmp(argv[1], "post") && argc == 13) {
    const std::vector<char> raw = slurp(argv[2]);
    const int attrs = std::atoi(argv[3]), boxes = std::atoi(argv[4]);
    if (raw.size() != (size_t)attrs * boxes * sizeof(float)) return 3;
    Ort::Value v;
    v.data.assign(reinterpret_cast<const float*>(raw.data()), reinterpret_cast<const float*>(raw.data()) + (size_t)attrs * boxes);
    v.shape = {1, attrs, boxes};
    eng.output_tensors_.clear();
    eng.output_tensors_.push_back(v);
    eng.model_output_channels_ = attrs;
    eng.model_output_predictions_ = boxes;
    eng.scale_ = (float)std::atof(argv[7]);
    eng.pad_x_ = std::atoi(argv[8]);
    eng.pad_y_ = std::atoi(argv[9]);
    eng.orig_width_ = std::atoi(argv[10]);
    eng.orig_height_ = std::atoi(argv[11]);
    const std::vector<Detection> det = eng.postProcess((float)std::atof(argv[5]), (float)std::atof(argv[6]));
    std::ofstream o(argv[12], std::ios::binary);
    const int n = (int)det.size();
    o.write(reinterpret_cast<const char*>(&n), sizeof n);
    o.write(reinterpret_cast<const char*>(det.data()), (std::streamsize)(det.size() * sizeof(Detection)));
    return 0;
  }
  if (!std::strcmp(argv[1], "pre") && argc == 6) {
    const std::vector<char> fr = slurp(argv[2]);
    const int h = std::atoi(argv[3]), w = std::atoi(argv[4]);
    if (fr.size() != (size_t)h * w * 3) return 3;
    cv::Mat img(h, w, CV_8UC3);
    std::memcpy(img.data, fr.data(), fr.size());
    std::vector<float> buf((size_t)3 * eng.getInputHeight() * eng.getInputWidth());
    eng.preprocessAutoSpeed(img, buf.data());
    std::ofstream o(argv[5], std::ios::binary);
    o.write(reinterpret_cast<const char*>(&eng.scale_), sizeof(float));
    o.write(reinterpret_cast<const char*>(&eng.pad_x_), sizeof(int));
    o.write(reinterpret_cast<const char*>(&eng.pad_y_), sizeof(int));
    o.write(reinterpret_cast<const char*>(buf.data()), (std::streamsize)(buf.size() * sizeof(float)));
    return 0;
  }
  return 2;
}
