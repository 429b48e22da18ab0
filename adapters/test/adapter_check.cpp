// Compile + behaviour check of the C++ drop-in adapters against libvp_hip.so.
//   adapter_check                      -> construction-failure conventions (no GPU needed)
//   adapter_check <kind> <blob> <out>  -> one synthetic 720p BGR frame through HipBackend (kind = segmentation|depth|
//                                         domain) or EgoLanesHipEngine (kind = egolanes); writes logits + mask to <out>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <string>
#include <thread>

#include "autospeed_hip_stages.hpp"
#include "egolanes_hip_engine.hpp"
#include "hip_backend.hpp"
#define VP_HIP_DEFINE_MASK_KERNELS 1
#include "masks_visualization_kernels_hip.hpp"  // the reference's static helper signature on libvp_hip

using autoware_pov::vision::HipBackend;
using autoware_pov::vision::egolanes::EgoLanesHipEngine;

static int fail(const char * what)
{
  std::fprintf(stderr, "adapter_check FAILED: %s\n", what);
  return 1;
}

int main(int argc, char ** argv)
{
  // --- error conventions (onnx_runtime_backend.cpp:86-91, tensorrt_backend.cpp:58, run_model_node.cpp:45)
  try {
    HipBackend b("/nonexistent/model.vpw", "fp16", 0, "segmentation");
    return fail("ctor with a missing weight file must throw");
  } catch (const std::runtime_error &) {
  } catch (const std::invalid_argument &) {
    return fail("missing file must be runtime_error, not invalid_argument");
  }
  try {
    HipBackend b("x.vpw", "fp16", 0, "detector");
    return fail("unknown model_type must throw invalid_argument");
  } catch (const std::invalid_argument &) {
  }
  try {
    HipBackend b("x.vpw", "int8", 0);
    return fail("unknown precision must throw invalid_argument");
  } catch (const std::invalid_argument &) {
  }
  try {
    EgoLanesHipEngine e("/nonexistent/model.vpw");
    return fail("EgoLanes ctor with a missing weight file must throw");
  } catch (const std::runtime_error &) {
  }
  if (argc < 4) {
#ifdef VP_ADAPTER_CHECK_REAL_HEADERS
    std::puts("adapter_check: compiled against the reference's own inference_backend_base.hpp / masks_visualization_kernels.hpp");
#endif
    std::puts("adapter_check: construction-failure conventions OK");
    return 0;
  }
  // --- one frame through the drop-in classes, exactly as RunModelNode::onImage / lateralInferenceThread call them
  const std::string kind = argv[1], blob = argv[2], out = argv[3];
  cv::Mat frame(720, 1280, CV_8UC3);
  for (int y = 0; y < 720; ++y)
    for (int x = 0; x < 1280 * 3; ++x) frame.data[(size_t)y * frame.step + x] = (uint8_t)((x * 7 + y * 13 + (x ^ y)) & 255);
  std::ofstream f(out, std::ios::binary);
  f.write(reinterpret_cast<const char *>(frame.data), (std::streamsize)frame.step * 720);
  if (kind == "autospeed") {
    // the AutoSpeed engine's two CPU stages (blob = a raw fp32 detector output [attrs][boxes] written by the test): letterbox tensor,
    // then the kept detections, appended to <out> after the frame
    using autoware_pov::vision::autospeed::AutoSpeedHipStages;
    using autoware_pov::vision::autospeed::Detection;
    if (argc < 6) return fail("autospeed: adapter_check autospeed RAW.bin OUT ATTRS BOXES");
    const int attrs = std::atoi(argv[4]), boxes = std::atoi(argv[5]);
    std::ifstream rf(blob, std::ios::binary);
    std::vector<float> raw((size_t)attrs * boxes);
    rf.read(reinterpret_cast<char *>(raw.data()), (std::streamsize)(raw.size() * sizeof(float)));
    if (!rf) return fail("autospeed: raw tensor file too short");
    AutoSpeedHipStages st(640, 640, boxes, attrs);
    std::vector<float> input((size_t)3 * 640 * 640);
    st.preprocessAutoSpeed(frame, input.data());
    if (st.inputDevice() == nullptr) return fail("autospeed: device tensor");
    f.write(reinterpret_cast<const char *>(input.data()), (std::streamsize)(input.size() * sizeof(float)));
    const std::vector<Detection> det = st.postProcess(raw.data(), attrs, boxes, 0.25f, 0.45f);
    const int n = (int)det.size();
    f.write(reinterpret_cast<const char *>(&n), sizeof n);
    f.write(reinterpret_cast<const char *>(det.data()), (std::streamsize)(det.size() * sizeof(Detection)));
    if (!st.postProcess(raw.data(), 3, boxes, 0.25f, 0.45f).empty()) return fail("autospeed: a tensor without class rows must give {}");
    cv::Mat bad(10, 10, CV_8UC1);
    try {
      st.preprocessAutoSpeed(bad, input.data());
      return fail("autospeed: a non-BGR8 image must throw");
    } catch (const std::runtime_error &) {
    }
  } else if (kind == "threads") {
    // round 6: the plan target is a creation flag of each engine -- a B1 backend (latency plan) and a B2 engine (throughput plan, the production
    // app's several-engines-per-process case: main.cpp:505-535) constructed CONCURRENTLY from two threads, one frame each; no process-wide option is
    // touched, and each engine's plan is the one a plain vp_create with / without VP_PLAN_LATENCY builds.  argv: threads <seg blob> <out> <ego blob>
    if (argc < 5) return fail("threads: <seg blob> <out> <ego blob>");
    const std::string ego_blob = argv[4];
    unsigned long long h_b1 = 0, h_b2 = 0;
    std::string err1, err2;
    std::thread t1([&] {
      try {
        HipBackend b(blob, "fp32", 0, "segmentation");
        if (!b.doInference(frame)) err1 = "doInference";
        h_b1 = b.planHash();
      } catch (const std::exception & ex) { err1 = ex.what(); }
    });
    std::thread t2([&] {
      try {
        EgoLanesHipEngine e(ego_blob, "hip", "fp32");
        if (e.inference(frame, 0.0f).ego_left.empty()) err2 = "inference";
        h_b2 = e.planHash();
      } catch (const std::exception & ex) { err2 = ex.what(); }
    });
    t1.join();
    t2.join();
    if (!err1.empty() || !err2.empty()) return fail(("threads: " + err1 + " / " + err2).c_str());
    if (vp_get_option("VP_PLAN_TARGET") != nullptr) return fail("threads: an adapter set a process-wide option");
    char err[256] = {0};
    vp_engine * ref = nullptr;
    unsigned long long want[4] = {0, 0, 0, 0};
    const struct { int kind; const char * path; int prec; } mk[4] = {{VP_SCENESEG, blob.c_str(), VP_FP16X3 | VP_PLAN_LATENCY}, {VP_SCENESEG, blob.c_str(), VP_FP16X3},
                                                                      {VP_EGOLANES, ego_blob.c_str(), VP_FP16X3}, {VP_EGOLANES, ego_blob.c_str(), VP_FP16X3 | VP_PLAN_LATENCY}};
    for (int i = 0; i < 4; ++i) {
      if (vp_create(&ref, mk[i].kind, mk[i].path, mk[i].prec, 0, err, sizeof(err)) != VP_OK) return fail(err);
      want[i] = vp_plan_hash(ref);
      vp_destroy(ref);
    }
    if (h_b1 != want[0] || h_b1 == want[1]) return fail("threads: the B1 backend is not on the latency plan");
    if (h_b2 != want[2] || h_b2 == want[3]) return fail("threads: the B2 engine is not on the throughput plan");
    f.write(reinterpret_cast<const char *>(&h_b1), 8);
    f.write(reinterpret_cast<const char *>(&h_b2), 8);
  } else if (kind == "egolanes") {
    EgoLanesHipEngine e(blob, "hip", "fp32");
    try {
      e.getRawTensorData();
      return fail("getRawTensorData before inference must throw");
    } catch (const std::runtime_error &) {
    }
    auto seg = e.inference(frame, 0.0f);
    if (seg.height != 80 || seg.width != 160 || seg.ego_left.empty()) return fail("LaneSegmentation geometry");
    const auto shape = e.getTensorShape();
    if (shape.size() != 4 || shape[1] != 3) return fail("egolanes tensor shape");
    f.write(reinterpret_cast<const char *>(e.getRawTensorData()), sizeof(float) * 3 * 80 * 160);
    f.write(reinterpret_cast<const char *>(seg.ego_left.data), sizeof(float) * 80 * 160);
  } else {
    HipBackend b(blob, "fp32", 0, kind);
    try {
      b.getRawTensorData();
      return fail("getRawTensorData before doInference must throw");
    } catch (const std::runtime_error &) {
    }
    if (b.getModelInputHeight() != 320 || b.getModelInputWidth() != 640) return fail("model input size");
    if (!b.doInference(frame)) return fail("doInference returned false");
    const auto shape = b.getTensorShape();
    const size_t n = (size_t)shape[1] * shape[2] * shape[3];
    f.write(reinterpret_cast<const char *>(b.getRawTensorData()), sizeof(float) * n);
    cv::Mat mask;
    if (kind == "depth") {
      if (!b.createDepth(mask, frame.size())) return fail("createDepth");
      f.write(reinterpret_cast<const char *>(mask.data), sizeof(float) * 720 * 1280);
    } else {
      if (!b.createMask(mask, frame.size())) return fail("createMask");
      f.write(reinterpret_cast<const char *>(mask.data), 720 * 1280);
    }
    if (kind != "depth") {
      // the reference's static helper, same signature: (a) on the backend's own tensor (no upload), (b) on a foreign copy
      using autoware_pov::common::MasksVisualizationKernels;
      cv::Mat m1, m2;
      if (!MasksVisualizationKernels::createMaskFromTensorHIP(b.getRawTensorData(), shape, m1)) return fail("createMaskFromTensorHIP (own tensor)");
      std::vector<float> copy(b.getRawTensorData(), b.getRawTensorData() + n);
      if (!MasksVisualizationKernels::createMaskFromTensorHIP(copy.data(), shape, m2)) return fail("createMaskFromTensorHIP (host tensor)");
      if (m1.rows != 320 || m1.cols != 640 || std::memcmp(m1.data, m2.data, 320 * 640) != 0) return fail("createMaskFromTensorHIP paths disagree");
      f.write(reinterpret_cast<const char *>(m1.data), 320 * 640);
      cv::Mat blended;
      if (b.visualizeMask("scene", blended, cv::Size(640, 360))) return fail("visualizeMask must refuse a size that is not the last frame's");
      if (!b.visualizeMask(kind == "domain" ? "domain" : "scene", blended, frame.size())) return fail("visualizeMask");
    }
    cv::Mat bad(10, 10, CV_8UC1);
    if (b.doInference(bad)) return fail("doInference must reject a non-BGR8 image with false");
  }
  std::puts("adapter_check: frame OK");
  return 0;
}
