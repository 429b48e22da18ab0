// Internal to the engine's translation units (engine.cpp = weights + plan, engine_dispatch.cpp = per-layer kernel selection and weight
// packing, engine_io.cpp = frame path, graph replay, outputs, visualisation).  Not part of the library's interface.
#pragma once
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdlib>
#include <cstring>

#include "conv_epilogue.hpp"
#include "engine.hpp"

namespace vp {

static inline int round_up(int v, int m) { return (v + m - 1) / m * m; }

constexpr float kBnEps = 1e-5f;  // torchvision efficientnet_b0 BatchNorm2d default

struct Folded {
  std::vector<float> w, b;
  int cout = 0, cin = 0, k = 0;  // cin = per-group input channels
};
Folded fold_conv(const WeightBlob& blob, const std::string& conv, const std::string& norm, float eps);
Folded fold_conv_bn(const WeightBlob& blob, const std::string& p);    // torchvision Conv2dNormActivation: `.0` conv, `.1` norm
Folded fold_conv_norm(const WeightBlob& blob, const std::string& p);  // common_layers.py:5-14 Conv: `.conv`, `.norm`
void split_half(float v, half_t* hi, half_t* lo);                     // throws RangeError beyond the fp16 range

template <class T>
T* Engine::dupload(const std::vector<T>& v) {
  T* d = static_cast<T*>(dalloc(v.size() * sizeof(T), false));
  VP_HIP_CHECK(hipMemcpy(d, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice));
  return d;
}

}  // namespace vp
