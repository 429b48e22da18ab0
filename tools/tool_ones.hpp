// Developer tools: a device vector of ones for ConvGemmParams::wscale / MbFrontParams::s_exp / MbBackParams::wscale (the per-row 2^-s of the
// engine's weight prescale; the tools pack un-scaled weights).  Leaked on purpose: the tools are short-lived processes.
#pragma once
#include <hip/hip_runtime.h>

#include <vector>

inline const float* tool_dev_ones(size_t n) {
  std::vector<float> h(n, 1.0f);
  float* d = nullptr;
  (void)hipMalloc(&d, n * sizeof(float));
  (void)hipMemcpy(d, h.data(), n * sizeof(float), hipMemcpyHostToDevice);
  return d;
}
