// Internal to the engine's translation units (engine.cpp = weights + plan, engine_dispatch.cpp = per-layer kernel selection and weight
// packing, engine_io.cpp = frame path, graph replay, outputs, visualisation).  Not part of the library's interface.
#pragma once
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <exception>
#include <thread>

#include "conv_epilogue.hpp"
#include "engine.hpp"

namespace vp {

static inline int round_up(int v, int m) { return (v + m - 1) / m * m; }

// VP_PLAN_OVERRIDE (engine_dispatch.cpp): one layer's kernel choice forced by name -- the in-frame tuner's lever (tools/plan_search.py)
bool plan_override(const std::string& name, int* tile, int* nsplit);

constexpr float kBnEps = 1e-5f;  // torchvision efficientnet_b0 BatchNorm2d default

struct Folded {
  std::vector<float> w, b;
  int cout = 0, cin = 0, k = 0;  // cin = per-group input channels
};
Folded fold_conv(const WeightBlob& blob, const std::string& conv, const std::string& norm, float eps);
Folded fold_conv_bn(const WeightBlob& blob, const std::string& p);    // torchvision Conv2dNormActivation: `.0` conv, `.1` norm
Folded fold_conv_norm(const WeightBlob& blob, const std::string& p);  // common_layers.py:5-14 Conv: `.conv`, `.norm`
// (hi, lo) fp16 planes of v * pre, pre = the weight row's power-of-two prescale (an exact product); throws RangeError when v itself is
// beyond the fp16 range or not finite
void split_half(float v, float pre, half_t* hi, half_t* lo);
half_t truncate_lo(half_t lo);   // experiment knob VP_WLO_KEEP_BITS (engine.cpp); identity by default
// Per-output-row power-of-two PRESCALE of a weight matrix (round 4).  Unscaled, lo = fp16(w - fp16(w)) of a typical decoder weight is an fp16
// SUBNORMAL (kaiming std 0.013 at K = 11520: |lo| <= 4e-6, spacing 6e-8) and the pair carries ~17 bits instead of 22.  With s such that the
// row maximum lands in [2^13, 2^14) every weight within 2^-10 of the row maximum keeps both planes normal, and any smaller one is still exact
// to 2^-25 absolute = 2^-38 of the row maximum; the kernels' epilogues multiply the fp32 accumulator by 2^-s (ConvGemmParams::wscale) before
// the bias, which is exact.  Returns s (0 for an all-zero row), clamped so that 2^s and 2^-s are normal floats.
int prescale_exp(float amax);
struct RowScale {
  std::vector<float> pre;    // [rows] 2^s
  std::vector<float> post;   // [rows_alloc] 2^-s, 1 in the padding
};
// rows x per (contiguous rows) -> scales; rows_alloc >= rows entries in `post`
RowScale row_prescale(const float* w, size_t rows, size_t per, size_t rows_alloc);

// ---- VP_WEIGHTS_FP8, real storage (round 4).  WeightBlob::quantize_fp8_e4m3 leaves every conv / linear weight as v = q * S with q on the OCP
// e4m3 grid and |q| <= 448 per output row.  Whatever per-row factor has been applied since (BatchNorm folding), the row's scale is recovered
// from its maximum -- the maximum element is +-448 by construction -- and each element's code by rounding v / S to the nearest grid value:
// the float rounding of the fold is ~1e-7 against a grid spacing of >= 6 %, so the recovery is exact.  The kernels then carry the CODES
// (one byte per weight in HBM) and multiply the fp32 accumulator by S in the epilogue (ConvGemmParams::wscale).
uint8_t e4m3_encode(float q);                                  // nearest OCP e4m3 code of q (|q| <= 448)
inline float fp8_row_scale(float amax) { return amax > 0.0f ? amax / 448.0f : 1.0f; }

// Weight packing is host work over every weight of the network (~40 M scattered (hi, lo) stores for a scene network: 1.1 s of vp_create on one
// thread, measured): rows [0, n) are handed to a few threads in contiguous ranges.  `fn(begin, end)` must write disjoint locations for disjoint
// rows; the first exception any range throws (a RangeError of split_half) is re-thrown on the calling thread after all have joined.
template <class Fn>
void parallel_rows(int n, Fn&& fn) {
  const unsigned hw = std::thread::hardware_concurrency();
  const int nt = std::max(1, std::min<int>({8, (int)(hw ? hw : 1), n / 16}));
  if (nt <= 1) {
    fn(0, n);
    return;
  }
  std::vector<std::thread> th;
  std::vector<std::exception_ptr> err(nt);
  for (int t = 0; t < nt; ++t)
    th.emplace_back([&, t] {
      try {
        fn((int)((long long)n * t / nt), (int)((long long)n * (t + 1) / nt));
      } catch (...) {
        err[t] = std::current_exception();
      }
    });
  for (std::thread& x : th) x.join();
  for (const std::exception_ptr& e : err)
    if (e) std::rethrow_exception(e);
}

template <class T>
T* Engine::dupload(const std::vector<T>& v) {
  T* d = static_cast<T*>(dalloc(v.size() * sizeof(T), false));
  copy_h2d(d, v.data(), v.size() * sizeof(T));
  return d;
}

template <class T>
void Engine::upload_grow(T*& d, size_t& cap_elems, const std::vector<T>& v) {
  if (v.size() > cap_elems) {
    dfree(d);
    d = static_cast<T*>(dalloc(v.size() * sizeof(T), false));
    cap_elems = v.size();
  }
  copy_h2d(d, v.data(), v.size() * sizeof(T));
}

}  // namespace vp
