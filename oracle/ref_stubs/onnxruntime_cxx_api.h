// TEST INFRASTRUCTURE (see oracle/__init__.py): a stand-in for <onnxruntime_cxx_api.h>, just wide enough to COMPILE the reference's
// autospeed/onnxruntime_engine.cpp where it lies under /root/reference (oracle/Makefile -> oracle/_ref/autospeed_ref), so that its own
// postProcess / computeIoU / applyNMS and its letterbox geometry can be executed and pinned.  No ONNX Runtime behaviour is modelled: the
// "session" returns the tensor the harness put there.
#pragma once
#include <cstdint>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

enum { OrtArenaAllocator = 0, OrtMemTypeDefault = 0 };

namespace Ort {

struct Exception : std::runtime_error {
  using std::runtime_error::runtime_error;
};
struct MemoryInfo {
  static MemoryInfo CreateCpu(int, int) { return {}; }
};
struct AllocatorWithDefaultOptions {};
struct RunOptions {
  explicit RunOptions(std::nullptr_t) {}
};
struct TensorTypeAndShapeInfo {
  std::vector<int64_t> shape;
  std::vector<int64_t> GetShape() const { return shape; }
};
struct TypeInfo {
  std::vector<int64_t> shape;
  TensorTypeAndShapeInfo GetTensorTypeAndShapeInfo() const { return {shape}; }
};
struct Value {
  std::vector<float> data;
  std::vector<int64_t> shape;
  template <class T>
  static Value CreateTensor(const MemoryInfo&, T*, size_t, const int64_t*, size_t) { return {}; }
  template <class T>
  const T* GetTensorData() const { return data.data(); }
  TensorTypeAndShapeInfo GetTensorTypeAndShapeInfo() const { return {shape}; }
};
struct AllocatedString {
  std::string s;
  const char* get() const { return s.c_str(); }
};
struct Session {
  std::vector<int64_t> in_shape{1, 3, 640, 640}, out_shape{1, -1, -1};
  Value next_output;  // what Run() hands back
  AllocatedString GetInputNameAllocated(size_t, AllocatorWithDefaultOptions&) const { return {"images"}; }
  AllocatedString GetOutputNameAllocated(size_t, AllocatorWithDefaultOptions&) const { return {"output0"}; }
  TypeInfo GetInputTypeInfo(size_t) const { return {in_shape}; }
  TypeInfo GetOutputTypeInfo(size_t) const { return {out_shape}; }
  std::vector<Value> Run(const RunOptions&, const char* const*, const Value*, size_t, const char* const*, size_t) { return {next_output}; }
};
struct Env {};

}  // namespace Ort
