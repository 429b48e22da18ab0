// TEST INFRASTRUCTURE (see oracle/__init__.py): a stand-in for <opencv2/opencv.hpp>, just wide enough to compile and run the reference's
// AutoSpeedOnnxEngine::preprocessAutoSpeed (autospeed/onnxruntime_engine.cpp:71-113) for its GEOMETRY and LAYOUT: the scale, the truncated
// size, the centred paste into a canvas of 114s, /255 and the plane order.  OpenCV itself is absent from this image, so the two arithmetic
// pieces are NOT OpenCV's: cv::resize here is nearest-neighbour (a placeholder the pin script mirrors), convertTo is float(u8) * float(alpha).
#pragma once
#include <algorithm>
#include <cstdint>
#include <cstring>
#include <memory>
#include <vector>

#define CV_8UC3 16
#define CV_32FC3 21
#define CV_32FC1 5

namespace cv {

enum { INTER_LINEAR = 1 };
struct Size {
  int width = 0, height = 0;
  Size() {}
  Size(int w, int h) : width(w), height(h) {}
};
struct Scalar {
  double v[4];
  Scalar(double a = 0, double b = 0, double c = 0, double d = 0) : v{a, b, c, d} {}
};
struct Rect {
  int x, y, width, height;
  Rect(int x_, int y_, int w, int h) : x(x_), y(y_), width(w), height(h) {}
};

class Mat {
 public:
  int rows = 0, cols = 0;
  uint8_t* data = nullptr;
  size_t step = 0;
  Mat() {}
  Mat(int r, int c, int type) { create(r, c, type); }
  Mat(int r, int c, int type, const Scalar& s) {
    create(r, c, type);
    for (int y = 0; y < rows; ++y)
      for (int x = 0; x < cols; ++x)
        for (int k = 0; k < channels(); ++k) {
          if (elem1() == 1) data[y * step + x * channels() + k] = (uint8_t)s.v[k];
          else reinterpret_cast<float*>(data + y * step)[x * channels() + k] = (float)s.v[k];
        }
  }
  void create(int r, int c, int type) {
    rows = r;
    cols = c;
    type_ = type;
    step = (size_t)c * channels() * elem1();
    store_ = std::make_shared<std::vector<uint8_t>>((size_t)r * step);
    data = store_->data();
  }
  int type() const { return type_; }
  int channels() const { return type_ == CV_32FC1 ? 1 : 3; }
  int elem1() const { return type_ == CV_8UC3 ? 1 : 4; }
  bool empty() const { return data == nullptr; }
  Mat operator()(const Rect& r) const {  // a view sharing the storage
    Mat m = *this;
    m.rows = r.height;
    m.cols = r.width;
    m.data = data + r.y * step + (size_t)r.x * channels() * elem1();
    return m;
  }
  void copyTo(Mat dst) const {
    for (int y = 0; y < rows; ++y) std::memcpy(dst.data + y * dst.step, data + y * step, (size_t)cols * channels() * elem1());
  }
  void convertTo(Mat& dst, int type, double alpha) const {  // 8UC3 -> 32FC3 only
    dst.create(rows, cols, type);
    const float a = (float)alpha;
    for (int y = 0; y < rows; ++y)
      for (int x = 0; x < cols * 3; ++x) reinterpret_cast<float*>(dst.data + y * dst.step)[x] = (float)data[y * step + x] * a;
  }

 private:
  int type_ = CV_8UC3;
  std::shared_ptr<std::vector<uint8_t>> store_;
};

inline void resize(const Mat& src, Mat& dst, Size sz, double, double, int) {  // PLACEHOLDER: nearest neighbour (see the header comment)
  dst.create(sz.height, sz.width, CV_8UC3);
  for (int y = 0; y < sz.height; ++y)
    for (int x = 0; x < sz.width; ++x) {
      const int sy = std::min(src.rows - 1, (int)((long long)y * src.rows / sz.height)), sx = std::min(src.cols - 1, (int)((long long)x * src.cols / sz.width));
      std::memcpy(dst.data + y * dst.step + x * 3, src.data + sy * src.step + sx * 3, 3);
    }
}
inline void split(const Mat& src, std::vector<Mat>& planes) {  // 32FC3 -> 3 x 32FC1
  planes.resize(3);
  for (int k = 0; k < 3; ++k) {
    planes[k].create(src.rows, src.cols, CV_32FC1);
    for (int y = 0; y < src.rows; ++y)
      for (int x = 0; x < src.cols; ++x)
        reinterpret_cast<float*>(planes[k].data + y * planes[k].step)[x] = reinterpret_cast<const float*>(src.data + y * src.step)[x * 3 + k];
  }
}

}  // namespace cv
