// Minimal stand-in for <opencv2/opencv.hpp> used ONLY to compile-check the adapters in this image (OpenCV is not
// installed here).  It implements the handful of cv::Mat / cv::Size members the adapters touch; real builds use OpenCV.
#pragma once
#include <cstddef>
#include <cstdint>
#include <vector>

#define CV_8UC1 0
#define CV_8UC3 16
#define CV_32FC1 5

namespace cv
{
struct Size
{
  int width = 0, height = 0;
  Size() = default;
  Size(int w, int h) : width(w), height(h) {}
};
struct Point
{
  int x = 0, y = 0;
  Point() = default;
  Point(int x_, int y_) : x(x_), y(y_) {}
};
struct Rect
{
  int x = 0, y = 0, width = 0, height = 0;
};
class Mat
{
public:
  int rows = 0, cols = 0;
  uint8_t * data = nullptr;
  size_t step = 0;
  Mat() = default;
  Mat(int r, int c, int type) { create(r, c, type); }
  Mat(int r, int c, int type, void * ext, size_t step_ = 0) : rows(r), cols(c), data(static_cast<uint8_t *>(ext)), type_(type)
  {
    step = step_ ? step_ : static_cast<size_t>(c) * elem(type);
  }
  void create(int r, int c, int type)
  {
    rows = r;
    cols = c;
    type_ = type;
    step = static_cast<size_t>(c) * elem(type);
    store_.assign(static_cast<size_t>(r) * step, 0);
    data = store_.data();
  }
  void create(Size s, int type) { create(s.height, s.width, type); }
  bool empty() const { return data == nullptr || rows == 0 || cols == 0; }
  int type() const { return type_; }
  bool isContinuous() const { return step == static_cast<size_t>(cols) * elem(type_); }
  Size size() const { return Size(cols, rows); }

private:
  static size_t elem(int type) { return type == CV_8UC3 ? 3 : (type == CV_32FC1 ? 4 : 1); }
  int type_ = 0;
  std::vector<uint8_t> store_;
};
}  // namespace cv
