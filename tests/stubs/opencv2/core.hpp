// TEST DOUBLE - NOT OpenCV.  The handful of declarations pose_estimation_amd/host/opencv_adapter.hpp touches (cv::Mat's
// rows / cols / step[0] / type() / ptr<float>(), cv::Matx44d's operator()(r, c), CV_Assert, CV_32FC1), with the layout
// semantics of the real types, so that the adapter - which can only be compiled where OpenCV exists, and OpenCV exists
// neither in the build container nor on the GPU box - is at least compiled and run by tests/test_opencv_adapter*.py.
// Nothing under pose_estimation_amd/ includes this file; it proves nothing about OpenCV itself.
#pragma once
#include <cstddef>
#include <cstdlib>
#include <cstdio>
#include <vector>

#define CV_32FC1 5
#define CV_Assert(expr) do { if (!(expr)) { std::fprintf(stderr, "CV_Assert failed: %s\n", #expr); std::abort(); } } while (0)

namespace cv {

class Mat {  // a dense row-major float matrix that owns its rows
 public:
  struct Step {
    std::size_t bytes[2];
    std::size_t operator[](int i) const { return bytes[i]; }
  };
  int rows = 0, cols = 0;
  Step step{{0, 4}};
  Mat() = default;
  Mat(int r, int c, int /*type*/) : rows(r), cols(c), data_(static_cast<std::size_t>(r) * c) { step.bytes[0] = sizeof(float) * c; }
  int type() const { return CV_32FC1; }
  template <typename T> T* ptr(int row = 0) { return reinterpret_cast<T*>(data_.data() + static_cast<std::size_t>(row) * cols); }
  template <typename T> const T* ptr(int row = 0) const { return reinterpret_cast<const T*>(data_.data() + static_cast<std::size_t>(row) * cols); }
  template <typename T> T& at(int r, int c) { return ptr<T>(r)[c]; }

 private:
  std::vector<float> data_;
};

template <typename T, int M, int N>
struct Matx {
  T val[M * N] = {};
  T& operator()(int r, int c) { return val[r * N + c]; }
  const T& operator()(int r, int c) const { return val[r * N + c]; }
};
using Matx44d = Matx<double, 4, 4>;

}  // namespace cv
