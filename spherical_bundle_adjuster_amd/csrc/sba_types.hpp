// Layout-compatible stand-ins for the few OpenCV types that cross the drop-in boundary, used ONLY when
// OpenCV is absent (it is absent in this image).  With OpenCV present, define SBA_WITH_OPENCV and the
// real headers are used instead; the layouts are identical:
//   cv::Point3d  = 3 x f64 (24 B)                       key_point_*_rect, spherical_bundle_adjuster.cpp:286-298
//   cv::KeyPoint = Point2f pt; float size, angle, response; int octave, class_id  (28 B, OpenCV 3.4)
//   cv::Mat      = only rows / cols / data of an 8UC3 image are touched on this path
#pragma once
#ifdef SBA_WITH_OPENCV
#include "opencv2/core.hpp"
#else
#include <cstdint>
#include <vector>
namespace cv {
struct Point2f { float x = 0, y = 0; };
struct Point3d { double x = 0, y = 0, z = 0; };
struct KeyPoint {
  Point2f pt;
  float size = 0, angle = -1, response = 0;
  int octave = 0, class_id = -1;
};
struct Vec3f { float v[3] = {0, 0, 0}; float& operator[](int i) { return v[i]; } float operator[](int i) const { return v[i]; } };
struct Mat {   // non-owning view of an 8UC3 image
  int rows = 0, cols = 0;
  uint8_t* data = nullptr;
  Mat() = default;
  Mat(int r, int c, uint8_t* d) : rows(r), cols(c), data(d) {}
  bool empty() const { return data == nullptr || rows == 0 || cols == 0; }
};
}  // namespace cv
static_assert(sizeof(cv::Point3d) == 24, "cv::Point3d layout");
static_assert(sizeof(cv::KeyPoint) == 28, "cv::KeyPoint layout");
#endif
