// TEST HARNESS (tests/ only): a MINIMAL FAKE of the few OpenCV declarations that the SBA_WITH_OPENCV branch of
// csrc/spherical_bundle_adjuster.{hpp,cpp}, csrc/sba_opencv_hooks.hpp, the reference's matcher headers and
// main/main.cpp name.  Declarations only -- used with `g++ -fsyntax-only` to prove that branch parses and
// type-checks (OpenCV itself is not in this image).  Never linked, never run.
#pragma once
#include <cstdint>
#include <string>
#include <vector>
#define CV_8UC3 16
typedef long long int64;
namespace cv {
typedef std::string String;
struct Point2f { float x, y; };
struct Point3d { double x, y, z; };
template <typename T, int N> struct Vec { T val[N]; T& operator[](int i) { return val[i]; } const T& operator[](int i) const { return val[i]; } };
typedef Vec<float, 3> Vec3f;
typedef Vec<int, 2> Vec2i;
struct KeyPoint { Point2f pt; float size, angle, response; int octave, class_id; };
struct DMatch { int queryIdx, trainIdx, imgIdx; float distance; };
struct Mat {
  int rows, cols;
  unsigned char* data;
  Mat();
  Mat(int rows, int cols, int type);
  int type() const;
  bool empty() const;
};
template <typename T> struct Ptr { T* operator->() const; };
struct Feature2D {};
struct DescriptorMatcher {};
enum { IMREAD_COLOR = 1 };
Mat imread(const String& name, int flags);
int64 getTickCount();
double getTickFrequency();
}  // namespace cv
static_assert(sizeof(cv::KeyPoint) == 28 && sizeof(cv::Point3d) == 24, "layouts the C-ABI relies on");
