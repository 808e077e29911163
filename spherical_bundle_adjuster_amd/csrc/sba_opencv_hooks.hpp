// OpenCV-typed one-liners over the C-ABI for the image / key-point maps of the reference's matchers, for a maintainer
// who keeps spherical_surf / equi2cube_surf (SURF + FLANN stay in OpenCV) but wants their remaps on the GPU.  Each
// function has the signature of the member function it stands in for; results are bit-identical to the reference's
// arithmetic (csrc/sba_maps.hip).  Only meaningful with OpenCV: compile with -DSBA_WITH_OPENCV.
//
//   equi2cube_surf.cpp:84-87   cube.get_all(im, cube_size)              -> sba_cv::get_all(im, cube_size)
//   equi2cube_surf.cpp:98-105  cube2equi_pixel(...) loop over key-points -> sba_cv::cube2equi_keypoints(key, ...)
//   spherical_surf.cpp:137-153 crop_rotated_image(pitch, im)            -> sba_cv::crop_rotated_image(pitch, im)
//   spherical_surf.cpp:181-191 rotate_keypoint(pitch, key, w, h)        -> sba_cv::rotate_keypoint(pitch, key, w, h)
#pragma once
#ifdef SBA_WITH_OPENCV
#include <iostream>
#include <vector>

#include "../../include/sba_hip.h"
#include "opencv2/core.hpp"

namespace sba_cv {

inline void report(int rc, const char* what) {
  if (rc != SBA_OK) std::cerr << what << ": " << sba_last_error() << std::endl;
}

// equi2cube::get_all (equi2cube.cpp:282-302): S x 6S strip, faces left, front, right, back, top, bottom
inline cv::Mat get_all(const cv::Mat& im, int cube_size, int device = 0) {
  cv::Mat out(cube_size, 6 * cube_size, CV_8UC3);
  report(sba_equi2cube(device, im.data, im.rows, im.cols, cube_size, out.data), "sba_equi2cube");
  return out;
}

// spherical_surf::crop_rotated_image (spherical_surf.cpp:76-108)
inline cv::Mat crop_rotated_image(float pitch_rot, const cv::Mat& im, int device = 0) {
  cv::Mat out(im.rows / 4, im.cols, im.type());
  report(sba_crop_rotated_image(device, im.data, im.rows, im.cols, pitch_rot, out.data), "sba_crop_rotated_image");
  return out;
}

// spherical_surf::rotate_keypoint (spherical_surf.cpp:110-123), in place
inline void rotate_keypoint(float pitch_rot_inv, std::vector<cv::KeyPoint>& key, int width, int height, int device = 0) {
  report(sba_rotate_keypoints(device, key.data(), key.size(), sizeof(cv::KeyPoint), pitch_rot_inv, width, height),
         "sba_rotate_keypoints");
}

// the cube2equi_pixel loop of equi2cube_surf::do_all (equi2cube_surf.cpp:19-76, :98-105), in place on key[i].pt
inline void cube2equi_keypoints(std::vector<cv::KeyPoint>& key, int cube_size, int im_width, int im_height, int device = 0) {
  report(sba_cube2equi_keypoints(device, key.data(), key.size(), sizeof(cv::KeyPoint), cube_size, im_width, im_height),
         "sba_cube2equi_keypoints");
}

}  // namespace sba_cv
#endif  // SBA_WITH_OPENCV
