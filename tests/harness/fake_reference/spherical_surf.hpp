// TEST HARNESS (tests/ only): the INTERFACE the mirror class needs from the reference's matcher header when the
// reference tree itself is not on the include path (the GPU box): class name and the two member functions called at
// reference spherical_bundle_adjuster.cpp:264-266.  Declarations only, for `g++ -fsyntax-only`.
#pragma once
#include <vector>
#include "opencv2/core.hpp"
class spherical_surf {
 public:
  void set_omp(int num_proc);
  void do_all(const cv::Mat& im_left, const cv::Mat& im_right, std::vector<cv::KeyPoint>& left_key,
              std::vector<cv::KeyPoint>& right_key, int& match_size, cv::Mat& match_output, int& total_key_num);
};
