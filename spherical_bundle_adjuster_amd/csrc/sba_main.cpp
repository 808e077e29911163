// Command-line driver with the reference's argument contract (main/main.cpp:8-27):
//   sba_main <L> <R> <exp roll> <exp pitch> <exp yaw> <exp Tx> <exp Ty> <exp Tz> <exp d>
// With OpenCV (SBA_WITH_OPENCV) <L>/<R> are ERP images and a matcher must be linked in by the
// integrator.  Without OpenCV -- this image -- <L>/<R> are files of matched
// cv::KeyPoint records (28 bytes each, same count, match i = record i), preceded by a 16-byte
// header {int32 count, int32 im_width, int32 im_height, int32 reserved}.
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <iostream>
#include <thread>

#include "spherical_bundle_adjuster.hpp"

namespace {
bool read_keypoints(const char* path, std::vector<cv::KeyPoint>* out, int* w, int* h) {
  std::ifstream f(path, std::ios::binary);
  int32_t hdr[4];
  if (!f.read(reinterpret_cast<char*>(hdr), sizeof(hdr)) || hdr[0] < 0) return false;
  out->resize(static_cast<size_t>(hdr[0]));
  *w = hdr[1];
  *h = hdr[2];
  return hdr[0] == 0 || static_cast<bool>(f.read(reinterpret_cast<char*>(out->data()), sizeof(cv::KeyPoint) * out->size()));
}
}  // namespace

int main(int argc, char** argv) {
  if (argc != 10) {
    std::cout << "usage : spherical_bundle_adjuster.out <L image> <R image> <exp roll> <exp pitch> <exp yaw> "
                 "<exp Tx> <exp Ty> <exp Tz> <exp d>" << std::endl;
    return 0;   // the reference returns 0 on a usage error too (main/main.cpp:11)
  }
  spherical_bundle_adjuster sph_ba(atof(argv[3]), atof(argv[4]), atof(argv[5]), atof(argv[6]), atof(argv[7]),
                                   atof(argv[8]), atof(argv[9]));
  sph_ba.set_omp(static_cast<int>(std::max(1u, std::thread::hardware_concurrency())));   // omp_get_num_procs(), main/main.cpp:31
  // SBA_INITIAL_GUESS=0: start from the expected values on the command line instead of the 8-point consensus
  if (const char* env = std::getenv("SBA_INITIAL_GUESS")) sph_ba.set_initial_guess(env[0] != '0');
  std::vector<cv::KeyPoint> left_key, right_key;
  int w = 0, h = 0, w2 = 0, h2 = 0;
  if (!read_keypoints(argv[1], &left_key, &w, &h) || !read_keypoints(argv[2], &right_key, &w2, &h2) ||
      left_key.size() != right_key.size() || w != w2 || h != h2) {
    std::cerr << "cannot read matched key-point files" << std::endl;
    return 1;
  }
  const int rc = sph_ba.do_bundle_adjustment_from_matches(left_key, right_key, static_cast<int>(left_key.size()), w, h);
  if (rc != SBA_OK) {
    std::cerr << "error " << rc << ": " << sba_last_error() << std::endl;
    return 2;
  }
  return 0;
}
