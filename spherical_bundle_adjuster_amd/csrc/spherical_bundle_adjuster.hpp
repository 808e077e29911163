// Host-side C++ mirror of the reference's `spherical_bundle_adjuster` class
// (reference spherical_bundle_adjuster.hpp:15-58): same constructor, set_omp and do_bundle_adjustment,
// so the call sequence of main/main.cpp:29-32 compiles unchanged.  The Ceres-backed private
// `solve_problem` is replaced by calls into the C-ABI (include/sba_hip.h); nothing here touches Ceres.
//
// Out of scope on this path: SURF/FLANN matching, which stays in OpenCV on the host.
//   * With OpenCV (-DSBA_WITH_OPENCV, the reference tree on the include path): this header pulls in the reference's
//     own "spherical_surf.hpp" like the header it replaces does (reference spherical_bundle_adjuster.hpp:3), and
//     do_bundle_adjustment() default-constructs the matcher exactly as the reference does
//     (`spherical_surf fm; fm.set_omp(num_proc); fm.do_all(...)`, spherical_bundle_adjuster.cpp:264-266) --
//     main/main.cpp:29-32 compiles AND runs unchanged.
//   * Without OpenCV (this image): layout-compatible stand-in types (sba_types.hpp); a matcher can be supplied as a
//     callable with the same `do_all` signature (set_matcher), and `do_bundle_adjustment_from_matches` enters right
//     after the matcher, with the matched key-points the reference's matchers return.
#pragma once
#include <array>
#include <functional>
#include <string>
#include <vector>

#ifdef SBA_WITH_OPENCV
// what the replaced header includes (reference spherical_bundle_adjuster.hpp:3-13), minus Ceres
#include "spherical_surf.hpp"
#include "debug_print.h"
#include <fstream>
#include <sstream>
#include <omp.h>
#endif

#include "../../include/sba_hip.h"
#include "sba_types.hpp"

class spherical_bundle_adjuster {
 public:
  // do_all(im_left, im_right, left_key, right_key, match_size, match_output, total_key_num)
  // -- the signature shared by spherical_surf / equi2cube_surf / feature_matcher
  //    (reference spherical_surf.hpp:13, equi2cube_surf.hpp:13, feature_matcher.hpp:37)
  using matcher_fn = std::function<void(const cv::Mat&, const cv::Mat&, std::vector<cv::KeyPoint>&,
                                        std::vector<cv::KeyPoint>&, int&, cv::Mat&, int&)>;

  spherical_bundle_adjuster(double roll = 0, double pitch = 0, double yaw = 0, double tx = 0,
                            double ty = 0, double tz = 0, double d = 0)
      : expected_roll(roll), expected_pitch(pitch), expected_yaw(yaw), expected_tx(tx),
        expected_ty(ty), expected_tz(tz), expected_d(d) {}
  ~spherical_bundle_adjuster();

  void set_omp(int num_proc);
  void do_bundle_adjustment(const cv::Mat& im_left, const cv::Mat& im_right);

  // ---- additions of this build (not in the reference) ------------------------------------------------
  // Replace the matcher (default with OpenCV: the reference's spherical_surf; without: none).
  void set_matcher(matcher_fn fn) { matcher = std::move(fn); }
  void set_device(int hip_device) { device = hip_device; }
  void set_log_path(const std::string& path) { log_path = path; }   // default "log.txt" (.cpp:349)
  // Per-match depth log (reference write_log_d, .cpp:219-225, called with "log_d" at :357): appends one "d1,d2" line
  // per match to <name>.txt.  An empty name switches it off (10^7 matches would be a 200 MB text file).
  void set_depth_log_name(const std::string& name) { depth_log_name = name; }
  // true (default, like the reference): start from the 8-point consensus; false: from the expected values.
  void set_initial_guess(bool on, unsigned long long seed = 0) { use_initial_guess = on; guess_seed = seed; }
  // Which subsets the 80 trials of the initial guess solve from.  GUESS_AUTO (default): up to kReferenceSamplingMaxN
  // matches -- every size the reference is run at -- the reference's OWN subsets (random_array: std::random_shuffle on the
  // process's rand() stream, .hpp:182-211, drawn where the reference draws them, so this process consumes rand() exactly
  // as the reference does and the trials see the reference's matches); above that, seeded groups of matches
  // (one streaming device pass instead of 80 shuffles of all indices).  SBA_GUESS_SAMPLING=groups|reference overrides.
  enum guess_sampling_t { GUESS_AUTO = 0, GUESS_GROUPS = 1, GUESS_REFERENCE = 2 };
  static constexpr int kReferenceSamplingMaxN = 65536;
  void set_guess_sampling(guess_sampling_t s) { guess_sampling = s; }
  // Everything after the matcher: pixel -> sphere, initial values, three-stage solve, log row.
  // Returns 0 or a negative SBA_ERR_* (message via sba_last_error()).
  int do_bundle_adjustment_from_matches(const std::vector<cv::KeyPoint>& left_key,
                                        const std::vector<cv::KeyPoint>& right_key, int match_size,
                                        int im_width, int im_height);
  struct result {
    double rot[3] = {0, 0, 0};    // angle-axis, radians (init_rot after solve_problem)
    double tran[3] = {0, 0, 0};   // init_tran after solve_problem
    int match_size = 0;
    sba_lm_summary depth_stage{}, rot_stage{}, tran_stage{};
    int guess_candidates = 0;     // valid rotation candidates of the 8-point consensus
  };
  const result& last_result() const { return res; }

 private:
  // reference spherical_bundle_adjuster.hpp:37-43 with ceres::Solver::Options -> sba_lm_options
  void write_log_d(const std::vector<std::array<double, 2>>& init_d, const std::string& name) const;   // .cpp:219-225
  int solve_problem(sba_lm_options& opt, std::vector<cv::Point3d>& key_point_left_rect,
                    std::vector<cv::Point3d>& key_point_right_rect, double* init_rot, double* init_tran,
                    std::vector<std::array<double, 2>>& init_d, int match_num);

  double expected_roll, expected_pitch, expected_yaw, expected_tx, expected_ty, expected_tz, expected_d;
  int num_proc = 1;
  int device = 0;
  std::string log_path = "log.txt";
  std::string depth_log_name = "log_d";
  matcher_fn matcher;
  bool use_initial_guess = true;
  unsigned long long guess_seed = 0;
  guess_sampling_t guess_sampling = GUESS_AUTO;
  const void* resident_left = nullptr;   // coordinates currently resident in `problem`
  int resident_n = -1;
  sba_problem* problem = nullptr;
  result res;
};
