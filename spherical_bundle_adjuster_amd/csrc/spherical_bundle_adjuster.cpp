// See spherical_bundle_adjuster.hpp.  Stage order, frozen blocks, the init_d[0][0]/init_d[1][0] quirk,
// the printed lines and the log.txt row follow the reference (spherical_bundle_adjuster.cpp:183-217,
// :255-357); every numeric step goes through the C-ABI in include/sba_hip.h (HIP kernels + host LM).
#include "spherical_bundle_adjuster.hpp"

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iostream>

namespace {
constexpr double kPi = 3.14159265358979323846;
}

spherical_bundle_adjuster::~spherical_bundle_adjuster() {
  if (problem) sba_problem_destroy(problem);
}

void spherical_bundle_adjuster::set_omp(int n) {
  // The reference sets the process-global OpenMP thread count (.cpp:835-841) for its CPU loops and for
  // Ceres.  The GPU path has no host loops over matches; the one host loop worth threads is the 80 trials of the
  // initial guess, which run on that many host threads (the result does not depend on the count).
  num_proc = n;
  if (n >= 1) sba_set_host_threads(n);
  std::cout << "Number of process: " << num_proc << std::endl;
}

void spherical_bundle_adjuster::do_bundle_adjustment(const cv::Mat& im_left, const cv::Mat& im_right) {
  std::vector<cv::KeyPoint> left_key, right_key;
  int match_size = 0, total_key_num = 0;
  cv::Mat match_output;
  std::cout << "Do feature finding and matching" << std::endl;                         // .cpp:262
  if (matcher) {
    matcher(im_left, im_right, left_key, right_key, match_size, match_output, total_key_num);
  } else {
#ifdef SBA_WITH_OPENCV
    spherical_surf fm;                                                                  // .cpp:264-266, verbatim
    fm.set_omp(this->num_proc);
    fm.do_all(im_left, im_right, left_key, right_key, match_size, match_output, total_key_num);
#else
    std::cerr << "spherical_bundle_adjuster: built without OpenCV, so the reference's default matcher "
                 "(spherical_surf, SURF/FLANN) is not available: call set_matcher() or use "
                 "do_bundle_adjustment_from_matches()" << std::endl;
    return;
#endif
  }
  const int rc = do_bundle_adjustment_from_matches(left_key, right_key, match_size, im_left.cols, im_left.rows);
  if (rc != SBA_OK) std::cerr << "spherical_bundle_adjuster: " << sba_last_error() << std::endl;
}

int spherical_bundle_adjuster::do_bundle_adjustment_from_matches(const std::vector<cv::KeyPoint>& left_key,
                                                                 const std::vector<cv::KeyPoint>& right_key,
                                                                 int match_size, int im_width, int im_height) {
  if (match_size < 0 || static_cast<size_t>(match_size) > left_key.size() ||
      static_cast<size_t>(match_size) > right_key.size())
    return SBA_ERR_INVALID_ARG;
  std::cout << "Do bundle adjustment" << std::endl;                                     // .cpp:300
  // pixel -> radian -> unit vector (.cpp:271-298) runs on the device and writes the coordinate planes directly:
  // the matched key-points go to the device once, the cv::Point3d arrays of the reference are never materialised on
  // the host, and the initial guess and all three solve stages work on that one resident copy.
  if (!problem) {
    int rc0 = sba_problem_create(&problem, device, nullptr);
    if (rc0) return rc0;
  }
  int rc = sba_problem_upload_keypoints(problem, left_key.data(), right_key.data(), static_cast<size_t>(match_size),
                                        sizeof(cv::KeyPoint), im_width, im_height, nullptr, SBA_STORE_F64);
  if (rc) return rc;
  std::vector<cv::Point3d> key_point_left_rect, key_point_right_rect;   // stay empty: the data is resident
  resident_left = &key_point_left_rect;
  resident_n = match_size;

  // Initial values (.cpp:302-331).  Default: the 8-point consensus (initial_guess, .cpp:304), then
  // init_rot = -Euler(R_vec_out) and init_tran = T_vec_out exactly like .cpp:330-331 -- including the reference's
  // use of the negated Euler triple as an angle-axis vector.  set_initial_guess(false) starts from the expected
  // values given on the command line instead, the alternative the reference keeps in a comment (.cpp:328-329).
  std::vector<std::array<double, 2>> init_d(match_size);
  for (auto& d : init_d) d = {expected_d, expected_d};                                   // .cpp:325-326
  double init_rot[3] = {expected_roll / 180 * kPi, expected_pitch / 180 * kPi, expected_yaw / 180 * kPi};
  double init_tran[3] = {expected_tx, expected_ty, expected_tz};
  if (use_initial_guess) {
    std::cout << "E matrix estimation with SVD" << std::endl;                           // .cpp:127
    double euler[3], tvec[3];
    int candidates = 0;
    guess_sampling_t how = guess_sampling;
    if (const char* env = std::getenv("SBA_GUESS_SAMPLING")) {
      if (std::strcmp(env, "groups") == 0) how = GUESS_GROUPS;
      else if (std::strcmp(env, "reference") == 0) how = GUESS_REFERENCE;
    }
    if (how == GUESS_AUTO) how = (match_size >= 4 && match_size <= kReferenceSamplingMaxN) ? GUESS_REFERENCE : GUESS_GROUPS;
    if (how == GUESS_REFERENCE)   // 80 x random_array(match_size), int(match_size * 0.25) matches each: .cpp:130-141
      rc = sba_problem_initial_guess_reference(problem, 80, 0.25, euler, tvec, &candidates);
    else
      rc = sba_problem_initial_guess(problem, 80, 0.25, guess_seed, euler, tvec, &candidates);
    if (rc) return rc;
    for (int i = 0; i < 3; ++i) { init_rot[i] = -euler[i]; init_tran[i] = tvec[i]; }     // .cpp:330-331
    res.guess_candidates = candidates;
  }

  sba_lm_options options;
  sba_lm_options_default(&options);
  options.max_num_iterations = 50;   // .cpp:336
  options.verbose = 1;               // minimizer_progress_to_stdout, .cpp:337

  rc = solve_problem(options, key_point_left_rect, key_point_right_rect, init_rot, init_tran, init_d, match_size);
  resident_left = nullptr;
  resident_n = -1;
  if (rc) return rc;

  for (int i = 0; i < 3; ++i) { res.rot[i] = init_rot[i]; res.tran[i] = init_tran[i]; }
  res.match_size = match_size;
  // log.txt row, same columns as .cpp:348-354
  std::ofstream log_file(log_path, std::ios_base::app);
  log_file << expected_roll << ',' << expected_pitch << ',' << expected_yaw << ','
           << init_rot[0] / kPi * 180.0 << ',' << init_rot[1] / kPi * 180.0 << ',' << init_rot[2] / kPi * 180.0 << ','
           << init_tran[0] << ',' << init_tran[1] << ',' << init_tran[2] << ',' << match_size << std::endl;
  // write_d_circle (.cpp:227-252, :356) draws into an image: outside the accelerated path.  write_log_d (.cpp:357):
  if (!depth_log_name.empty()) write_log_d(init_d, depth_log_name);
  std::cout << "Done." << std::endl;
  return SBA_OK;
}

void spherical_bundle_adjuster::write_log_d(const std::vector<std::array<double, 2>>& init_d,
                                            const std::string& name) const {
  std::ofstream log_d_file(name + ".txt", std::ios_base::app);
  for (const auto& d : init_d) log_d_file << d[0] << ',' << d[1] << '\n';
}

int spherical_bundle_adjuster::solve_problem(sba_lm_options& opt, std::vector<cv::Point3d>& key_point_left_rect,
                                             std::vector<cv::Point3d>& key_point_right_rect, double* init_rot,
                                             double* init_tran, std::vector<std::array<double, 2>>& init_d,
                                             int match_num) {
  int rc;
  if (!problem) {
    rc = sba_problem_create(&problem, device, nullptr);
    if (rc) return rc;
  }
  // One flat upload replaces the per-match `new AutoDiffCostFunction / new HuberLoss` of the four
  // add_residual loops (.cpp:870-889, :921-945, :978-1002, :1034-1063).  When do_bundle_adjustment_from_matches
  // already made these coordinates resident, only the depths (init_d) are sent.
  if (resident_left == &key_point_left_rect && resident_n == match_num && match_num > 0) {
    rc = sba_problem_set_depths(problem, reinterpret_cast<const double*>(init_d.data()));
  } else {
    if (key_point_left_rect.size() < static_cast<size_t>(match_num) ||
        key_point_right_rect.size() < static_cast<size_t>(match_num) || init_d.size() < static_cast<size_t>(match_num))
      return SBA_ERR_INVALID_ARG;
    rc = sba_problem_upload(problem, reinterpret_cast<const double*>(key_point_left_rect.data()),
                            reinterpret_cast<const double*>(key_point_right_rect.data()),
                            match_num > 0 ? reinterpret_cast<const double*>(init_d.data()) : nullptr,
                            static_cast<size_t>(match_num), SBA_STORE_F64);
  }
  if (rc) return rc;

  auto report = [](const char* stage, const sba_lm_summary& s) {
    std::printf("%s: iterations %d, initial cost %.6e, final cost %.6e, termination %d, %.3f ms (%.3f ms in sweeps)\n",
                stage, s.num_iterations, s.initial_cost, s.final_cost, s.termination, s.seconds_total * 1e3,
                s.seconds_eval * 1e3);
  };

  // stage 1: d-only (.cpp:196-197) -- per-match bounded depth refinement, lambda = c = 1 (.cpp:1057-1058)
  if (match_num > 0) {
    rc = sba_problem_solve_depths(problem, init_rot, init_tran, 1.0, 1.0, &opt,
                                  reinterpret_cast<double*>(init_d.data()), &res.depth_stage);
    if (rc) return rc;
    report("d-only", res.depth_stage);
  }
  // stages 2 and 3 use init_d[0][0] and init_d[1][0] for EVERY match (.cpp:941-942, :998-999)
  const double d1 = match_num > 0 ? init_d[0][0] : 0.0;
  const double d2 = match_num > 1 ? init_d[1][0] : d1;

  rc = sba_problem_solve(problem, SBA_MODE_ROT, SBA_DEPTH_UNIFORM, init_rot, init_tran, d1, d2, &opt, &res.rot_stage);   // .cpp:202-203
  if (rc) return rc;
  report("rot-only", res.rot_stage);
  rc = sba_problem_solve(problem, SBA_MODE_TRAN, SBA_DEPTH_UNIFORM, init_rot, init_tran, d1, d2, &opt, &res.tran_stage); // .cpp:208-209
  if (rc) return rc;
  report("tran-only", res.tran_stage);

  std::cout << "expected rotation vector " << expected_roll << ' ' << expected_pitch << ' ' << expected_yaw << ' ' << std::endl;
  std::cout << "rotation vector in degree " << init_rot[0] / kPi * 180.0 << ' ' << init_rot[1] / kPi * 180.0 << ' '
            << init_rot[2] / kPi * 180.0 << std::endl;
  std::cout << "translation vector " << init_tran[0] << ' ' << init_tran[1] << ' ' << init_tran[2] << std::endl;
  return SBA_OK;
}
