// TEST HARNESS (tests/ only): exposes the product's host-side 8-point math (csrc/sba_epipolar.hpp).
#include "../../spherical_bundle_adjuster_amd/csrc/sba_epipolar.hpp"
using namespace sba::epi;
extern "C" {
void harness_jacobi(int n, const double* A, double* w, double* V) { jacobi_eigen(n, A, w, V); }
int harness_smallest_eigvec(int n, const double* A, double* v, double* lambda) { return smallest_eigvec(n, A, v, lambda) ? 1 : 0; }
void harness_svd3(const double* E, double* U, double* w, double* Vt) { svd3(E, U, w, Vt); }
void harness_decompose(const double* E, double* R1, double* R2, double* t) { decompose_essential(E, R1, R2, t); }
void harness_euler(const double* R, float* out) { rot_to_euler(R, out); }
void harness_trial_groups(unsigned long long seed, int trial, int count, int* out) { trial_groups(seed, trial, count, out); }
void harness_trial_groups_from(unsigned long long seed, int trial, int count, int* out, const int* items, int n_items) {
  trial_groups(seed, trial, count, out, items, n_items);
}
void harness_trial(const double* mom45, float* e1, float* e2, float* tv, int* v1, int* v2, double* E) {
  bool a, b;
  trial_from_moments(mom45, e1, e2, tv, &a, &b, E);
  *v1 = a; *v2 = b;
}
void harness_trial_rows(const double* mom45, int rows, float* e1, float* e2, float* tv, int* v1, int* v2, double* E) {
  bool a, b;
  trial_from_moments(mom45, e1, e2, tv, &a, &b, E, rows);
  *v1 = a; *v2 = b;
}
// moments [trials][45] over `rows` matches each -> R_vec_out, T_vec_out, candidates (the host half of
// sba_problem_initial_guess_reference)
int harness_guess_from_trial_moments(const double* moments, int trials, int rows, float* euler, float* tran, int* ncand) {
  const GuessResult r = initial_guess_from_trial_moments(moments, trials, rows);
  *ncand = r.num_candidates;
  for (int i = 0; i < 3; ++i) { euler[i] = r.euler[i]; tran[i] = r.tran[i]; }
  return r.picked;
}
}
