// TEST HARNESS (tests/ only): exposes the product's host LM (csrc/sba_lm.hpp, header-only, no HIP)
// with a caller-supplied evaluator so that the host logic and the sharded all-reduce path can be
// exercised on CPU (evaluator = oracle sweep of the local shard + gloo all-reduce of the pack).
#include "../../spherical_bundle_adjuster_amd/csrc/sba_lm.hpp"

typedef int (*harness_eval_cb)(const double* rot, const double* tran, double* pack24, void* user);

extern "C" int harness_lm_solve(int mode, double* rot, double* tran, const sba_lm_options* opt,
                                harness_eval_cb cb, void* user, sba_lm_summary* summary) {
  auto evaluate = [&](const double r[3], const double t[3], sba_normal_eq* ne) -> bool {
    double pack[SBA_PACK_SIZE];
    if (cb(r, t, pack, user) != 0) return false;
    sba::expand_pack(mode, pack, ne);
    return true;
  };
  return sba::lm_solve(mode, rot, tran, *opt, evaluate, summary);
}

extern "C" void harness_default_options(sba_lm_options* o) { sba::lm_default_options(o); }

// ---- projected Armijo line search of the d-only stage (csrc/sba_line_search.hpp, header-only) ----------------
#include "../../spherical_bundle_adjuster_amd/csrc/sba_line_search.hpp"

typedef int (*harness_phi_cb)(double step_size, double* value, double* slope, void* user);

// Drives ArmijoSearch with phi(a) supplied by the caller; out3 = {success, step size, contractions}.
extern "C" int harness_armijo(const sba_lm_options* opt, double cost0, double slope0, double direction_max_norm,
                              harness_phi_cb phi, void* user, double* out3) {
  sba::ls::ArmijoSearch s;
  s.start(*opt, cost0, slope0, direction_max_norm);
  while (!s.done()) {
    double v = 0, g = 0;
    if (phi(s.query(), &v, &g, user) != 0) return -1;
    s.feed(v, g);
  }
  out3[0] = s.success() ? 1.0 : 0.0;
  out3[1] = s.step_size();
  out3[2] = s.num_iterations();
  return 0;
}

// samples: k rows (x, value, slope)
extern "C" double harness_hermite_argmin(const double* samples, int k, double lo, double hi) {
  sba::ls::Sample s[3];
  for (int i = 0; i < k && i < 3; ++i) s[i] = sba::ls::Sample{samples[3 * i], samples[3 * i + 1], samples[3 * i + 2]};
  return sba::ls::hermite_argmin(s, k, lo, hi);
}
