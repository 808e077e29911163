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
