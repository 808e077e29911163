// TEST HARNESS (tests/ only): exposes the product's d-only step logic (csrc/sba_depth_solver.hpp, header-only, no HIP) so
// that it can be driven on the CPU with an emulated device pass -- also sharded over gloo ranks.
#include "../../spherical_bundle_adjuster_amd/csrc/sba_depth_solver.hpp"
#include "../../spherical_bundle_adjuster_amd/csrc/sba_lm.hpp"

extern "C" {
void* depth_harness_create(const sba_lm_options* opt) {
  sba::DepthStageSolver* s = new sba::DepthStageSolver();
  s->start(*opt);
  return s;
}
void depth_harness_destroy(void* h) { delete static_cast<sba::DepthStageSolver*>(h); }
int depth_harness_done(void* h) { return static_cast<sba::DepthStageSolver*>(h)->done() ? 1 : 0; }
int depth_harness_status(void* h) { return static_cast<sba::DepthStageSolver*>(h)->status(); }
// out4: alpha, keep_diagonal, first, radius
void depth_harness_request(void* h, double* out4) {
  const sba::DepthPassRequest& r = static_cast<sba::DepthStageSolver*>(h)->request();
  out4[0] = r.alpha; out4[1] = r.keep_diagonal ? 1.0 : 0.0; out4[2] = r.first ? 1.0 : 0.0; out4[3] = r.radius;
}
void depth_harness_feed(void* h, const double* out9) { static_cast<sba::DepthStageSolver*>(h)->feed(out9); }
int depth_harness_take_candidate(void* h) { return static_cast<sba::DepthStageSolver*>(h)->take_candidate() ? 1 : 0; }
void depth_harness_summary(void* h, sba_lm_summary* s) { *s = static_cast<sba::DepthStageSolver*>(h)->summary(); }
void depth_harness_default_options(sba_lm_options* o) { sba::lm_default_options(o); }
}
