// Host step logic of the bounded d-only stage as a resumable state machine (no HIP in here: the same source drives the
// device passes in sba_stages.cpp and is exercised on the CPU -- also sharded over two gloo ranks -- by
// tests/test_depth_solver_cpu.py with an emulated pass).
//
// The reference's first stage (spherical_bundle_adjuster.cpp:196-197, functor :1004-1032, bounds :1060-1061) is ONE
// Ceres problem: a single trust-region radius, a single accept / reject on the total cost, global convergence tests --
// and, because the problem is bounds-constrained and max_num_line_search_step_size_iterations stays at Ceres' default of
// 20 (.cpp:334-338), a projected Armijo line search on every step (sba_line_search.hpp).  The device does all per-match
// work of a "pass" (depth_step_kernel: step delta of the damped system at the current depths, projected candidate
// P(d + alpha delta), nine global reductions); this class decides what the next pass is and what to do with its result:
//
//   request()  -> { alpha, keep_diagonal, first, radius } of the pass to run next        (valid while !done())
//   feed(out)  <- the nine reductions of that pass (DEPTH_OUT_* slots; already all-reduced on a sharded problem)
//   take_candidate() -> true exactly once after an accepted step: the caller's candidate planes become its depths
//
// Order of the checks = Ceres' TrustRegionMinimizer: after every iteration the iteration limit, the gradient tolerance,
// the minimum radius (the gradient at the current point arrives with the alpha = 1 pass); inside an iteration step
// validity (model decrease > 0, five invalid steps in a row fail), the line search, parameter tolerance, function
// tolerance, step quality.
#pragma once
#include <algorithm>
#include <cmath>

#include "../../include/sba_hip.h"
#include "sba_line_search.hpp"

namespace sba {

// Results of one pass (slots of the out / host_out arrays): seven sums and two maxima.
enum {
  DEPTH_OUT_COST = 0,        // cost at d
  DEPTH_OUT_MODEL = 1,       // model cost change of the trust-region step delta
  DEPTH_OUT_CAND_COST = 2,   // cost at the candidate P(d + alpha delta)
  DEPTH_OUT_STEP2 = 3,       // |candidate - d|^2
  DEPTH_OUT_X2 = 4,          // |d|^2
  DEPTH_OUT_GDELTA = 5,      // gradient(d) . delta                         (line search: initial slope)
  DEPTH_OUT_CAND_GDELTA = 6, // gradient(candidate) . delta                 (line search: slope at the trial point)
  DEPTH_OUT_GMAX = 7,        // max: projected gradient max-norm at d
  DEPTH_OUT_DMAX = 8,        // max: |delta|_inf
  DEPTH_OUT_SUMS = 7, DEPTH_OUT_COUNT = 9, DEPTH_ROW = 16
};

struct DepthPassRequest {
  double alpha = 1.0;          // step size of the candidate: 1 = the trust-region step, < 1 = a line-search trial
  bool keep_diagonal = false;  // use the stored LM diagonal (after a rejected step; always for line-search passes)
  bool first = true;           // first pass of the stage: compute and store the Jacobi scaling
  double radius = 0.0;
};

class DepthStageSolver {
 public:
  SBA_HD void start(const sba_lm_options& o) {
    o_ = o;
    sum_ = sba_lm_summary{};
    sum_.termination = SBA_TERM_FAILURE;
    radius_ = o.initial_trust_region_radius;
    nu_ = 2.0;
    reuse_ = false;
    first_ = true;
    invalid_ = 0;
    it_ = 0;
    phase_ = kMain;
    done_ = false;
    swap_ = false;
    rc_ = SBA_OK;
    make_request(1.0, false);
  }
  SBA_HD bool done() const { return done_; }
  SBA_HD int status() const { return rc_; }                       // SBA_OK or SBA_ERR_NUMERIC
  SBA_HD const DepthPassRequest& request() const { return rq_; }
  SBA_HD const sba_lm_summary& summary() const { return sum_; }
  SBA_HD bool take_candidate() { const bool s = swap_; swap_ = false; return s; }

  SBA_HD void feed(const double* out) {
    if (done_) return;
    sum_.num_evaluations++;
    first_ = false;
    if (phase_ == kMain) { feed_main(out); return; }
    if (phase_ == kSearch) {
      search_.feed(out[DEPTH_OUT_CAND_COST], out[DEPTH_OUT_CAND_GDELTA]);
      after_search_trial(out);
      return;
    }
    decide(out);     // kRestore: the candidate planes hold the final step again
  }

 private:
  enum Phase { kMain, kSearch, kRestore };

  SBA_HD void make_request(double alpha, bool keep_diagonal) {
    rq_.alpha = alpha;
    rq_.keep_diagonal = keep_diagonal;
    rq_.first = first_;
    rq_.radius = radius_;
  }
  SBA_HD void finish(int term, int rc = SBA_OK) {
    sum_.termination = term;
    sum_.final_cost = cost_;
    sum_.final_gradient_max_norm = gmax_;
    sum_.final_radius = radius_;
    rc_ = rc;
    done_ = true;
  }
  SBA_HD void next_iteration() { phase_ = kMain; make_request(1.0, reuse_); }

  // the alpha = 1 pass at the current depths
  SBA_HD void feed_main(const double* out) {
    SBA_NO_CONTRACT
    cost_ = out[DEPTH_OUT_COST];
    gmax_ = out[DEPTH_OUT_GMAX];
    model_ = out[DEPTH_OUT_MODEL];
    x2_ = out[DEPTH_OUT_X2];
    if (it_ == 0) {
      sum_.initial_cost = cost_;
      if (!std::isfinite(cost_)) { finish(SBA_TERM_FAILURE, SBA_ERR_NUMERIC); return; }
    }
    if (it_ >= o_.max_num_iterations) { finish(SBA_TERM_NO_CONVERGENCE); return; }
    if (gmax_ <= o_.gradient_tolerance) { finish(SBA_TERM_CONVERGENCE_GRADIENT); return; }
    if (radius_ < o_.min_trust_region_radius) { finish(SBA_TERM_MIN_RADIUS); return; }
    sum_.num_iterations = ++it_;
    if (!(model_ > 0.0)) {
      if (++invalid_ >= 5) { finish(SBA_TERM_FAILURE, SBA_ERR_NUMERIC); return; }
      radius_ /= nu_; nu_ *= 2.0; reuse_ = true;
      next_iteration();
      return;
    }
    invalid_ = 0;
    if (o_.max_num_line_search_step_size_iterations > 0) {
      // Ceres' DoLineSearch: this pass already holds the first trial (alpha = 1)
      search_.start(o_, cost_, out[DEPTH_OUT_GDELTA], out[DEPTH_OUT_DMAX]);
      search_.feed(out[DEPTH_OUT_CAND_COST], out[DEPTH_OUT_CAND_GDELTA]);
      planes_alpha_ = 1.0;
      after_search_trial(out);
      return;
    }
    decide(out);
  }
  SBA_HD void after_search_trial(const double* out) {
    SBA_NO_CONTRACT
    if (!search_.done()) {                       // a contraction: same delta, smaller step
      phase_ = kSearch;
      planes_alpha_ = search_.query();
      make_request(planes_alpha_, true);
      return;
    }
    sum_.num_line_search_steps += search_.num_iterations();
    if (search_.step_size() != planes_alpha_) {  // failed search: Ceres keeps the full step -- restore the candidate
      phase_ = kRestore;
      planes_alpha_ = search_.step_size();
      make_request(planes_alpha_, true);
      return;
    }
    decide(out);
  }
  // `out` belongs to the final candidate of this iteration
  SBA_HD void decide(const double* out) {
    SBA_NO_CONTRACT
    const double cand_cost = out[DEPTH_OUT_CAND_COST];
    if (std::sqrt(out[DEPTH_OUT_STEP2]) <= o_.parameter_tolerance * (std::sqrt(x2_) + o_.parameter_tolerance)) {
      finish(SBA_TERM_CONVERGENCE_PARAMETER);
      return;
    }
    const double change = cost_ - cand_cost;
    if (std::fabs(change) <= o_.function_tolerance * cost_) { finish(SBA_TERM_CONVERGENCE_FUNCTION); return; }
    const double quality = change / model_;
    if (quality > o_.min_relative_decrease) {
      swap_ = true;                              // the candidate planes become the current depths
      sum_.num_successful_steps++;
      const double q = 2.0 * quality - 1.0;
      radius_ = std::min(o_.max_trust_region_radius, radius_ / std::max(1.0 / 3.0, 1.0 - q * q * q));
      nu_ = 2.0; reuse_ = false;
    } else {
      radius_ /= nu_; nu_ *= 2.0; reuse_ = true;
    }
    next_iteration();
  }

  sba_lm_options o_{};
  sba_lm_summary sum_{};
  DepthPassRequest rq_;
  ls::ArmijoSearch search_;
  double radius_ = 0, nu_ = 2, cost_ = 0, gmax_ = 0, model_ = 0, x2_ = 0, planes_alpha_ = 1.0;
  bool reuse_ = false, first_ = true, done_ = false, swap_ = false;
  int invalid_ = 0, it_ = 0, rc_ = SBA_OK;
  Phase phase_ = kMain;
};

}  // namespace sba
