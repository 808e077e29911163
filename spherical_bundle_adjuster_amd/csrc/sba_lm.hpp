// Host-side damped (Levenberg-Marquardt) solve over the 3 / 5 / 6 free parameters.
//
// Replaces `ceres::Solve(opt, &problem_rot|problem_tran, &summary)` of the reference
// (spherical_bundle_adjuster.cpp:197-209, options :334-338).  The reference hands Ceres an
// N-block problem and lets ITERATIVE_SCHUR/CG solve the damped system; here the device sweep
// already returns the exact 6x6 normal equations, so the damped system is solved by Cholesky.
// The iterate schedule restates Ceres' TrustRegionMinimizer + LevenbergMarquardtStrategy defaults
// (documented public behaviour; Ceres is not vendored in the reference and its version is unpinned):
//   - Jacobi scaling  s_i = 1 / (1 + sqrt(H_ii)) fixed at iteration 0
//   - D^2 = clamp(diag(H_s), 1e-6, 1e32) / radius, kept across rejected steps
//   - step quality rho = (cost - cost_candidate) / model_cost_change; accept if rho > 1e-3
//   - accepted: radius /= max(1/3, 1 - (2 rho - 1)^3); rejected: radius /= nu, nu *= 2
//   - stop on parameter / function tolerance inside an iteration; after it on the iteration limit, the gradient
//     tolerance, the minimum radius (in that order, as Ceres checks them)
//   - additive update of the angle-axis vector (no manifold is set anywhere in the reference)
// One device sweep per LM iteration: the candidate point is evaluated with its Jacobian terms, so an
// accepted step needs no second sweep (and a multi-GPU run needs one all-reduce per iteration).
//
// The solver is a resumable state machine (LmSolver): it names the point it wants evaluated next
// (query_rot / query_tran) and is fed the normal equations there.  A single problem is a loop around it
// (lm_solve); the batched per-pair solve advances many solvers in lock-step off one device launch.
//
// Build extension (SBA_TRAN_SPHERE): the translation moves in the 2-dim tangent plane of the
// sphere |tran| = const (5-DoF R|t); Ceres users would get this with a local parameterization.
#pragma once
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstring>

#include "../../include/sba_hip.h"

namespace sba {

inline void lm_default_options(sba_lm_options* o) {
  o->max_num_iterations = 50;
  o->initial_trust_region_radius = 1e4;
  o->max_trust_region_radius = 1e16;
  o->min_trust_region_radius = 1e-32;
  o->min_relative_decrease = 1e-3;
  o->min_lm_diagonal = 1e-6;
  o->max_lm_diagonal = 1e32;
  o->function_tolerance = 1e-6;
  o->gradient_tolerance = 1e-10;
  o->parameter_tolerance = 1e-8;
  o->jacobi_scaling = 1;
  o->huber_delta = 1.0;
  o->tran_param = SBA_TRAN_FREE;
  o->verbose = 0;
  o->max_num_line_search_step_size_iterations = 20;
  o->line_search_sufficient_function_decrease = 1e-4;
  o->max_line_search_step_contraction = 1e-3;
  o->min_line_search_step_contraction = 0.6;
  o->min_line_search_step_size = 1e-9;
}

// pack (SBA_PACK_* layout) -> symmetric 6x6 over [rot | tran]
inline void expand_pack(int mode, const double* pack, sba_normal_eq* ne) {
  std::memset(ne, 0, sizeof(*ne));
  double* H = ne->H;
  if (mode == SBA_MODE_ROT || mode == SBA_MODE_RT) {
    int k = SBA_PACK_HAA;
    for (int a = 0; a < 3; ++a)
      for (int b = a; b < 3; ++b) {
        H[6 * a + b] = pack[k];
        H[6 * b + a] = pack[k];
        ++k;
      }
    for (int a = 0; a < 3; ++a) ne->g[a] = pack[SBA_PACK_GA + a];
  }
  if (mode == SBA_MODE_TRAN || mode == SBA_MODE_RT) {
    for (int c = 0; c < 3; ++c) {
      H[6 * (3 + c) + (3 + c)] = pack[SBA_PACK_SW];
      ne->g[3 + c] = pack[SBA_PACK_GT + c];
    }
  }
  if (mode == SBA_MODE_RT)
    for (int a = 0; a < 3; ++a)
      for (int c = 0; c < 3; ++c) {
        H[6 * a + (3 + c)] = pack[SBA_PACK_HAT + 3 * a + c];
        H[6 * (3 + c) + a] = pack[SBA_PACK_HAT + 3 * a + c];
      }
  ne->cost = pack[SBA_PACK_COST];
  ne->sum_w = pack[SBA_PACK_SW];
  ne->n_outlier = pack[SBA_PACK_NOUT];
}

namespace detail {

// Cholesky solve of the m x m SPD system A y = b (m <= 6).  Returns false if not SPD.
inline bool cholesky_solve(int m, const double* A, const double* b, double* y) {
  double L[36];
  for (int i = 0; i < m; ++i)
    for (int j = 0; j <= i; ++j) {
      double s = A[i * m + j];
      for (int k = 0; k < j; ++k) s -= L[i * m + k] * L[j * m + k];
      if (i == j) {
        if (!(s > 0.0) || !std::isfinite(s)) return false;
        L[i * m + i] = std::sqrt(s);
      } else {
        L[i * m + j] = s / L[j * m + j];
      }
    }
  double z[6];
  for (int i = 0; i < m; ++i) {
    double s = b[i];
    for (int k = 0; k < i; ++k) s -= L[i * m + k] * z[k];
    z[i] = s / L[i * m + i];
  }
  for (int i = m - 1; i >= 0; --i) {
    double s = z[i];
    for (int k = i + 1; k < m; ++k) s -= L[k * m + i] * y[k];
    y[i] = s / L[i * m + i];
  }
  return true;
}

// Orthonormal basis B (3x2, columns b0,b1) of the plane perpendicular to t.
inline void tangent_basis(const double t[3], double B[6]) {
  const double n = std::sqrt(t[0] * t[0] + t[1] * t[1] + t[2] * t[2]);
  double u[3] = {1, 0, 0};
  if (n > 0) { u[0] = t[0] / n; u[1] = t[1] / n; u[2] = t[2] / n; }
  int k = 0;   // the coordinate axis least aligned with u
  if (std::fabs(u[1]) < std::fabs(u[k])) k = 1;
  if (std::fabs(u[2]) < std::fabs(u[k])) k = 2;
  double a[3] = {0, 0, 0};
  a[k] = 1.0;
  double b0[3] = {u[1] * a[2] - u[2] * a[1], u[2] * a[0] - u[0] * a[2], u[0] * a[1] - u[1] * a[0]};
  const double n0 = std::sqrt(b0[0] * b0[0] + b0[1] * b0[1] + b0[2] * b0[2]);
  for (double& v : b0) v /= n0;
  const double b1[3] = {u[1] * b0[2] - u[2] * b0[1], u[2] * b0[0] - u[0] * b0[2],
                        u[0] * b0[1] - u[1] * b0[0]};
  for (int r = 0; r < 3; ++r) {
    B[2 * r + 0] = b0[r];
    B[2 * r + 1] = b1[r];
  }
}

// The free-parameter view of a mode: P (6 x m) maps a local step to the ambient [rot|tran] step.
struct Param {
  int m = 0;
  double P[36] = {0};  // row-major 6 x m
  bool sphere = false;
  double tnorm = 0.0;
  void build(int mode, int tran_param, const double tran[3]) {
    std::memset(P, 0, sizeof(P));
    const bool rot = mode != SBA_MODE_TRAN, tr = mode != SBA_MODE_ROT;
    sphere = tr && tran_param == SBA_TRAN_SPHERE;
    m = (rot ? 3 : 0) + (tr ? (sphere ? 2 : 3) : 0);
    int c = 0;
    if (rot) { for (int a = 0; a < 3; ++a) P[a * m + (c + a)] = 1.0; c += 3; }
    if (tr) {
      if (sphere) {
        double B[6];
        tangent_basis(tran, B);
        for (int r = 0; r < 3; ++r) { P[(3 + r) * m + c] = B[2 * r]; P[(3 + r) * m + c + 1] = B[2 * r + 1]; }
        tnorm = std::sqrt(tran[0] * tran[0] + tran[1] * tran[1] + tran[2] * tran[2]);
      } else {
        for (int a = 0; a < 3; ++a) P[(3 + a) * m + (c + a)] = 1.0;
      }
    }
  }
  // Hf = P^T H P, gf = P^T g
  void project(const sba_normal_eq& ne, double* Hf, double* gf) const {
    double HP[36];
    for (int i = 0; i < 6; ++i)
      for (int j = 0; j < m; ++j) {
        double s = 0;
        for (int k = 0; k < 6; ++k) s += ne.H[6 * i + k] * P[k * m + j];
        HP[i * m + j] = s;
      }
    for (int i = 0; i < m; ++i) {
      for (int j = 0; j < m; ++j) {
        double s = 0;
        for (int k = 0; k < 6; ++k) s += P[k * m + i] * HP[k * m + j];
        Hf[i * m + j] = s;
      }
      double s = 0;
      for (int k = 0; k < 6; ++k) s += P[k * m + i] * ne.g[k];
      gf[i] = s;
    }
  }
  // x_plus_delta
  void plus(const double rot[3], const double tran[3], const double* delta, double rot_out[3],
            double tran_out[3]) const {
    double d6[6];
    for (int i = 0; i < 6; ++i) {
      double s = 0;
      for (int j = 0; j < m; ++j) s += P[i * m + j] * delta[j];
      d6[i] = s;
    }
    for (int a = 0; a < 3; ++a) {
      rot_out[a] = rot[a] + d6[a];
      tran_out[a] = tran[a] + d6[3 + a];
    }
    if (sphere) {
      const double nn = std::sqrt(tran_out[0] * tran_out[0] + tran_out[1] * tran_out[1] +
                                  tran_out[2] * tran_out[2]);
      if (nn > 0)
        for (int a = 0; a < 3; ++a) tran_out[a] *= tnorm / nn;
    }
  }
};

}  // namespace detail

class LmSolver {
 public:
  // After start() and after every feed(): either done() or (query_rot, query_tran) is the point to evaluate next.
  void start(int mode, const double rot0[3], const double tran0[3], const sba_lm_options& opt) {
    mode_ = mode;
    o_ = opt;
    for (int a = 0; a < 3; ++a) { rot_[a] = qrot_[a] = rot0[a]; tran_[a] = qtran_[a] = tran0[a]; }
    std::memset(&sum_, 0, sizeof(sum_));
    sum_.termination = SBA_TERM_FAILURE;
    phase_ = kInitial;
    done_ = false;
    rc_ = SBA_OK;
    radius_ = o_.initial_trust_region_radius;
    decrease_ = 2.0;
    reuse_diagonal_ = false;
    invalid_steps_ = 0;
    iter_ = 0;
    gmax_ = 0.0;
    t_start_ = std::chrono::steady_clock::now();
  }
  bool done() const { return done_; }
  int status() const { return rc_; }                 // SBA_OK or SBA_ERR_NUMERIC
  const double* query_rot() const { return qrot_; }
  const double* query_tran() const { return qtran_; }
  const double* rot() const { return rot_; }         // current accepted point (the result once done)
  const double* tran() const { return tran_; }
  const sba_lm_summary& summary() const { return sum_; }

  // The evaluation at the query point failed on the device side.
  void fail() { finish(SBA_TERM_FAILURE, SBA_ERR_NUMERIC); }

  // Normal equations at (query_rot, query_tran).
  void feed(const sba_normal_eq& ne) {
    using namespace detail;
    if (done_) return;
    sum_.num_evaluations++;
    if (phase_ == kInitial) {
      cur_ = ne;
      if (!std::isfinite(cur_.cost)) { finish(SBA_TERM_FAILURE, SBA_ERR_NUMERIC); return; }
      sum_.initial_cost = cur_.cost;
      par_.build(mode_, o_.tran_param, tran_);
      par_.project(cur_, Hf_, gf_);
      for (int i = 0; i < par_.m; ++i)
        scale_[i] = o_.jacobi_scaling ? 1.0 / (1.0 + std::sqrt(std::max(Hf_[i * par_.m + i], 0.0))) : 1.0;
      gmax_ = gmax_of(gf_);
      if (o_.verbose)
        std::printf("iter      cost      cost_change  |gradient|   |step|    tr_ratio  tr_radius\n"
                    "%4d % .6e    0.00e+00    %.2e   0.00e+00   0.00e+00  %.2e\n", 0, cur_.cost, gmax_, radius_);
      phase_ = kCandidate;
      next_candidate();
      return;
    }
    // phase_ == kCandidate: ne is the evaluation at the candidate
    const bool rot_free = mode_ != SBA_MODE_TRAN, tran_free = mode_ != SBA_MODE_ROT;
    double step2 = 0, x2 = 0;
    for (int a = 0; a < 3; ++a) {
      if (rot_free) { step2 += (qrot_[a] - rot_[a]) * (qrot_[a] - rot_[a]); x2 += rot_[a] * rot_[a]; }
      if (tran_free) { step2 += (qtran_[a] - tran_[a]) * (qtran_[a] - tran_[a]); x2 += tran_[a] * tran_[a]; }
    }
    const double step_norm = std::sqrt(step2), x_norm = std::sqrt(x2);
    const double cost_change = cur_.cost - ne.cost;
    const double rho = std::isfinite(ne.cost) ? cost_change / model_change_ : -1.0;
    if (o_.verbose)
      std::printf("%4d % .6e   % .2e    %.2e   %.2e  % .2e  %.2e\n", iter_, ne.cost, cost_change, gmax_, step_norm,
                  rho, radius_);
    if (step_norm <= o_.parameter_tolerance * (x_norm + o_.parameter_tolerance)) {
      finish(SBA_TERM_CONVERGENCE_PARAMETER, SBA_OK);
      return;
    }
    if (std::isfinite(ne.cost) && std::fabs(cost_change) <= o_.function_tolerance * cur_.cost) {
      finish(SBA_TERM_CONVERGENCE_FUNCTION, SBA_OK);
      return;
    }
    if (rho > o_.min_relative_decrease) {
      for (int a = 0; a < 3; ++a) { rot_[a] = qrot_[a]; tran_[a] = qtran_[a]; }
      cur_ = ne;
      sum_.num_successful_steps++;
      par_.build(mode_, o_.tran_param, tran_);
      par_.project(cur_, Hf_, gf_);
      gmax_ = gmax_of(gf_);
      const double t3 = 2.0 * rho - 1.0;
      radius_ = std::min(o_.max_trust_region_radius, radius_ / std::max(1.0 / 3.0, 1.0 - t3 * t3 * t3));
      decrease_ = 2.0;
      reuse_diagonal_ = false;
    } else {
      radius_ /= decrease_; decrease_ *= 2.0; reuse_diagonal_ = true;
    }
    next_candidate();
  }

 private:
  enum Phase { kInitial, kCandidate };

  double gmax_of(const double* g) const {
    double v = 0;
    for (int i = 0; i < par_.m; ++i) v = std::max(v, std::fabs(g[i]));
    return v;
  }
  void finish(int term, int rc) {
    sum_.termination = term;
    sum_.final_cost = cur_.cost;
    sum_.final_gradient_max_norm = gmax_;
    sum_.final_radius = radius_;
    sum_.seconds_total = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_start_).count();
    for (int a = 0; a < 3; ++a) { qrot_[a] = rot_[a]; qtran_[a] = tran_[a]; }
    rc_ = rc;
    done_ = true;
  }
  // Trust-region step(s) from the current point until one is valid: sets the query point, or terminates.
  void next_candidate() {
    using namespace detail;
    const int m = par_.m;
    for (;;) {
      // Ceres' end-of-iteration checks (FinalizeIterationAndCheckIfMinimizerCanContinue), in its order: iteration
      // limit, gradient tolerance, minimum radius -- after iteration 0 and after every (valid or invalid) step.
      if (iter_ >= o_.max_num_iterations) { finish(SBA_TERM_NO_CONVERGENCE, SBA_OK); return; }
      if (gmax_ <= o_.gradient_tolerance) { finish(SBA_TERM_CONVERGENCE_GRADIENT, SBA_OK); return; }
      if (radius_ < o_.min_trust_region_radius) { finish(SBA_TERM_MIN_RADIUS, SBA_OK); return; }
      sum_.num_iterations = ++iter_;
      double Hs[36], gs[6], A[36], rhs[6], y[6];
      for (int i = 0; i < m; ++i) {
        gs[i] = scale_[i] * gf_[i];
        for (int j = 0; j < m; ++j) Hs[i * m + j] = scale_[i] * Hf_[i * m + j] * scale_[j];
      }
      if (!reuse_diagonal_)
        for (int i = 0; i < m; ++i)
          diag_[i] = std::min(std::max(Hs[i * m + i], o_.min_lm_diagonal), o_.max_lm_diagonal);
      std::memcpy(A, Hs, sizeof(double) * m * m);
      for (int i = 0; i < m; ++i) { A[i * m + i] += diag_[i] / radius_; rhs[i] = -gs[i]; }
      bool valid = cholesky_solve(m, A, rhs, y);
      if (valid) {
        // -(J y)^T (f + J y / 2) = -g^T y - y^T H y / 2
        double gy = 0, yHy = 0;
        for (int i = 0; i < m; ++i) {
          gy += gs[i] * y[i];
          double s = 0;
          for (int j = 0; j < m; ++j) s += Hs[i * m + j] * y[j];
          yHy += y[i] * s;
        }
        model_change_ = -gy - 0.5 * yHy;
        valid = model_change_ > 0.0;
      }
      if (!valid) {
        if (++invalid_steps_ >= 5) { finish(SBA_TERM_FAILURE, SBA_ERR_NUMERIC); return; }
        radius_ /= decrease_; decrease_ *= 2.0; reuse_diagonal_ = true;
        continue;
      }
      invalid_steps_ = 0;
      double delta[6];
      for (int i = 0; i < m; ++i) delta[i] = scale_[i] * y[i];
      par_.plus(rot_, tran_, delta, qrot_, qtran_);
      return;
    }
  }

  int mode_ = 0;
  sba_lm_options o_{};
  sba_lm_summary sum_{};
  detail::Param par_;
  sba_normal_eq cur_{};
  double rot_[3] = {0, 0, 0}, tran_[3] = {0, 0, 0}, qrot_[3] = {0, 0, 0}, qtran_[3] = {0, 0, 0};
  double Hf_[36] = {0}, gf_[6] = {0}, scale_[6] = {0}, diag_[6] = {0};
  double radius_ = 0, decrease_ = 2, model_change_ = 0, gmax_ = 0;
  bool reuse_diagonal_ = false, done_ = false;
  int invalid_steps_ = 0, iter_ = 0, rc_ = SBA_OK;
  Phase phase_ = kInitial;
  std::chrono::steady_clock::time_point t_start_;
};

// Evaluator: bool(const double rot[3], const double tran[3], sba_normal_eq* out)
template <typename Evaluator>
int lm_solve(int mode, double rot[3], double tran[3], const sba_lm_options& o, Evaluator&& evaluate,
             sba_lm_summary* sum) {
  LmSolver s;
  s.start(mode, rot, tran, o);
  while (!s.done()) {
    sba_normal_eq ne;
    if (evaluate(s.query_rot(), s.query_tran(), &ne)) s.feed(ne); else s.fail();
  }
  for (int a = 0; a < 3; ++a) { rot[a] = s.rot()[a]; tran[a] = s.tran()[a]; }
  *sum = s.summary();
  return s.status();
}

}  // namespace sba
