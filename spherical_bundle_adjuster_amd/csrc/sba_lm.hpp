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
//
// The whole solver is __host__ __device__: the host drives one problem (or a lock-step batch) with it, and the batched
// per-pair kernel (sba_batch_kernels.hip: batch_lm_kernel) runs the very same source on the device, one solver per pair,
// so that an entire per-pair solve needs no host round trip.  FP contraction is switched off for this header so that both
// compilations round identically (the host has no FMA contraction to begin with).
#pragma once
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstring>

#include "../../include/sba_hip.h"

#ifndef SBA_HD
#if defined(__HIPCC__)
#define SBA_HD __host__ __device__
#else
#define SBA_HD
#endif
#endif
#if defined(__clang__)
#pragma STDC FP_CONTRACT OFF
#define SBA_UNROLL _Pragma("unroll")
#elif defined(__GNUC__)
#define SBA_UNROLL _Pragma("GCC unroll 8")
#else
#define SBA_UNROLL
#endif

namespace sba {

SBA_HD inline void lm_default_options(sba_lm_options* o) {
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
SBA_HD inline void expand_pack(int mode, const double* pack, sba_normal_eq* ne) {
  *ne = sba_normal_eq{};
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

// All small matrices of the solver are stored with a FIXED row stride of 6 and every loop runs over the full 0..5 range
// under an `index < m` guard, fully unrolled: indices are then compile-time constants and the arrays live in registers
// -- on the device a run-time stride would put them in scratch memory, and the one thread that advances a pair's solver
// (batch_lm_kernel) would spend its time on memory round trips.  Only the order of the arithmetic matters for the
// result, and that is the plain i, j, k order of the textbook loops.
constexpr int kDim = 6;

// Cholesky solve of the m x m SPD system A y = b (m <= 6, row stride 6).  Returns false if not SPD.
SBA_HD inline bool cholesky_solve(int m, const double* A, const double* b, double* y) {
  double L[kDim * kDim];
  bool ok = true;
  SBA_UNROLL
  for (int i = 0; i < kDim; ++i) {
    SBA_UNROLL
    for (int j = 0; j < kDim; ++j) {
      if (j <= i && i < m && ok) {
        double s = A[i * kDim + j];
        SBA_UNROLL
        for (int k = 0; k < kDim; ++k)
          if (k < j) s -= L[i * kDim + k] * L[j * kDim + k];
        if (i == j) {
          if (!(s > 0.0) || !std::isfinite(s)) ok = false;
          else L[i * kDim + i] = std::sqrt(s);
        } else {
          L[i * kDim + j] = s / L[j * kDim + j];
        }
      }
    }
  }
  if (!ok) return false;
  double z[kDim];
  SBA_UNROLL
  for (int i = 0; i < kDim; ++i) {
    if (i < m) {
      double s = b[i];
      SBA_UNROLL
      for (int k = 0; k < kDim; ++k)
        if (k < i) s -= L[i * kDim + k] * z[k];
      z[i] = s / L[i * kDim + i];
    }
  }
  SBA_UNROLL
  for (int i = kDim - 1; i >= 0; --i) {
    if (i < m) {
      double s = z[i];
      SBA_UNROLL
      for (int k = 0; k < kDim; ++k)
        if (k > i && k < m) s -= L[k * kDim + i] * y[k];
      y[i] = s / L[i * kDim + i];
    }
  }
  return true;
}

// Orthonormal basis B (3x2, columns b0,b1) of the plane perpendicular to t.
SBA_HD inline void tangent_basis(const double t[3], double B[6]) {
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

// The free-parameter view of a mode: P (6 x m, row stride 6) maps a local step to the ambient [rot|tran] step.
struct Param {
  int m = 0;
  double P[kDim * kDim] = {0};  // row-major, row stride 6, columns 0..m-1 used
  bool sphere = false;
  double tnorm = 0.0;
  SBA_HD void build(int mode, int tran_param, const double tran[3]) {
    SBA_UNROLL
    for (int i = 0; i < kDim * kDim; ++i) P[i] = 0.0;
    const bool rot = mode != SBA_MODE_TRAN, tr = mode != SBA_MODE_ROT;
    sphere = tr && tran_param == SBA_TRAN_SPHERE;
    m = (rot ? 3 : 0) + (tr ? (sphere ? 2 : 3) : 0);
    const int c = rot ? 3 : 0;     // first translation column
    if (rot) { P[0 * kDim + 0] = 1.0; P[1 * kDim + 1] = 1.0; P[2 * kDim + 2] = 1.0; }
    if (tr) {
      if (sphere) {
        double B[6];
        tangent_basis(tran, B);
        SBA_UNROLL
        for (int r = 0; r < 3; ++r) {
          if (c == 3) { P[(3 + r) * kDim + 3] = B[2 * r]; P[(3 + r) * kDim + 4] = B[2 * r + 1]; }
          else { P[(3 + r) * kDim + 0] = B[2 * r]; P[(3 + r) * kDim + 1] = B[2 * r + 1]; }
        }
        tnorm = std::sqrt(tran[0] * tran[0] + tran[1] * tran[1] + tran[2] * tran[2]);
      } else {
        SBA_UNROLL
        for (int a = 0; a < 3; ++a) {
          if (c == 3) P[(3 + a) * kDim + 3 + a] = 1.0;
          else P[(3 + a) * kDim + a] = 1.0;
        }
      }
    }
  }
  // Hf = P^T H P, gf = P^T g   (Hf: row stride 6)
  SBA_HD void project(const sba_normal_eq& ne, double* Hf, double* gf) const {
    double HP[kDim * kDim];
    SBA_UNROLL
    for (int i = 0; i < kDim; ++i) {
      SBA_UNROLL
      for (int j = 0; j < kDim; ++j) {
        if (j < m) {
          double s = 0;
          SBA_UNROLL
          for (int k = 0; k < kDim; ++k) s += ne.H[6 * i + k] * P[k * kDim + j];
          HP[i * kDim + j] = s;
        }
      }
    }
    SBA_UNROLL
    for (int i = 0; i < kDim; ++i) {
      if (i < m) {
        SBA_UNROLL
        for (int j = 0; j < kDim; ++j) {
          if (j < m) {
            double s = 0;
            SBA_UNROLL
            for (int k = 0; k < kDim; ++k) s += P[k * kDim + i] * HP[k * kDim + j];
            Hf[i * kDim + j] = s;
          }
        }
        double s = 0;
        SBA_UNROLL
        for (int k = 0; k < kDim; ++k) s += P[k * kDim + i] * ne.g[k];
        gf[i] = s;
      }
    }
  }
  // x_plus_delta
  SBA_HD void plus(const double rot[3], const double tran[3], const double* delta, double rot_out[3],
                   double tran_out[3]) const {
    double d6[kDim];
    SBA_UNROLL
    for (int i = 0; i < kDim; ++i) {
      double s = 0;
      SBA_UNROLL
      for (int j = 0; j < kDim; ++j)
        if (j < m) s += P[i * kDim + j] * delta[j];
      d6[i] = s;
    }
    SBA_UNROLL
    for (int a = 0; a < 3; ++a) {
      rot_out[a] = rot[a] + d6[a];
      tran_out[a] = tran[a] + d6[3 + a];
    }
    if (sphere) {
      const double nn = std::sqrt(tran_out[0] * tran_out[0] + tran_out[1] * tran_out[1] +
                                  tran_out[2] * tran_out[2]);
      if (nn > 0) {
        SBA_UNROLL
        for (int a = 0; a < 3; ++a) tran_out[a] *= tnorm / nn;
      }
    }
  }
};

}  // namespace detail

class LmSolver {
 public:
  // After start() and after every feed(): either done() or (query_rot, query_tran) is the point to evaluate next.
  SBA_HD void start(int mode, const double rot0[3], const double tran0[3], const sba_lm_options& opt) {
    mode_ = mode;
    o_ = opt;
    for (int a = 0; a < 3; ++a) { rot_[a] = qrot_[a] = rot0[a]; tran_[a] = qtran_[a] = tran0[a]; }
    sum_ = sba_lm_summary{};
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
    t_start_ = now_seconds();
  }
  SBA_HD bool done() const { return done_; }
  SBA_HD int status() const { return rc_; }                 // SBA_OK or SBA_ERR_NUMERIC
  SBA_HD const double* query_rot() const { return qrot_; }
  SBA_HD const double* query_tran() const { return qtran_; }
  SBA_HD const double* rot() const { return rot_; }         // current accepted point (the result once done)
  SBA_HD const double* tran() const { return tran_; }
  SBA_HD const sba_lm_summary& summary() const { return sum_; }

  // The evaluation at the query point failed on the device side.
  SBA_HD void fail() { finish(SBA_TERM_FAILURE, SBA_ERR_NUMERIC); }

  // Normal equations at (query_rot, query_tran).
  SBA_HD void feed(const sba_normal_eq& ne) {
    using namespace detail;
    if (done_) return;
    sum_.num_evaluations++;
    if (phase_ == kInitial) {
      cur_ = ne;
      if (!std::isfinite(cur_.cost)) { finish(SBA_TERM_FAILURE, SBA_ERR_NUMERIC); return; }
      sum_.initial_cost = cur_.cost;
      par_.build(mode_, o_.tran_param, tran_);
      par_.project(cur_, Hf_, gf_);
      SBA_UNROLL
      for (int i = 0; i < detail::kDim; ++i)
        if (i < par_.m)
          scale_[i] = o_.jacobi_scaling ? 1.0 / (1.0 + std::sqrt(std::max(Hf_[i * detail::kDim + i], 0.0))) : 1.0;
      gmax_ = gmax_of(gf_);
#if !defined(__HIP_DEVICE_COMPILE__)
      if (o_.verbose)
        std::printf("iter      cost      cost_change  |gradient|   |step|    tr_ratio  tr_radius\n"
                    "%4d % .6e    0.00e+00    %.2e   0.00e+00   0.00e+00  %.2e\n", 0, cur_.cost, gmax_, radius_);
#endif
      phase_ = kCandidate;
      next_candidate();
      return;
    }
    // phase_ == kCandidate: ne is the evaluation at the candidate
    const bool rot_free = mode_ != SBA_MODE_TRAN, tran_free = mode_ != SBA_MODE_ROT;
    double step2 = 0, x2 = 0;
    SBA_UNROLL
    for (int a = 0; a < 3; ++a) {
      if (rot_free) { step2 += (qrot_[a] - rot_[a]) * (qrot_[a] - rot_[a]); x2 += rot_[a] * rot_[a]; }
      if (tran_free) { step2 += (qtran_[a] - tran_[a]) * (qtran_[a] - tran_[a]); x2 += tran_[a] * tran_[a]; }
    }
    const double step_norm = std::sqrt(step2), x_norm = std::sqrt(x2);
    const double cost_change = cur_.cost - ne.cost;
    const double rho = std::isfinite(ne.cost) ? cost_change / model_change_ : -1.0;
#if !defined(__HIP_DEVICE_COMPILE__)
    if (o_.verbose)
      std::printf("%4d % .6e   % .2e    %.2e   %.2e  % .2e  %.2e\n", iter_, ne.cost, cost_change, gmax_, step_norm,
                  rho, radius_);
#endif
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

  // wall clock of the host; the device-side solver reports 0 (the host times the whole batched solve)
  SBA_HD static double now_seconds() {
#if defined(__HIP_DEVICE_COMPILE__)
    return 0.0;
#else
    return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
#endif
  }
  SBA_HD double gmax_of(const double* g) const {
    double v = 0;
    SBA_UNROLL
    for (int i = 0; i < detail::kDim; ++i)
      if (i < par_.m) v = std::max(v, std::fabs(g[i]));
    return v;
  }
  SBA_HD void finish(int term, int rc) {
    sum_.termination = term;
    sum_.final_cost = cur_.cost;
    sum_.final_gradient_max_norm = gmax_;
    sum_.final_radius = radius_;
    sum_.seconds_total = now_seconds() - t_start_;
    for (int a = 0; a < 3; ++a) { qrot_[a] = rot_[a]; qtran_[a] = tran_[a]; }
    rc_ = rc;
    done_ = true;
  }
  // Trust-region step(s) from the current point until one is valid: sets the query point, or terminates.
  SBA_HD void next_candidate() {
    using namespace detail;
    const int m = par_.m;
    for (;;) {
      // Ceres' end-of-iteration checks (FinalizeIterationAndCheckIfMinimizerCanContinue), in its order: iteration
      // limit, gradient tolerance, minimum radius -- after iteration 0 and after every (valid or invalid) step.
      if (iter_ >= o_.max_num_iterations) { finish(SBA_TERM_NO_CONVERGENCE, SBA_OK); return; }
      if (gmax_ <= o_.gradient_tolerance) { finish(SBA_TERM_CONVERGENCE_GRADIENT, SBA_OK); return; }
      if (radius_ < o_.min_trust_region_radius) { finish(SBA_TERM_MIN_RADIUS, SBA_OK); return; }
      sum_.num_iterations = ++iter_;
      double Hs[kDim * kDim], gs[kDim], A[kDim * kDim], rhs[kDim], y[kDim];
      SBA_UNROLL
      for (int i = 0; i < kDim; ++i) {
        if (i < m) {
          gs[i] = scale_[i] * gf_[i];
          SBA_UNROLL
          for (int j = 0; j < kDim; ++j)
            if (j < m) Hs[i * kDim + j] = scale_[i] * Hf_[i * kDim + j] * scale_[j];
        }
      }
      if (!reuse_diagonal_) {
        SBA_UNROLL
        for (int i = 0; i < kDim; ++i)
          if (i < m) diag_[i] = std::min(std::max(Hs[i * kDim + i], o_.min_lm_diagonal), o_.max_lm_diagonal);
      }
      SBA_UNROLL
      for (int i = 0; i < kDim; ++i) {
        SBA_UNROLL
        for (int j = 0; j < kDim; ++j)
          if (i < m && j < m) A[i * kDim + j] = Hs[i * kDim + j];
      }
      SBA_UNROLL
      for (int i = 0; i < kDim; ++i)
        if (i < m) { A[i * kDim + i] += diag_[i] / radius_; rhs[i] = -gs[i]; }
      bool valid = cholesky_solve(m, A, rhs, y);
      if (valid) {
        // -(J y)^T (f + J y / 2) = -g^T y - y^T H y / 2
        double gy = 0, yHy = 0;
        SBA_UNROLL
        for (int i = 0; i < kDim; ++i) {
          if (i < m) {
            gy += gs[i] * y[i];
            double s = 0;
            SBA_UNROLL
            for (int j = 0; j < kDim; ++j)
              if (j < m) s += Hs[i * kDim + j] * y[j];
            yHy += y[i] * s;
          }
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
      double delta[kDim];
      SBA_UNROLL
      for (int i = 0; i < kDim; ++i) delta[i] = i < m ? scale_[i] * y[i] : 0.0;
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
  double t_start_ = 0.0;
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

#if defined(__clang__)
#pragma STDC FP_CONTRACT DEFAULT
#endif
