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
//   - stop on parameter / function / gradient tolerance, max iterations, min radius
//   - additive update of the angle-axis vector (no manifold is set anywhere in the reference)
// One device sweep per LM iteration: the candidate point is evaluated with its Jacobian, so an
// accepted step needs no second sweep (and a multi-GPU run needs one all-reduce per iteration).
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
  // pick the coordinate axis least aligned with u
  int k = 0;
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

// Evaluator: bool(const double rot[3], const double tran[3], sba_normal_eq* out)
template <typename Evaluator>
int lm_solve(int mode, double rot[3], double tran[3], const sba_lm_options& o, Evaluator&& evaluate,
             sba_lm_summary* sum) {
  using namespace detail;
  const auto t_start = std::chrono::steady_clock::now();
  std::memset(sum, 0, sizeof(*sum));
  sum->termination = SBA_TERM_FAILURE;
  auto finish = [&](int term, double cost, double gmax, double radius) {
    sum->termination = term;
    sum->final_cost = cost;
    sum->final_gradient_max_norm = gmax;
    sum->final_radius = radius;
    sum->seconds_total =
        std::chrono::duration<double>(std::chrono::steady_clock::now() - t_start).count();
  };
  const bool rot_free = mode != SBA_MODE_TRAN, tran_free = mode != SBA_MODE_ROT;

  sba_normal_eq cur;
  if (!evaluate(rot, tran, &cur)) { finish(SBA_TERM_FAILURE, 0, 0, 0); return SBA_ERR_NUMERIC; }
  sum->num_evaluations = 1;
  if (!std::isfinite(cur.cost)) { finish(SBA_TERM_FAILURE, cur.cost, 0, 0); return SBA_ERR_NUMERIC; }
  sum->initial_cost = cur.cost;

  Param par;
  par.build(mode, o.tran_param, tran);
  const int m = par.m;
  double Hf[36], gf[6], scale[6];
  par.project(cur, Hf, gf);
  for (int i = 0; i < m; ++i)
    scale[i] = o.jacobi_scaling ? 1.0 / (1.0 + std::sqrt(std::max(Hf[i * m + i], 0.0))) : 1.0;
  auto gmax_of = [&](const double* g) {
    double v = 0;
    for (int i = 0; i < m; ++i) v = std::max(v, std::fabs(g[i]));
    return v;
  };
  double gmax = gmax_of(gf);
  double radius = o.initial_trust_region_radius, decrease = 2.0;
  bool reuse_diagonal = false;
  double diag[6] = {0};
  if (o.verbose)
    std::printf("iter      cost      cost_change  |gradient|   |step|    tr_ratio  tr_radius\n"
                "%4d % .6e    0.00e+00    %.2e   0.00e+00   0.00e+00  %.2e\n", 0, cur.cost, gmax, radius);
  if (gmax <= o.gradient_tolerance) { finish(SBA_TERM_CONVERGENCE_GRADIENT, cur.cost, gmax, radius); return SBA_OK; }

  int invalid_steps = 0;
  for (int iter = 1;; ++iter) {
    if (iter > o.max_num_iterations) { finish(SBA_TERM_NO_CONVERGENCE, cur.cost, gmax, radius); return SBA_OK; }
    if (radius < o.min_trust_region_radius) { finish(SBA_TERM_MIN_RADIUS, cur.cost, gmax, radius); return SBA_OK; }
    sum->num_iterations = iter;

    // scaled system
    double Hs[36], gs[6], A[36], rhs[6], y[6];
    for (int i = 0; i < m; ++i) {
      gs[i] = scale[i] * gf[i];
      for (int j = 0; j < m; ++j) Hs[i * m + j] = scale[i] * Hf[i * m + j] * scale[j];
    }
    if (!reuse_diagonal)
      for (int i = 0; i < m; ++i)
        diag[i] = std::min(std::max(Hs[i * m + i], o.min_lm_diagonal), o.max_lm_diagonal);
    std::memcpy(A, Hs, sizeof(double) * m * m);
    for (int i = 0; i < m; ++i) { A[i * m + i] += diag[i] / radius; rhs[i] = -gs[i]; }
    bool valid = cholesky_solve(m, A, rhs, y);
    double model_change = 0.0;
    if (valid) {
      // -(J y)^T (f + J y / 2) = -g^T y - y^T H y / 2
      double gy = 0, yHy = 0;
      for (int i = 0; i < m; ++i) {
        gy += gs[i] * y[i];
        double s = 0;
        for (int j = 0; j < m; ++j) s += Hs[i * m + j] * y[j];
        yHy += y[i] * s;
      }
      model_change = -gy - 0.5 * yHy;
      valid = model_change > 0.0;
    }
    if (!valid) {
      if (++invalid_steps >= 5) { finish(SBA_TERM_FAILURE, cur.cost, gmax, radius); return SBA_ERR_NUMERIC; }
      radius /= decrease; decrease *= 2.0; reuse_diagonal = true;
      continue;
    }
    invalid_steps = 0;
    double delta[6];
    for (int i = 0; i < m; ++i) delta[i] = scale[i] * y[i];

    double rot_c[3], tran_c[3];
    par.plus(rot, tran, delta, rot_c, tran_c);
    sba_normal_eq cand;
    if (!evaluate(rot_c, tran_c, &cand)) { finish(SBA_TERM_FAILURE, cur.cost, gmax, radius); return SBA_ERR_NUMERIC; }
    sum->num_evaluations++;

    double step2 = 0, x2 = 0;
    for (int a = 0; a < 3; ++a) {
      if (rot_free) { step2 += (rot_c[a] - rot[a]) * (rot_c[a] - rot[a]); x2 += rot[a] * rot[a]; }
      if (tran_free) { step2 += (tran_c[a] - tran[a]) * (tran_c[a] - tran[a]); x2 += tran[a] * tran[a]; }
    }
    const double step_norm = std::sqrt(step2), x_norm = std::sqrt(x2);
    const double cost_change = cur.cost - cand.cost;
    const double rho = std::isfinite(cand.cost) ? cost_change / model_change : -1.0;
    if (o.verbose)
      std::printf("%4d % .6e   % .2e    %.2e   %.2e  % .2e  %.2e\n", iter, cand.cost, cost_change,
                  gmax, step_norm, rho, radius);
    if (step_norm <= o.parameter_tolerance * (x_norm + o.parameter_tolerance)) {
      finish(SBA_TERM_CONVERGENCE_PARAMETER, cur.cost, gmax, radius);
      return SBA_OK;
    }
    if (std::isfinite(cand.cost) && std::fabs(cost_change) <= o.function_tolerance * cur.cost) {
      finish(SBA_TERM_CONVERGENCE_FUNCTION, cur.cost, gmax, radius);
      return SBA_OK;
    }
    if (rho > o.min_relative_decrease) {
      for (int a = 0; a < 3; ++a) { rot[a] = rot_c[a]; tran[a] = tran_c[a]; }
      cur = cand;
      sum->num_successful_steps++;
      par.build(mode, o.tran_param, tran);
      par.project(cur, Hf, gf);
      gmax = gmax_of(gf);
      const double t3 = 2.0 * rho - 1.0;
      radius = std::min(o.max_trust_region_radius, radius / std::max(1.0 / 3.0, 1.0 - t3 * t3 * t3));
      decrease = 2.0;
      reuse_diagonal = false;
      if (gmax <= o.gradient_tolerance) { finish(SBA_TERM_CONVERGENCE_GRADIENT, cur.cost, gmax, radius); return SBA_OK; }
    } else {
      radius /= decrease; decrease *= 2.0; reuse_diagonal = true;
    }
  }
}

}  // namespace sba
