// Projected Armijo line search of the bounded d-only stage (host side).
//
// The reference's d-only problem sets lower bounds (spherical_bundle_adjuster.cpp:1060-1061) and leaves
// Solver::Options::max_num_line_search_step_size_iterations at Ceres' default of 20 (.cpp:334-338 set four other
// fields), so Ceres' TrustRegionMinimizer runs DoLineSearch on every trust-region step of that stage: an ARMIJO search
// along the step `delta`, first trial size 1, accepted when
//     cost(P(x + a delta)) <= cost(x) + 1e-4 * a * gradient(x).delta              (P = projection onto the bounds),
// otherwise the next size is the minimiser over [1e-3 a, 0.6 a] of the polynomial that interpolates value and
// directional derivative at a = 0, at the current and (from the second contraction on) at the previous trial
// (line_search_interpolation_type = CUBIC), at most 20 contractions, and never below 1e-9 / |delta|_inf.  On success
// the step is scaled by the accepted size; on failure it is left as it was.
//
// ArmijoSearch is a resumable state machine (like LmSolver): query() names the step size whose projected candidate
// the device must evaluate, feed() takes that candidate's cost and directional derivative.
//
// Numerics: Ceres solves a dense system for the monomial coefficients and takes companion-matrix eigenvalues of the
// derivative; here the interpolant is kept in Newton (divided-difference) form on the doubled nodes and its minimum on
// the interval is found from the real critical points only, isolated by recursive differentiation + bisection.  The
// real part of a complex critical point, which Ceres also looks at, can never undercut the smallest of {mid point, end
// points, real critical points}, so both procedures pick the same minimiser (up to rounding).
#pragma once
#include <algorithm>
#include <cmath>

#include "../../include/sba_hip.h"

// The whole search is __host__ __device__: the host drives the single-problem and the lock-step batched d-only stage with
// it, the one-launch batched stage (batch_depth_solve_kernel) runs one per pair on the device -- the same source.
#ifndef SBA_HD
#if defined(__HIPCC__)
#define SBA_HD __host__ __device__
#else
#define SBA_HD
#endif
#endif
// The step logic is the same SOURCE on both sides and is meant to give the same BITS: no fused multiply-adds on the device
// where the host (x86-64 baseline) has none.  Block-scoped, so the per-match loops of the including kernels keep theirs.
#ifndef SBA_NO_CONTRACT
#if defined(__clang__)
#define SBA_NO_CONTRACT _Pragma("clang fp contract(off)")
#else
#define SBA_NO_CONTRACT
#endif
#endif

namespace sba {
namespace ls {

struct Sample { double x, f, df; };

// Hermite interpolant through (x_k, f_k, f'_k), k < count <= 3, in Newton form on nodes z = x0,x0,x1,x1,...
struct Hermite {
  int m = 0;            // number of coefficients (2 * count)
  double z[6], c[6];    // p(x) = c0 + c1 (x - z0) + c2 (x - z0)(x - z1) + ...
  SBA_HD void fit(const Sample* s, int count) {
    SBA_NO_CONTRACT
    m = 2 * count;
    double q[6][6];
    for (int k = 0; k < count; ++k) {
      z[2 * k] = z[2 * k + 1] = s[k].x;
      q[2 * k][0] = q[2 * k + 1][0] = s[k].f;
    }
    for (int i = 1; i < m; ++i)
      q[i][1] = (i % 2 == 1) ? s[i / 2].df : (q[i][0] - q[i - 1][0]) / (z[i] - z[i - 1]);
    for (int j = 2; j < m; ++j)
      for (int i = j; i < m; ++i) q[i][j] = (q[i][j - 1] - q[i - 1][j - 1]) / (z[i] - z[i - j]);
    for (int i = 0; i < m; ++i) c[i] = q[i][i];
  }
  // monomial coefficients a[0] + a[1] x + ... + a[m-1] x^(m-1)
  SBA_HD void monomial(double* a) const {
    SBA_NO_CONTRACT
    for (int i = 0; i < m; ++i) a[i] = 0.0;
    a[0] = c[m - 1];
    int deg = 0;
    for (int k = m - 2; k >= 0; --k) {      // a(x) <- a(x) (x - z_k) + c_k
      ++deg;
      for (int i = deg; i >= 1; --i) a[i] = a[i - 1] - z[k] * a[i];
      a[0] = c[k] - z[k] * a[0];
    }
  }
};

SBA_HD inline double horner(const double* a, int n, double x) {   // a[0..n-1], ascending powers
  SBA_NO_CONTRACT
  double v = 0.0;
  for (int i = n - 1; i >= 0; --i) v = v * x + a[i];
  return v;
}

// All real roots of a[0] + ... + a[n-1] x^(n-1) inside [lo, hi] (n <= 6), by isolating them between the critical
// points of the polynomial (found the same way one degree lower) and bisecting each monotone piece.  The descent over the
// degree is a template (NMAX = most coefficients this level can be handed), not a run-time recursion: the device build
// needs no call stack for it.
template <int NMAX>
SBA_HD inline int real_roots_in(const double* a, int n, double lo, double hi, double* roots) {
  SBA_NO_CONTRACT
  while (n > 1 && a[n - 1] == 0.0) --n;
  if (n <= 1) return 0;
  if (n == 2) {
    const double r = -a[0] / a[1];
    if (r >= lo && r <= hi) { roots[0] = r; return 1; }
    return 0;
  }
  double d[6], crit[6];
  for (int i = 1; i < n; ++i) d[i - 1] = i * a[i];
  const int nc = real_roots_in<(NMAX > 2 ? NMAX - 1 : 2)>(d, n - 1, lo, hi, crit);
  for (int i = 1; i < nc; ++i) {          // ascending (at most 4 values): insertion sort
    const double v = crit[i];
    int j = i - 1;
    for (; j >= 0 && crit[j] > v; --j) crit[j + 1] = crit[j];
    crit[j + 1] = v;
  }
  double edge[8];
  int ne = 0;
  edge[ne++] = lo;
  for (int i = 0; i < nc; ++i) edge[ne++] = crit[i];
  edge[ne++] = hi;
  int nr = 0;
  for (int i = 0; i + 1 < ne; ++i) {
    double xa = edge[i], xb = edge[i + 1];
    if (!(xb > xa)) continue;
    double fa = horner(a, n, xa), fb = horner(a, n, xb);
    if (fa == 0.0) { if (nr == 0 || roots[nr - 1] != xa) roots[nr++] = xa; continue; }
    if (fb == 0.0) { if (i + 2 == ne) roots[nr++] = xb; continue; }   // an interior edge is picked up as the next xa
    if ((fa < 0.0) == (fb < 0.0)) continue;
    for (int it = 0; it < 200; ++it) {
      const double xm = 0.5 * (xa + xb);
      if (xm == xa || xm == xb) break;
      const double fm = horner(a, n, xm);
      if (fm == 0.0) { xa = xb = xm; break; }
      if ((fm < 0.0) == (fa < 0.0)) { xa = xm; fa = fm; } else { xb = xm; fb = fm; }
    }
    roots[nr++] = 0.5 * (xa + xb);
  }
  return nr;
}
template <>
SBA_HD inline int real_roots_in<2>(const double* a, int n, double lo, double hi, double* roots) {
  SBA_NO_CONTRACT
  while (n > 1 && a[n - 1] == 0.0) --n;
  if (n <= 1) return 0;
  const double r = -a[0] / a[1];       // n == 2 here: a level that can be handed at most two coefficients
  if (r >= lo && r <= hi) { roots[0] = r; return 1; }
  return 0;
}

// argmin over [lo, hi] of the Hermite interpolant of the samples; candidates in Ceres' order (mid, lo, hi, critical
// points), a later candidate wins only if strictly smaller.
SBA_HD inline double hermite_argmin(const Sample* s, int count, double lo, double hi) {
  SBA_NO_CONTRACT
  Hermite h;
  h.fit(s, count);
  double a[6], d[6], roots[6];
  h.monomial(a);
  const int n = h.m;
  double best_x = 0.5 * (lo + hi), best_v = horner(a, n, best_x);
  const double vlo = horner(a, n, lo), vhi = horner(a, n, hi);
  if (vlo < best_v) { best_v = vlo; best_x = lo; }
  if (vhi < best_v) { best_v = vhi; best_x = hi; }
  for (int i = 1; i < n; ++i) d[i - 1] = i * a[i];
  const int nr = real_roots_in<5>(d, n - 1, lo, hi, roots);      // the derivative of a quintic at most: 5 coefficients
  for (int i = 0; i < nr; ++i) {
    const double v = horner(a, n, roots[i]);
    if (v < best_v) { best_v = v; best_x = roots[i]; }
  }
  return best_x;
}

class ArmijoSearch {
 public:
  SBA_HD void start(const sba_lm_options& o, double cost0, double gradient0, double direction_max_norm) {
    SBA_NO_CONTRACT
    o_ = o;
    s0_ = Sample{0.0, cost0, gradient0};
    dmax_ = direction_max_norm;
    have_prev_ = false;
    iterations_ = 0;
    done_ = success_ = false;
    query_ = 1.0;
  }
  SBA_HD bool done() const { return done_; }
  SBA_HD bool success() const { return success_; }
  SBA_HD double query() const { return query_; }             // step size to evaluate next
  SBA_HD double step_size() const { return success_ ? cur_.x : 1.0; }
  SBA_HD int num_iterations() const { return iterations_; }  // contractions (what Ceres adds to num_line_search_steps)

  // value / directional derivative (gradient at the projected trial point . delta) at query()
  SBA_HD void feed(double value, double dir_gradient) {
    SBA_NO_CONTRACT
    if (done_) return;
    const bool valid = std::isfinite(value);
    cur_ = Sample{query_, value, dir_gradient};
    if (valid && !(value > s0_.f + o_.line_search_sufficient_function_decrease * s0_.df * cur_.x)) {
      done_ = success_ = true;
      return;
    }
    if (++iterations_ >= o_.max_num_line_search_step_size_iterations) { done_ = true; return; }
    const double lo = o_.max_line_search_step_contraction * cur_.x, hi = o_.min_line_search_step_contraction * cur_.x;
    // A non-finite sample cannot be interpolated: halve inside the allowed interval (Ceres does this for an invalid
    // value; a finite cost with a non-finite gradient cannot occur for this residual and is treated the same way).
    const bool usable = valid && std::isfinite(dir_gradient);
    double next;
    if (!usable) {
      next = std::min(std::max(cur_.x * 0.5, lo), hi);
    } else {
      Sample s[3] = {s0_, cur_, prev_};
      next = hermite_argmin(s, have_prev_ ? 3 : 2, lo, hi);
    }
    if (next * dmax_ < o_.min_line_search_step_size) { done_ = true; return; }
    prev_ = cur_;
    have_prev_ = usable;
    query_ = next;
  }

 private:
  sba_lm_options o_{};
  Sample s0_{}, cur_{}, prev_{};
  double dmax_ = 0, query_ = 1.0;
  bool have_prev_ = false, done_ = false, success_ = false;
  int iterations_ = 0;
};

}  // namespace ls
}  // namespace sba
