// ORACLE -- TEST INFRASTRUCTURE ONLY.  PARITY UNPINNED AT THE CERES BOUNDARY (see sba_oracle.cpp).
//
// Restatement of the projected line search Ceres' TrustRegionMinimizer runs on every step of a
// bounds-constrained problem.  It matters for the reference's d-only stage only: that stage sets lower bounds
// (spherical_bundle_adjuster.cpp:1060-1061) and leaves every Solver::Options field except the four of
// .cpp:334-338 at its default, so
//     options.is_constrained == true  &&  max_num_line_search_step_size_iterations == 20 (> 0)
// and TrustRegionMinimizer::Minimize calls DoLineSearch(x, gradient, cost, &delta) between
// ComputeTrustRegionStep and ComputeCandidatePointAndEvaluateCost.  Restated here from Ceres' published
// sources (Ceres Solver is a third-party dependency that is not in /root/reference; version unpinned,
// CMakeLists.txt:12; the behaviour below is the same in 1.12 .. 2.2):
//   - TrustRegionMinimizer::DoLineSearch: an ARMIJO line search along delta, first trial step size 1.0,
//     initial cost = cost(x), initial gradient = gradient . delta; on success delta *= the accepted step size,
//     on failure delta is left unchanged.
//   - LineSearchFunction::Evaluate(a): x_a = Plus(x, a * delta) -- Plus projects onto the bounds -- value =
//     cost(x_a); with CUBIC interpolation (the default line_search_interpolation_type) also the directional
//     derivative gradient(x_a) . delta.
//   - ArmijoLineSearch::DoSearch: while cost(x_a) > cost(x) + sufficient_decrease * (gradient . delta) * a:
//     count an iteration (fail at max_num_iterations), choose the next a by minimising the polynomial that
//     interpolates the samples {a = 0, previous, current} (values and gradients) over
//     [max_step_contraction * a, min_step_contraction * a] = [1e-3 a, 0.6 a], fail if
//     a * |delta|_inf < min_line_search_step_size (1e-9).
//   - FindInterpolatingPolynomial (full-pivot LU of the Vandermonde-type system) and MinimizePolynomial
//     (mid point, end points, and the REAL PARTS of all roots of the derivative that fall in the interval).
#ifndef ORACLE_CERES_LINE_SEARCH_HPP_
#define ORACLE_CERES_LINE_SEARCH_HPP_

#include <algorithm>
#include <cmath>
#include <complex>
#include <functional>
#include <vector>

namespace orc_ls {

struct Sample {
  double x = 0, value = 0, gradient = 0;
  bool value_is_valid = false, gradient_is_valid = false;
};

// coefficients in decreasing order of degree, like Ceres' polynomial.h
inline double evaluate_polynomial(const std::vector<double>& p, double x) {
  double v = 0;
  for (double c : p) v = v * x + c;
  return v;
}

// FindInterpolatingPolynomial: one row per valid value / gradient, solved with a full-pivot LU (threshold 0).
inline std::vector<double> find_interpolating_polynomial(const std::vector<Sample>& samples) {
  int m = 0;
  for (const Sample& s : samples) m += (s.value_is_valid ? 1 : 0) + (s.gradient_is_valid ? 1 : 0);
  const int degree = m - 1;
  std::vector<double> A(static_cast<size_t>(m) * m, 0.0), b(static_cast<size_t>(m), 0.0);
  int row = 0;
  for (const Sample& s : samples) {
    if (s.value_is_valid) {
      for (int j = 0; j <= degree; ++j) A[row * m + j] = std::pow(s.x, degree - j);
      b[row++] = s.value;
    }
    if (s.gradient_is_valid) {
      for (int j = 0; j < degree; ++j) A[row * m + j] = (degree - j) * std::pow(s.x, degree - j - 1);
      b[row++] = s.gradient;
    }
  }
  std::vector<int> colperm(static_cast<size_t>(m));
  for (int i = 0; i < m; ++i) colperm[i] = i;
  for (int k = 0; k < m; ++k) {
    int pr = k, pc = k;
    double best = -1;
    for (int i = k; i < m; ++i)
      for (int j = k; j < m; ++j)
        if (std::fabs(A[i * m + j]) > best) { best = std::fabs(A[i * m + j]); pr = i; pc = j; }
    if (best == 0.0) break;   // the remaining block is exactly zero: those unknowns stay 0 (Eigen's convention)
    if (pr != k) { for (int j = 0; j < m; ++j) std::swap(A[k * m + j], A[pr * m + j]); std::swap(b[k], b[pr]); }
    if (pc != k) { for (int i = 0; i < m; ++i) std::swap(A[i * m + k], A[i * m + pc]); std::swap(colperm[k], colperm[pc]); }
    for (int i = k + 1; i < m; ++i) {
      const double f = A[i * m + k] / A[k * m + k];
      for (int j = k; j < m; ++j) A[i * m + j] -= f * A[k * m + j];
      b[i] -= f * b[k];
    }
  }
  std::vector<double> y(static_cast<size_t>(m), 0.0), out(static_cast<size_t>(m), 0.0);
  for (int i = m - 1; i >= 0; --i) {
    if (A[i * m + i] == 0.0) { y[i] = 0.0; continue; }
    double v = b[i];
    for (int j = i + 1; j < m; ++j) v -= A[i * m + j] * y[j];
    y[i] = v / A[i * m + i];
  }
  for (int i = 0; i < m; ++i) out[colperm[i]] = y[i];
  return out;
}

// FindPolynomialRoots: leading zeros removed; degree 1 and 2 in closed form exactly as Ceres does; degree >= 3
// Ceres takes the eigenvalues of the balanced companion matrix -- here all complex roots by simultaneous
// (Weierstrass / Durand-Kerner) iteration followed by Newton polishing, which yields the same set of roots.
inline std::vector<std::complex<double>> find_polynomial_roots(std::vector<double> p) {
  std::vector<std::complex<double>> roots;
  size_t lead = 0;
  while (lead + 1 < p.size() && p[lead] == 0.0) ++lead;
  p.erase(p.begin(), p.begin() + static_cast<long>(lead));
  const int degree = static_cast<int>(p.size()) - 1;
  if (degree <= 0) return roots;
  if (degree == 1) { roots.emplace_back(-p[1] / p[0], 0.0); return roots; }
  if (degree == 2) {
    const double a = p[0], b = p[1], c = p[2];
    const double D = b * b - 4 * a * c, sqrt_D = std::sqrt(std::fabs(D));
    if (D >= 0) {
      if (b >= 0) { roots.emplace_back((-b - sqrt_D) / (2.0 * a), 0.0); roots.emplace_back((2.0 * c) / (-b - sqrt_D), 0.0); }
      else { roots.emplace_back((2.0 * c) / (-b + sqrt_D), 0.0); roots.emplace_back((-b + sqrt_D) / (2.0 * a), 0.0); }
    } else {
      roots.emplace_back(-b / (2.0 * a), sqrt_D / (2.0 * a));
      roots.emplace_back(-b / (2.0 * a), -sqrt_D / (2.0 * a));
    }
    return roots;
  }
  typedef std::complex<long double> cld;
  std::vector<long double> q(p.size());
  for (size_t i = 0; i < p.size(); ++i) q[i] = static_cast<long double>(p[i]) / p[0];
  long double radius = 0;
  for (size_t i = 1; i < q.size(); ++i) radius = std::max(radius, std::fabs(q[i]));
  radius = 1 + radius;   // Cauchy bound
  std::vector<cld> z(static_cast<size_t>(degree));
  for (int k = 0; k < degree; ++k) z[k] = std::pow(cld(0.4L, 0.9L), k) * (radius * 0.5L);
  auto eval = [&](cld x) { cld v = 0; for (long double c : q) v = v * x + c; return v; };
  for (int it = 0; it < 2000; ++it) {
    long double change = 0, scale = 0;
    for (int k = 0; k < degree; ++k) {
      cld den = 1;
      for (int j = 0; j < degree; ++j) if (j != k) den *= (z[k] - z[j]);
      if (std::abs(den) == 0) den = cld(1e-300L, 0);
      const cld dz = eval(z[k]) / den;
      z[k] -= dz;
      change = std::max(change, std::abs(dz));
      scale = std::max(scale, std::abs(z[k]));
    }
    if (change <= 1e-19L * std::max(scale, 1.0L)) break;
  }
  for (int k = 0; k < degree; ++k) {   // Newton polish on the original polynomial
    for (int it = 0; it < 4; ++it) {
      cld v = 0, dv = 0;
      for (long double c : q) { dv = dv * z[k] + v; v = v * z[k] + c; }
      if (std::abs(dv) == 0) break;
      z[k] -= v / dv;
    }
    roots.emplace_back(static_cast<double>(z[k].real()), static_cast<double>(z[k].imag()));
  }
  return roots;
}

// MinimizePolynomial over [x_min, x_max].
inline void minimize_polynomial(const std::vector<double>& p, double x_min, double x_max, double* opt_x, double* opt_v) {
  *opt_x = (x_min + x_max) / 2.0;
  *opt_v = evaluate_polynomial(p, *opt_x);
  const double vmin = evaluate_polynomial(p, x_min);
  if (vmin < *opt_v) { *opt_v = vmin; *opt_x = x_min; }
  const double vmax = evaluate_polynomial(p, x_max);
  if (vmax < *opt_v) { *opt_v = vmax; *opt_x = x_max; }
  if (p.size() <= 2) return;
  std::vector<double> d(p.size() - 1);
  const int degree = static_cast<int>(p.size()) - 1;
  for (int i = 0; i < degree; ++i) d[i] = (degree - i) * p[i];
  for (const std::complex<double>& r : find_polynomial_roots(d)) {
    const double root = r.real();   // Ceres looks at the real part of every root, complex ones included
    if (root < x_min || root > x_max) continue;
    const double v = evaluate_polynomial(p, root);
    if (v < *opt_v) { *opt_v = v; *opt_x = root; }
  }
}

// LineSearch::InterpolatingPolynomialMinimizingStepSize with interpolation_type == CUBIC.
inline double interpolating_step_size(const Sample& lowerbound, const Sample& previous, const Sample& current,
                                      double min_step_size, double max_step_size) {
  if (!current.value_is_valid) return std::min(std::max(current.x * 0.5, min_step_size), max_step_size);
  std::vector<Sample> samples;
  samples.push_back(lowerbound);
  samples.push_back(current);
  if (previous.value_is_valid) samples.push_back(previous);
  double step = 0, unused = 0;
  minimize_polynomial(find_interpolating_polynomial(samples), min_step_size, max_step_size, &step, &unused);
  return step;
}

struct Options {
  int max_num_iterations = 20;            // Solver::Options::max_num_line_search_step_size_iterations
  double sufficient_decrease = 1e-4;      // line_search_sufficient_function_decrease
  double max_step_contraction = 1e-3;     // max_line_search_step_contraction
  double min_step_contraction = 0.6;      // min_line_search_step_contraction
  double min_step_size = 1e-9;            // min_line_search_step_size
};
struct Summary { bool success = false; double step_size = 1.0; int num_iterations = 0, num_evaluations = 0; };

// ArmijoLineSearch::DoSearch(step_size_estimate = 1.0, ...).  evaluate(a, &value, &directional_gradient) returns
// false when the value is not finite.
inline Summary armijo_search(const Options& o, double initial_cost, double initial_gradient, double direction_max_norm,
                             const std::function<bool(double, double*, double*)>& evaluate) {
  Summary sum;
  Sample initial;
  initial.x = 0; initial.value = initial_cost; initial.gradient = initial_gradient;
  initial.value_is_valid = initial.gradient_is_valid = true;
  Sample previous, current;
  auto eval_at = [&](double a) {
    current = Sample();
    current.x = a;
    ++sum.num_evaluations;
    const bool ok = evaluate(a, &current.value, &current.gradient);
    current.value_is_valid = ok && std::isfinite(current.value);
    current.gradient_is_valid = current.value_is_valid && std::isfinite(current.gradient);
  };
  eval_at(1.0);
  while (!current.value_is_valid || current.value > initial_cost + o.sufficient_decrease * initial_gradient * current.x) {
    ++sum.num_iterations;
    if (sum.num_iterations >= o.max_num_iterations) return sum;
    const double step = interpolating_step_size(initial, previous, current, o.max_step_contraction * current.x,
                                                o.min_step_contraction * current.x);
    if (step * direction_max_norm < o.min_step_size) return sum;
    previous = current;
    eval_at(step);
  }
  sum.success = true;
  sum.step_size = current.x;
  return sum;
}

}  // namespace orc_ls
#endif  // ORACLE_CERES_LINE_SEARCH_HPP_
