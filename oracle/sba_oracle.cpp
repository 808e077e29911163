// ORACLE -- TEST INFRASTRUCTURE ONLY.  PARITY UNPINNED AT THE CERES BOUNDARY.
//
// CPU restatement of the reference's spherical bundle-adjustment hot path, used ONLY by tests/,
// __graft_entry__.smoke() and the cpu_baseline leg of bench.py as the checker / timed baseline.
// Nothing under spherical_bundle_adjuster_amd/ may include, link or call this file.
//
// Why "parity unpinned": the reference (whdlgp/spherical_bundle_adjuster) delegates the Jacobians,
// the rotation, the robust loss and the whole minimiser to Ceres Solver (find_package(Ceres) with
// no version pin, CMakeLists.txt:12) and ships no test, golden vector or data file that touches the
// BA path (test/feature_test.cpp exercises the matchers only).  Neither Ceres nor OpenCV exist in
// this image, so the reference cannot be built or run here.  What follows restates
//   - the reference's own functor bodies, line by line in meaning (citations on each function), and
//   - Ceres' documented public semantics for AngleAxisRotatePoint, AutoDiffCostFunction (forward-mode
//     dual numbers), HuberLoss + Corrector, and the default Levenberg-Marquardt trust-region loop.
// It is cross-checked in tests/ against an independent numpy analytic Jacobian, torch f64 autograd
// and central finite differences, but not against a running Ceres.
//
// Build: make -C oracle   (g++ -O2 -fopenmp, see oracle/Makefile)
#include <algorithm>
#include <cfloat>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <vector>

#include "ceres_line_search.hpp"

#ifdef _OPENMP
#include <omp.h>
#endif

namespace {

// ---- forward-mode dual number: stands in for ceres::Jet<double, N> ---------------------------------
template <int N>
struct Dual {
  double v;
  double d[N];
  Dual() : v(0) { for (int i = 0; i < N; ++i) d[i] = 0; }
  Dual(double x) : v(x) { for (int i = 0; i < N; ++i) d[i] = 0; }  // NOLINT: implicit like Jet
  static Dual var(double x, int k) { Dual r(x); r.d[k] = 1.0; return r; }
};
template <int N> Dual<N> operator+(const Dual<N>& a, const Dual<N>& b) { Dual<N> r; r.v = a.v + b.v; for (int i = 0; i < N; ++i) r.d[i] = a.d[i] + b.d[i]; return r; }
template <int N> Dual<N> operator-(const Dual<N>& a, const Dual<N>& b) { Dual<N> r; r.v = a.v - b.v; for (int i = 0; i < N; ++i) r.d[i] = a.d[i] - b.d[i]; return r; }
template <int N> Dual<N> operator*(const Dual<N>& a, const Dual<N>& b) { Dual<N> r; r.v = a.v * b.v; for (int i = 0; i < N; ++i) r.d[i] = a.d[i] * b.v + a.v * b.d[i]; return r; }
template <int N> Dual<N> operator/(const Dual<N>& a, const Dual<N>& b) { Dual<N> r; const double inv = 1.0 / b.v; r.v = a.v * inv; for (int i = 0; i < N; ++i) r.d[i] = (a.d[i] - r.v * b.d[i]) * inv; return r; }
template <int N> Dual<N> operator*(double a, const Dual<N>& b) { return Dual<N>(a) * b; }
template <int N> Dual<N> operator*(const Dual<N>& a, double b) { return a * Dual<N>(b); }
template <int N> Dual<N> operator-(const Dual<N>& a, double b) { return a - Dual<N>(b); }
template <int N> Dual<N> operator-(double a, const Dual<N>& b) { return Dual<N>(a) - b; }
template <int N> Dual<N> operator+(const Dual<N>& a, double b) { return a + Dual<N>(b); }
template <int N> bool operator>(const Dual<N>& a, double b) { return a.v > b; }
template <int N> Dual<N> sqrt(const Dual<N>& a) { Dual<N> r; r.v = std::sqrt(a.v); const double k = 0.5 / r.v; for (int i = 0; i < N; ++i) r.d[i] = a.d[i] * k; return r; }
template <int N> Dual<N> sin(const Dual<N>& a) { Dual<N> r; r.v = std::sin(a.v); const double c = std::cos(a.v); for (int i = 0; i < N; ++i) r.d[i] = a.d[i] * c; return r; }
template <int N> Dual<N> cos(const Dual<N>& a) { Dual<N> r; r.v = std::cos(a.v); const double s = -std::sin(a.v); for (int i = 0; i < N; ++i) r.d[i] = a.d[i] * s; return r; }
template <int N> Dual<N> exp(const Dual<N>& a) { Dual<N> r; r.v = std::exp(a.v); for (int i = 0; i < N; ++i) r.d[i] = a.d[i] * r.v; return r; }
inline double sqrt(double a) { return std::sqrt(a); }
inline double sin(double a) { return std::sin(a); }
inline double cos(double a) { return std::cos(a); }
inline double exp(double a) { return std::exp(a); }
inline double value_of(double a) { return a; }
template <int N> double value_of(const Dual<N>& a) { return a.v; }

// ---- Ceres AngleAxisRotatePoint semantics (call sites: spherical_bundle_adjuster.cpp:857,908,965,1019)
// Rodrigues' formula on the unit axis; for theta^2 <= DBL_EPSILON the first-order form p + w x p, so
// that derivatives stay finite at the origin.
template <typename T>
void angle_axis_rotate(const T w[3], const T p[3], T out[3]) {
  const T th2 = w[0] * w[0] + w[1] * w[1] + w[2] * w[2];
  if (th2 > DBL_EPSILON) {
    const T th = sqrt(th2);
    const T c = cos(th), s = sin(th);
    const T inv = T(1.0) / th;
    const T a[3] = {w[0] * inv, w[1] * inv, w[2] * inv};
    const T axp[3] = {a[1] * p[2] - a[2] * p[1], a[2] * p[0] - a[0] * p[2], a[0] * p[1] - a[1] * p[0]};
    const T k = (a[0] * p[0] + a[1] * p[1] + a[2] * p[2]) * (T(1.0) - c);
    for (int i = 0; i < 3; ++i) out[i] = p[i] * c + axp[i] * s + a[i] * k;
  } else {
    const T wxp[3] = {w[1] * p[2] - w[2] * p[1], w[2] * p[0] - w[0] * p[2], w[0] * p[1] - w[1] * p[0]};
    for (int i = 0; i < 3; ++i) out[i] = p[i] + wxp[i];
  }
}

// ---- the residual model common to all four functors ---------------------------------------------
// spherical_bundle_adjuster.cpp:846-865 (joint), :897-916 (rot-only), :953-973 (tran-only),
// :1008-1027 (d-only):  X1 = cam1*d[0]; X2 = cam2*d[1]; X1r = R(r) X1; X1_RT = X1r - t;
// residual = X2 - X1_RT.
template <typename T>
void reprojection_residual(const double cam1[3], const double cam2[3], const T d[2], const T r[3],
                           const T t[3], T res[3]) {
  T X1[3], X2[3], X1r[3];
  for (int i = 0; i < 3; ++i) { X1[i] = cam1[i] * d[0]; X2[i] = cam2[i] * d[1]; }
  angle_axis_rotate(r, X1, X1r);
  for (int i = 0; i < 3; ++i) {
    const T x1_rt = X1r[i] - t[i];
    res[i] = X2[i] - x1_rt;
  }
}

enum { MODE_ROT = 0, MODE_TRAN = 1, MODE_RT = 2 };

// residual e[3] and Jacobian J[3][6] over [rot | tran]; columns of frozen blocks are zero.
void point_residual_jacobian(int mode, const double cam1[3], const double cam2[3], const double rot[3],
                             const double tran[3], double d1, double d2, double e[3], double J[18]) {
  std::memset(J, 0, 18 * sizeof(double));
  if (mode == MODE_ROT) {          // AutoDiffCostFunction<rot_only, 3, 3>  (.cpp:931)
    typedef Dual<3> T;
    T r[3] = {T::var(rot[0], 0), T::var(rot[1], 1), T::var(rot[2], 2)};
    T t[3] = {T(tran[0]), T(tran[1]), T(tran[2])};
    T d[2] = {T(d1), T(d2)};
    T res[3];
    reprojection_residual(cam1, cam2, d, r, t, res);
    for (int i = 0; i < 3; ++i) { e[i] = res[i].v; for (int k = 0; k < 3; ++k) J[6 * i + k] = res[i].d[k]; }
  } else if (mode == MODE_TRAN) {  // AutoDiffCostFunction<tran_only, 3, 3> (.cpp:988)
    typedef Dual<3> T;
    T r[3] = {T(rot[0]), T(rot[1]), T(rot[2])};
    T t[3] = {T::var(tran[0], 0), T::var(tran[1], 1), T::var(tran[2], 2)};
    T d[2] = {T(d1), T(d2)};
    T res[3];
    reprojection_residual(cam1, cam2, d, r, t, res);
    for (int i = 0; i < 3; ++i) { e[i] = res[i].v; for (int k = 0; k < 3; ++k) J[6 * i + 3 + k] = res[i].d[k]; }
  } else {                         // joint functor (.cpp:880) with the d block held constant
    typedef Dual<6> T;
    T r[3] = {T::var(rot[0], 0), T::var(rot[1], 1), T::var(rot[2], 2)};
    T t[3] = {T::var(tran[0], 3), T::var(tran[1], 4), T::var(tran[2], 5)};
    T d[2] = {T(d1), T(d2)};
    T res[3];
    reprojection_residual(cam1, cam2, d, r, t, res);
    for (int i = 0; i < 3; ++i) { e[i] = res[i].v; for (int k = 0; k < 6; ++k) J[6 * i + k] = res[i].d[k]; }
  }
}

// ---- ceres::HuberLoss(a)::Evaluate + Corrector (loss attached at .cpp:887, :943, :1000) -------------
// rho = {rho(s), rho'(s), rho''(s)}.  rho'' <= 0 everywhere for Huber, so the corrector reduces to
// scaling residuals and Jacobian rows by sqrt(rho').
void huber_evaluate(double a, double s, double rho[3]) {
  const double b = a * a;
  if (s > b) {
    const double r = std::sqrt(s);
    rho[0] = 2.0 * a * r - b;
    rho[1] = std::max(DBL_MIN, a / r);
    rho[2] = -rho[1] / (2.0 * s);
  } else {
    rho[0] = s; rho[1] = 1.0; rho[2] = 0.0;
  }
}

struct NormalEq {
  long double H[36], g[6], cost, sum_w, n_out;
  NormalEq() { std::memset(this, 0, sizeof(*this)); }
  void add(const NormalEq& o) {
    for (int i = 0; i < 36; ++i) H[i] += o.H[i];
    for (int i = 0; i < 6; ++i) g[i] += o.g[i];
    cost += o.cost; sum_w += o.sum_w; n_out += o.n_out;
  }
};

// One residual block as Ceres evaluates it: functor through dual numbers, loss, corrector, then
// J^T J / J^T e accumulation (in long double so the oracle is far more accurate than any f64 sum order).
inline void accumulate_block(int mode, const double cam1[3], const double cam2[3], const double rot[3],
                             const double tran[3], double d1, double d2, double delta, NormalEq* ne) {
  double e[3], J[18];
  point_residual_jacobian(mode, cam1, cam2, rot, tran, d1, d2, e, J);
  const double s = e[0] * e[0] + e[1] * e[1] + e[2] * e[2];
  double rho[3] = {s, 1.0, 0.0};
  if (delta > 0.0) huber_evaluate(delta, s, rho);
  const double scale = std::sqrt(rho[1]);   // Corrector: rho'' <= 0 branch
  for (int i = 0; i < 3; ++i) { e[i] *= scale; for (int k = 0; k < 6; ++k) J[6 * i + k] *= scale; }
  for (int a = 0; a < 6; ++a) {
    for (int b = 0; b < 6; ++b) {
      long double acc = 0;
      for (int i = 0; i < 3; ++i) acc += static_cast<long double>(J[6 * i + a]) * J[6 * i + b];
      ne->H[6 * a + b] += acc;
    }
    long double acc = 0;
    for (int i = 0; i < 3; ++i) acc += static_cast<long double>(J[6 * i + a]) * e[i];
    ne->g[a] += acc;
  }
  ne->cost += 0.5L * rho[0];
  ne->sum_w += rho[1];
  ne->n_out += (delta > 0.0 && s > delta * delta) ? 1.0L : 0.0L;
}

struct EvalOut { double H[36], g[6], cost, sum_w, n_out; };

void evaluate_all(int mode, int per_match_depth, const double* x1, const double* x2, const double* d12,
                  size_t n, const double rot[3], const double tran[3], double d1, double d2, double delta,
                  int threads, EvalOut* out) {
  NormalEq total;
#ifdef _OPENMP
  if (threads <= 0) threads = omp_get_num_procs();
#else
  threads = 1;
#endif
  std::vector<NormalEq> part(static_cast<size_t>(threads));
#pragma omp parallel for num_threads(threads) schedule(static)
  for (int t = 0; t < threads; ++t) {
    const size_t lo = n * static_cast<size_t>(t) / threads, hi = n * static_cast<size_t>(t + 1) / threads;
    NormalEq local;
    for (size_t i = lo; i < hi; ++i) {
      const double a = per_match_depth ? d12[2 * i] : d1, b = per_match_depth ? d12[2 * i + 1] : d2;
      accumulate_block(mode, x1 + 3 * i, x2 + 3 * i, rot, tran, a, b, delta, &local);
    }
    part[static_cast<size_t>(t)] = local;
  }
  for (int t = 0; t < threads; ++t) total.add(part[static_cast<size_t>(t)]);
  for (int i = 0; i < 36; ++i) out->H[i] = static_cast<double>(total.H[i]);
  for (int i = 0; i < 6; ++i) out->g[i] = static_cast<double>(total.g[i]);
  out->cost = static_cast<double>(total.cost);
  out->sum_w = static_cast<double>(total.sum_w);
  out->n_out = static_cast<double>(total.n_out);
}

// The same faithful per-block evaluation (dual numbers, per-match trig, Huber corrector) but accumulating in plain
// double like Ceres' own linear-algebra back ends do: this is the TIMED cpu baseline of bench.py (the long-double
// accumulation above is for checking, and x87 arithmetic would make the CPU look slower than it is).
void evaluate_all_f64(int mode, int per_match_depth, const double* x1, const double* x2, const double* d12,
                      size_t n, const double rot[3], const double tran[3], double d1, double d2, double delta,
                      int threads, EvalOut* out) {
#ifdef _OPENMP
  if (threads <= 0) threads = omp_get_num_procs();
#else
  threads = 1;
#endif
  std::vector<EvalOut> part(static_cast<size_t>(threads));
#pragma omp parallel for num_threads(threads) schedule(static)
  for (int t = 0; t < threads; ++t) {
    const size_t lo = n * static_cast<size_t>(t) / threads, hi = n * static_cast<size_t>(t + 1) / threads;
    EvalOut acc;
    std::memset(&acc, 0, sizeof(acc));
    for (size_t i = lo; i < hi; ++i) {
      const double a = per_match_depth ? d12[2 * i] : d1, b = per_match_depth ? d12[2 * i + 1] : d2;
      double e[3], J[18];
      point_residual_jacobian(mode, x1 + 3 * i, x2 + 3 * i, rot, tran, a, b, e, J);
      const double s = e[0] * e[0] + e[1] * e[1] + e[2] * e[2];
      double rho[3] = {s, 1.0, 0.0};
      if (delta > 0.0) huber_evaluate(delta, s, rho);
      const double scale = std::sqrt(rho[1]);
      for (int r = 0; r < 3; ++r) { e[r] *= scale; for (int k = 0; k < 6; ++k) J[6 * r + k] *= scale; }
      for (int p = 0; p < 6; ++p) {
        for (int q = p; q < 6; ++q) acc.H[6 * p + q] += J[p] * J[q] + J[6 + p] * J[6 + q] + J[12 + p] * J[12 + q];
        acc.g[p] += J[p] * e[0] + J[6 + p] * e[1] + J[12 + p] * e[2];
      }
      acc.cost += 0.5 * rho[0];
      acc.sum_w += rho[1];
      acc.n_out += (delta > 0.0 && s > delta * delta) ? 1.0 : 0.0;
    }
    part[static_cast<size_t>(t)] = acc;
  }
  std::memset(out, 0, sizeof(*out));
  for (int t = 0; t < threads; ++t) {
    const EvalOut& a = part[static_cast<size_t>(t)];
    for (int i = 0; i < 36; ++i) out->H[i] += a.H[i];
    for (int i = 0; i < 6; ++i) out->g[i] += a.g[i];
    out->cost += a.cost; out->sum_w += a.sum_w; out->n_out += a.n_out;
  }
  for (int p = 0; p < 6; ++p) for (int q = 0; q < p; ++q) out->H[6 * p + q] = out->H[6 * q + p];
}

// "Optimised CPU" baseline (BASELINE.md variant B): rotation and its derivative matrices hoisted out
// of the loop (central differences are NOT used: the matrices come from dual numbers once), analytic
// per-match arithmetic in plain double, OpenMP reduction.  Timed next to the faithful loop so the
// GPU speed-up is not quoted against per-match trig alone.
void evaluate_all_hoisted(int mode, int per_match_depth, const double* x1, const double* x2,
                          const double* d12, size_t n, const double rot[3], const double tran[3],
                          double d1u, double d2u, double delta, int threads, EvalOut* out) {
  // R and G_j = dR/dw_j from the dual-number rotation applied to the basis vectors
  double R[9], G[27];
  {
    typedef Dual<3> T;
    T w[3] = {T::var(rot[0], 0), T::var(rot[1], 1), T::var(rot[2], 2)};
    for (int c = 0; c < 3; ++c) {
      T p[3] = {T(c == 0 ? 1.0 : 0.0), T(c == 1 ? 1.0 : 0.0), T(c == 2 ? 1.0 : 0.0)}, q[3];
      angle_axis_rotate(w, p, q);
      for (int r = 0; r < 3; ++r) { R[3 * r + c] = q[r].v; for (int j = 0; j < 3; ++j) G[9 * j + 3 * r + c] = q[r].d[j]; }
    }
  }
#ifdef _OPENMP
  if (threads <= 0) threads = omp_get_num_procs();
#else
  threads = 1;
#endif
  const bool rot_free = mode != MODE_TRAN, tran_free = mode != MODE_ROT;
  std::vector<EvalOut> part(static_cast<size_t>(threads));
#pragma omp parallel for num_threads(threads) schedule(static)
  for (int t = 0; t < threads; ++t) {
    const size_t lo = n * static_cast<size_t>(t) / threads, hi = n * static_cast<size_t>(t + 1) / threads;
    EvalOut acc;
    std::memset(&acc, 0, sizeof(acc));
    for (size_t i = lo; i < hi; ++i) {
      const double* p = x1 + 3 * i;
      const double* q = x2 + 3 * i;
      const double d1 = per_match_depth ? d12[2 * i] : d1u, d2 = per_match_depth ? d12[2 * i + 1] : d2u;
      double e[3], J[18] = {0};
      for (int r = 0; r < 3; ++r) {
        const double rx = R[3 * r] * p[0] + R[3 * r + 1] * p[1] + R[3 * r + 2] * p[2];
        e[r] = d2 * q[r] - (d1 * rx - tran[r]);
        if (rot_free)
          for (int j = 0; j < 3; ++j)
            J[6 * r + j] = -d1 * (G[9 * j + 3 * r] * p[0] + G[9 * j + 3 * r + 1] * p[1] + G[9 * j + 3 * r + 2] * p[2]);
        if (tran_free) J[6 * r + 3 + r] = 1.0;
      }
      const double s = e[0] * e[0] + e[1] * e[1] + e[2] * e[2];
      double rho[3] = {s, 1.0, 0.0};
      if (delta > 0.0) huber_evaluate(delta, s, rho);
      const double w = rho[1];
      for (int a = 0; a < 6; ++a) {
        for (int b = a; b < 6; ++b)
          acc.H[6 * a + b] += w * (J[a] * J[b] + J[6 + a] * J[6 + b] + J[12 + a] * J[12 + b]);
        acc.g[a] += w * (J[a] * e[0] + J[6 + a] * e[1] + J[12 + a] * e[2]);
      }
      acc.cost += 0.5 * rho[0];
      acc.sum_w += w;
      acc.n_out += (delta > 0.0 && s > delta * delta) ? 1.0 : 0.0;
    }
    part[static_cast<size_t>(t)] = acc;
  }
  std::memset(out, 0, sizeof(*out));
  for (int t = 0; t < threads; ++t) {
    const EvalOut& a = part[static_cast<size_t>(t)];
    for (int i = 0; i < 36; ++i) out->H[i] += a.H[i];
    for (int i = 0; i < 6; ++i) out->g[i] += a.g[i];
    out->cost += a.cost; out->sum_w += a.sum_w; out->n_out += a.n_out;
  }
  for (int a = 0; a < 6; ++a) for (int b = 0; b < a; ++b) out->H[6 * a + b] = out->H[6 * b + a];
}

// ---- dense helpers for the minimiser restatement ----------------------------------------------------
bool spd_solve(int m, const double* A, const double* b, double* x) {
  std::vector<double> L(static_cast<size_t>(m) * m, 0.0), z(static_cast<size_t>(m));
  for (int j = 0; j < m; ++j) {
    double dj = A[j * m + j];
    for (int k = 0; k < j; ++k) dj -= L[j * m + k] * L[j * m + k];
    if (!(dj > 0.0) || !std::isfinite(dj)) return false;
    L[j * m + j] = std::sqrt(dj);
    for (int i = j + 1; i < m; ++i) {
      double v = A[i * m + j];
      for (int k = 0; k < j; ++k) v -= L[i * m + k] * L[j * m + k];
      L[i * m + j] = v / L[j * m + j];
    }
  }
  for (int i = 0; i < m; ++i) { double v = b[i]; for (int k = 0; k < i; ++k) v -= L[i * m + k] * z[k]; z[i] = v / L[i * m + i]; }
  for (int i = m - 1; i >= 0; --i) { double v = z[i]; for (int k = i + 1; k < m; ++k) v -= L[k * m + i] * x[k]; x[i] = v / L[i * m + i]; }
  return true;
}

struct LmOptions {   // mirrors the fields of sba_lm_options that the tests set (same order)
  int max_num_iterations;
  double initial_trust_region_radius, max_trust_region_radius, min_trust_region_radius;
  double min_relative_decrease, min_lm_diagonal, max_lm_diagonal;
  double function_tolerance, gradient_tolerance, parameter_tolerance;
  int jacobi_scaling;
  double huber_delta;
  int tran_param;
  int verbose;
  // projected line search of bounds-constrained problems (d-only stage), Ceres Solver::Options names
  int max_num_line_search_step_size_iterations;      // 20; 0 = no line search
  double line_search_sufficient_function_decrease;   // 1e-4
  double max_line_search_step_contraction;           // 1e-3
  double min_line_search_step_contraction;           // 0.6
  double min_line_search_step_size;                  // 1e-9
};
struct LmSummary { int termination, num_iterations, num_successful_steps, num_evaluations; double initial_cost, final_cost, final_gradient_max_norm, final_radius; int num_line_search_steps; };

void perp_basis(const double t[3], double b0[3], double b1[3]) {
  const double n = std::sqrt(t[0] * t[0] + t[1] * t[1] + t[2] * t[2]);
  double u[3] = {1, 0, 0};
  if (n > 0) for (int i = 0; i < 3; ++i) u[i] = t[i] / n;
  int k = 0;
  if (std::fabs(u[1]) < std::fabs(u[k])) k = 1;
  if (std::fabs(u[2]) < std::fabs(u[k])) k = 2;
  double a[3] = {0, 0, 0}; a[k] = 1;
  b0[0] = u[1] * a[2] - u[2] * a[1]; b0[1] = u[2] * a[0] - u[0] * a[2]; b0[2] = u[0] * a[1] - u[1] * a[0];
  const double n0 = std::sqrt(b0[0] * b0[0] + b0[1] * b0[1] + b0[2] * b0[2]);
  for (int i = 0; i < 3; ++i) b0[i] /= n0;
  b1[0] = u[1] * b0[2] - u[2] * b0[1]; b1[1] = u[2] * b0[0] - u[0] * b0[2]; b1[2] = u[0] * b0[1] - u[1] * b0[0];
}

}  // namespace

// ==================================================================================================
extern "C" {

int orc_num_procs(void) {
#ifdef _OPENMP
  return omp_get_num_procs();
#else
  return 1;
#endif
}

// Single residual block: e[3], J[3][6] (row-major) -- golden point-wise vectors.
void orc_point(int mode, const double* cam1, const double* cam2, const double* rot, const double* tran,
               double d1, double d2, double* e, double* J) {
  point_residual_jacobian(mode, cam1, cam2, rot, tran, d1, d2, e, J);
}

void orc_huber(double a, double s, double* rho) { huber_evaluate(a, s, rho); }

void orc_rotate(const double* w, const double* p, double* out) { angle_axis_rotate<double>(w, p, out); }

// out: H[36] g[6] cost sum_w n_out  (45 doubles)
void orc_eval(int mode, int per_match_depth, const double* x1, const double* x2, const double* d12,
              size_t n, const double* rot, const double* tran, double d1, double d2, double delta,
              int threads, double* out45) {
  EvalOut o;
  evaluate_all(mode, per_match_depth, x1, x2, d12, n, rot, tran, d1, d2, delta, threads, &o);
  std::memcpy(out45, &o, sizeof(o));
}

void orc_eval_f64(int mode, int per_match_depth, const double* x1, const double* x2, const double* d12,
                  size_t n, const double* rot, const double* tran, double d1, double d2, double delta,
                  int threads, double* out45) {
  EvalOut o;
  evaluate_all_f64(mode, per_match_depth, x1, x2, d12, n, rot, tran, d1, d2, delta, threads, &o);
  std::memcpy(out45, &o, sizeof(o));
}

void orc_eval_hoisted(int mode, int per_match_depth, const double* x1, const double* x2, const double* d12,
                      size_t n, const double* rot, const double* tran, double d1, double d2, double delta,
                      int threads, double* out45) {
  EvalOut o;
  evaluate_all_hoisted(mode, per_match_depth, x1, x2, d12, n, rot, tran, d1, d2, delta, threads, &o);
  std::memcpy(out45, &o, sizeof(o));
}

// Restatement of Ceres' default trust-region Levenberg-Marquardt loop (what ceres::Solve runs for the
// reference with the options of spherical_bundle_adjuster.cpp:334-338), on the dense normal equations.
// `faithful` = 1 evaluates with the dual-number loop, 0 with the hoisted loop.
int orc_lm_solve(int mode, int per_match_depth, const double* x1, const double* x2, const double* d12,
                 size_t n, double* rot, double* tran, double d1, double d2, const LmOptions* opt,
                 int threads, int faithful, LmSummary* sum) {
  const LmOptions& o = *opt;
  std::memset(sum, 0, sizeof(*sum));
  auto evaluate = [&](const double* r, const double* t, EvalOut* e) {
    if (faithful) evaluate_all(mode, per_match_depth, x1, x2, d12, n, r, t, d1, d2, o.huber_delta, threads, e);
    else evaluate_all_hoisted(mode, per_match_depth, x1, x2, d12, n, r, t, d1, d2, o.huber_delta, threads, e);
    sum->num_evaluations++;
  };
  const bool rot_free = mode != MODE_TRAN, tran_free = mode != MODE_ROT;
  const bool sphere = tran_free && o.tran_param == 1;
  const int m = (rot_free ? 3 : 0) + (tran_free ? (sphere ? 2 : 3) : 0);
  double tnorm = std::sqrt(tran[0] * tran[0] + tran[1] * tran[1] + tran[2] * tran[2]);

  // local Jacobian of Plus at x: columns of P (6 x m)
  std::vector<double> P(6 * static_cast<size_t>(m));
  auto build_P = [&](const double* t) {
    std::fill(P.begin(), P.end(), 0.0);
    int c = 0;
    if (rot_free) { for (int a = 0; a < 3; ++a) P[a * m + c + a] = 1; c += 3; }
    if (tran_free) {
      if (sphere) {
        double b0[3], b1[3];
        perp_basis(t, b0, b1);
        for (int r = 0; r < 3; ++r) { P[(3 + r) * m + c] = b0[r]; P[(3 + r) * m + c + 1] = b1[r]; }
      } else {
        for (int a = 0; a < 3; ++a) P[(3 + a) * m + c + a] = 1;
      }
    }
  };
  std::vector<double> H(static_cast<size_t>(m) * m), g(static_cast<size_t>(m)), scale(static_cast<size_t>(m)), diag(static_cast<size_t>(m));
  auto reduce = [&](const EvalOut& e) {
    for (int i = 0; i < m; ++i) {
      for (int j = 0; j < m; ++j) {
        double s = 0;
        for (int a = 0; a < 6; ++a) for (int b = 0; b < 6; ++b) s += P[a * m + i] * e.H[6 * a + b] * P[b * m + j];
        H[i * m + j] = s;
      }
      double s = 0;
      for (int a = 0; a < 6; ++a) s += P[a * m + i] * e.g[a];
      g[i] = s;
    }
  };
  auto max_abs = [&](const std::vector<double>& v) { double r = 0; for (double x : v) r = std::max(r, std::fabs(x)); return r; };

  EvalOut cur;
  evaluate(rot, tran, &cur);
  sum->initial_cost = cur.cost;
  build_P(tran);
  reduce(cur);
  for (int i = 0; i < m; ++i) scale[i] = o.jacobi_scaling ? 1.0 / (1.0 + std::sqrt(std::max(H[i * m + i], 0.0))) : 1.0;
  double gmax = max_abs(g), radius = o.initial_trust_region_radius, nu = 2.0;
  bool reuse_diag = false;
  int invalid = 0;
  auto done = [&](int term) { sum->termination = term; sum->final_cost = cur.cost; sum->final_gradient_max_norm = gmax; sum->final_radius = radius; return 0; };

  for (int it = 0;;) {
    // FinalizeIterationAndCheckIfMinimizerCanContinue: iteration limit, gradient tolerance, minimum radius
    if (it >= o.max_num_iterations) return done(4);
    if (gmax <= o.gradient_tolerance) return done(2);
    if (radius < o.min_trust_region_radius) return done(5);
    sum->num_iterations = ++it;
    std::vector<double> Hs(static_cast<size_t>(m) * m), gs(static_cast<size_t>(m)), A, rhs(static_cast<size_t>(m)), y(static_cast<size_t>(m));
    for (int i = 0; i < m; ++i) { gs[i] = scale[i] * g[i]; for (int j = 0; j < m; ++j) Hs[i * m + j] = scale[i] * H[i * m + j] * scale[j]; }
    if (!reuse_diag) for (int i = 0; i < m; ++i) diag[i] = std::min(std::max(Hs[i * m + i], o.min_lm_diagonal), o.max_lm_diagonal);
    A = Hs;
    for (int i = 0; i < m; ++i) { A[i * m + i] += diag[i] / radius; rhs[i] = -gs[i]; }
    bool ok = spd_solve(m, A.data(), rhs.data(), y.data());
    double model = 0;
    if (ok) {
      for (int i = 0; i < m; ++i) { double Hy = 0; for (int j = 0; j < m; ++j) Hy += Hs[i * m + j] * y[j]; model -= y[i] * (gs[i] + 0.5 * Hy); }
      ok = model > 0.0;
    }
    if (!ok) { if (++invalid >= 5) { done(6); return -6; } radius /= nu; nu *= 2; reuse_diag = true; continue; }
    invalid = 0;
    double step6[6] = {0};
    for (int a = 0; a < 6; ++a) for (int i = 0; i < m; ++i) step6[a] += P[a * m + i] * scale[i] * y[i];
    double rc[3], tc[3];
    for (int a = 0; a < 3; ++a) { rc[a] = rot[a] + step6[a]; tc[a] = tran[a] + step6[3 + a]; }
    if (sphere) { const double nn = std::sqrt(tc[0] * tc[0] + tc[1] * tc[1] + tc[2] * tc[2]); if (nn > 0) for (int a = 0; a < 3; ++a) tc[a] *= tnorm / nn; }
    EvalOut cand;
    evaluate(rc, tc, &cand);
    double step2 = 0, x2n = 0;
    for (int a = 0; a < 3; ++a) {
      if (rot_free) { step2 += (rc[a] - rot[a]) * (rc[a] - rot[a]); x2n += rot[a] * rot[a]; }
      if (tran_free) { step2 += (tc[a] - tran[a]) * (tc[a] - tran[a]); x2n += tran[a] * tran[a]; }
    }
    if (std::sqrt(step2) <= o.parameter_tolerance * (std::sqrt(x2n) + o.parameter_tolerance)) return done(3);
    const double change = cur.cost - cand.cost;
    if (std::isfinite(cand.cost) && std::fabs(change) <= o.function_tolerance * cur.cost) return done(1);
    const double quality = std::isfinite(cand.cost) ? change / model : -1.0;
    if (quality > o.min_relative_decrease) {
      for (int a = 0; a < 3; ++a) { rot[a] = rc[a]; tran[a] = tc[a]; }
      cur = cand;
      sum->num_successful_steps++;
      build_P(tran);
      reduce(cur);
      gmax = max_abs(g);
      const double q = 2.0 * quality - 1.0;
      radius = std::min(o.max_trust_region_radius, radius / std::max(1.0 / 3.0, 1.0 - q * q * q));
      nu = 2.0; reuse_diag = false;
    } else {
      radius /= nu; nu *= 2; reuse_diag = true;
    }
  }
}

// ---- d-only stage, spherical_bundle_adjuster.cpp:1004-1063 ------------------------------------------
// One Ceres problem with N residual blocks of 5 residuals over the block's own 2 parameters d[0], d[1]
// (AutoDiffCostFunction<d_only, 5, 2>, .cpp:1044), no loss (NULL, .cpp:1059), lower bound 0 on both
// (.cpp:1060-1061), lambda = c = 1 (.cpp:1057-1058).  The Hessian is block diagonal, but it is ONE trust-region
// problem: a single radius, a single accept/reject on the total cost, global convergence tests, Jacobi
// scaling per parameter, and Plus() projects the candidate onto the bounds (ParameterBlock::Plus).
// Because the problem is bounds-constrained and Solver::Options::max_num_line_search_step_size_iterations is
// left at its default of 20 (.cpp:334-338 touch only four other fields), every trust-region step first goes
// through Ceres' projected Armijo line search (ceres_line_search.hpp) before the candidate is evaluated.
// (ITERATIVE_SCHUR, .cpp:335: every parameter block here is its own independent set, so all of them are
// eliminated, the Schur complement is empty and the "iterative" solver is an exact block back-substitution --
// the 2x2 solves below.)
}  // extern "C"
namespace {
template <typename T>
void depth_residuals(const double cam1[3], const double cam2[3], const double rot[3], const double tran[3],
                     double lambda, double c, const T d[2], T res[5]) {
  T r[3] = {T(rot[0]), T(rot[1]), T(rot[2])};
  T t[3] = {T(tran[0]), T(tran[1]), T(tran[2])};
  reprojection_residual(cam1, cam2, d, r, t, res);          // .cpp:1008-1027
  res[3] = lambda * exp(T(-c) * d[0]);                        // .cpp:1028
  res[4] = lambda * exp(T(-c) * d[1]);                        // .cpp:1029
}
}  // namespace
extern "C" {

int orc_depth_solve(const double* x1, const double* x2, size_t n, const double* rot, const double* tran,
                    double lambda, double c, double* d12 /* in: initial, out: result */, const LmOptions* opt,
                    LmSummary* sum) {
  const LmOptions& o = *opt;
  std::memset(sum, 0, sizeof(*sum));
  std::vector<double> scale(2 * n), diag(2 * n), cand(2 * n), delta(2 * n), H(3 * n), g(2 * n), gtmp(2 * n);
  // cost at d; with_jac: also the block Hessians H, the gradient into gout and the projected-gradient max norm
  auto evaluate = [&](const double* d, bool with_jac, double* gout, double* Hout, double* gmax) {
    long double cost = 0;
    double gm = 0;
    for (size_t i = 0; i < n; ++i) {
      if (with_jac) {
        typedef Dual<2> T;
        T dd[2] = {T::var(d[2 * i], 0), T::var(d[2 * i + 1], 1)}, res[5];
        depth_residuals(x1 + 3 * i, x2 + 3 * i, rot, tran, lambda, c, dd, res);
        double h11 = 0, h12 = 0, h22 = 0, g1 = 0, g2 = 0, s = 0;
        for (int k = 0; k < 5; ++k) {
          h11 += res[k].d[0] * res[k].d[0]; h12 += res[k].d[0] * res[k].d[1]; h22 += res[k].d[1] * res[k].d[1];
          g1 += res[k].d[0] * res[k].v; g2 += res[k].d[1] * res[k].v; s += res[k].v * res[k].v;
        }
        if (Hout) { Hout[3 * i] = h11; Hout[3 * i + 1] = h12; Hout[3 * i + 2] = h22; }
        gout[2 * i] = g1; gout[2 * i + 1] = g2;
        cost += 0.5L * s;
        // projected gradient norm for the bounded problem: |x - P(x - g)|_inf
        gm = std::max(gm, std::fabs(d[2 * i] - std::max(d[2 * i] - g1, 0.0)));
        gm = std::max(gm, std::fabs(d[2 * i + 1] - std::max(d[2 * i + 1] - g2, 0.0)));
      } else {
        double dd[2] = {d[2 * i], d[2 * i + 1]}, res[5], s = 0;
        depth_residuals(x1 + 3 * i, x2 + 3 * i, rot, tran, lambda, c, dd, res);
        for (int k = 0; k < 5; ++k) s += res[k] * res[k];
        cost += 0.5L * s;
      }
    }
    if (gmax) *gmax = gm;
    sum->num_evaluations++;
    return static_cast<double>(cost);
  };
  double gmax = 0;
  double cost = evaluate(d12, true, g.data(), H.data(), &gmax);
  sum->initial_cost = cost;
  for (size_t k = 0; k < 2 * n; ++k) {
    const double hkk = H[3 * (k / 2) + (k % 2 ? 2 : 0)];
    scale[k] = o.jacobi_scaling ? 1.0 / (1.0 + std::sqrt(hkk)) : 1.0;
  }
  double radius = o.initial_trust_region_radius, nu = 2.0;
  bool reuse = false;
  int invalid = 0;
  auto done = [&](int term) { sum->termination = term; sum->final_cost = cost; sum->final_gradient_max_norm = gmax; sum->final_radius = radius; return 0; };
  for (int it = 0;;) {
    // FinalizeIterationAndCheckIfMinimizerCanContinue: iteration limit, gradient tolerance, minimum radius
    if (it >= o.max_num_iterations) return done(4);
    if (gmax <= o.gradient_tolerance) return done(2);
    if (radius < o.min_trust_region_radius) return done(5);
    sum->num_iterations = ++it;
    long double model = 0, x2n = 0, gdelta = 0;
    double dmax = 0;
    for (size_t i = 0; i < n; ++i) {
      const double s1 = scale[2 * i], s2 = scale[2 * i + 1];
      const double h11 = s1 * H[3 * i] * s1, h12 = s1 * H[3 * i + 1] * s2, h22 = s2 * H[3 * i + 2] * s2;
      const double g1 = s1 * g[2 * i], g2 = s2 * g[2 * i + 1];
      if (!reuse) {
        diag[2 * i] = std::min(std::max(h11, o.min_lm_diagonal), o.max_lm_diagonal);
        diag[2 * i + 1] = std::min(std::max(h22, o.min_lm_diagonal), o.max_lm_diagonal);
      }
      const double a11 = h11 + diag[2 * i] / radius, a22 = h22 + diag[2 * i + 1] / radius, a12 = h12;
      const double det = a11 * a22 - a12 * a12;
      const double y1 = (-g1 * a22 + g2 * a12) / det, y2 = (-g2 * a11 + g1 * a12) / det;
      model += -(g1 * y1 + g2 * y2) - 0.5 * (h11 * y1 * y1 + 2 * h12 * y1 * y2 + h22 * y2 * y2);
      delta[2 * i] = s1 * y1; delta[2 * i + 1] = s2 * y2;      // delta = step .* jacobian_scaling
      for (int k = 0; k < 2; ++k) {
        gdelta += static_cast<long double>(g[2 * i + k]) * delta[2 * i + k];
        dmax = std::max(dmax, std::fabs(delta[2 * i + k]));
        x2n += static_cast<long double>(d12[2 * i + k]) * d12[2 * i + k];
      }
    }
    if (!(model > 0)) { if (++invalid >= 5) { done(6); return -6; } radius /= nu; nu *= 2; reuse = true; continue; }
    invalid = 0;
    // x_plus(a) = Plus(x, a * delta): add, then project onto d >= 0
    auto plus = [&](double a, double* out) { for (size_t k = 0; k < 2 * n; ++k) out[k] = std::max(d12[k] + a * delta[k], 0.0); };
    if (o.max_num_line_search_step_size_iterations > 0) {     // TrustRegionMinimizer::DoLineSearch
      orc_ls::Options lo;
      lo.max_num_iterations = o.max_num_line_search_step_size_iterations;
      lo.sufficient_decrease = o.line_search_sufficient_function_decrease;
      lo.max_step_contraction = o.max_line_search_step_contraction;
      lo.min_step_contraction = o.min_line_search_step_contraction;
      lo.min_step_size = o.min_line_search_step_size;
      const orc_ls::Summary ls = orc_ls::armijo_search(lo, cost, static_cast<double>(gdelta), dmax,
          [&](double a, double* value, double* dir_gradient) {
            plus(a, cand.data());
            *value = evaluate(cand.data(), true, gtmp.data(), nullptr, nullptr);
            long double s = 0;
            for (size_t k = 0; k < 2 * n; ++k) s += static_cast<long double>(gtmp[k]) * delta[k];
            *dir_gradient = static_cast<double>(s);
            return true;
          });
      sum->num_line_search_steps += ls.num_iterations;
      if (ls.success) for (size_t k = 0; k < 2 * n; ++k) delta[k] *= ls.step_size;
    }
    plus(1.0, cand.data());
    long double step2 = 0;
    for (size_t k = 0; k < 2 * n; ++k) { const double dd = cand[k] - d12[k]; step2 += static_cast<long double>(dd) * dd; }
    const double cand_cost = evaluate(cand.data(), false, nullptr, nullptr, nullptr);
    if (std::sqrt(static_cast<double>(step2)) <= o.parameter_tolerance * (std::sqrt(static_cast<double>(x2n)) + o.parameter_tolerance)) return done(3);
    const double change = cost - cand_cost;
    if (std::fabs(change) <= o.function_tolerance * cost) return done(1);
    const double quality = change / static_cast<double>(model);
    if (quality > o.min_relative_decrease) {
      std::memcpy(d12, cand.data(), sizeof(double) * 2 * n);
      cost = evaluate(d12, true, g.data(), H.data(), &gmax);
      sum->num_successful_steps++;
      const double q = 2.0 * quality - 1.0;
      radius = std::min(o.max_trust_region_radius, radius / std::max(1.0 / 3.0, 1.0 - q * q * q));
      nu = 2.0; reuse = false;
    } else {
      radius /= nu; nu *= 2; reuse = true;
    }
  }
}

// Line-search pieces on their own (tests pin them against numpy.linalg.solve / numpy.roots).
void orc_ls_interpolating_polynomial(const double* samples /* k x (x, value, gradient) */, int k, double* coeffs) {
  std::vector<orc_ls::Sample> v(static_cast<size_t>(k));
  for (int i = 0; i < k; ++i) {
    v[i].x = samples[3 * i]; v[i].value = samples[3 * i + 1]; v[i].gradient = samples[3 * i + 2];
    v[i].value_is_valid = v[i].gradient_is_valid = true;
  }
  const std::vector<double> p = orc_ls::find_interpolating_polynomial(v);
  std::memcpy(coeffs, p.data(), sizeof(double) * p.size());
}
// ArmijoLineSearch::DoSearch on a caller-supplied phi(a) -> (value, slope); out3 = {success, step size, contractions}.
typedef int (*orc_phi_cb)(double step_size, double* value, double* slope, void* user);
int orc_ls_armijo(const LmOptions* o, double cost0, double slope0, double direction_max_norm, orc_phi_cb phi,
                  void* user, double* out3) {
  orc_ls::Options lo;
  lo.max_num_iterations = o->max_num_line_search_step_size_iterations;
  lo.sufficient_decrease = o->line_search_sufficient_function_decrease;
  lo.max_step_contraction = o->max_line_search_step_contraction;
  lo.min_step_contraction = o->min_line_search_step_contraction;
  lo.min_step_size = o->min_line_search_step_size;
  int cb_rc = 0;
  const orc_ls::Summary r = orc_ls::armijo_search(lo, cost0, slope0, direction_max_norm,
      [&](double a, double* v, double* g) { if (phi(a, v, g, user) != 0) { cb_rc = -1; return false; } return true; });
  out3[0] = r.success ? 1.0 : 0.0; out3[1] = r.step_size; out3[2] = r.num_iterations;
  return cb_rc;
}
void orc_ls_minimize_polynomial(const double* poly, int size, double x_min, double x_max, double* out2) {
  orc_ls::minimize_polynomial(std::vector<double>(poly, poly + size), x_min, x_max, &out2[0], &out2[1]);
}

// ---- pixel -> unit sphere, spherical_bundle_adjuster.cpp:271-298 -------------------------------------
void orc_keypoints_to_sphere(const uint8_t* kp, size_t n, size_t stride, int im_w, int im_h, double* out) {
  const double w = im_w, h = im_h;
  for (size_t i = 0; i < n; ++i) {
    float px, py;
    std::memcpy(&px, kp + i * stride, 4);
    std::memcpy(&py, kp + i * stride + 4, 4);
    const double lon = 2 * M_PI * (px / w);      // .cpp:279
    const double colat = M_PI * (py / h);        // .cpp:281
    out[3 * i + 0] = std::sin(colat) * std::cos(lon);   // .cpp:291-293
    out[3 * i + 1] = std::sin(colat) * std::sin(lon);
    out[3 * i + 2] = std::cos(colat);
  }
}

// ---- key-point coordinate maps of the matchers -------------------------------------------------------------
// eular2rot(Vec3f(0, RAD(pitch), 0)) (spherical_surf.cpp:18-45, call sites :84, :112): theta is a float, and
// with <cmath> + `using namespace std` cos/sin of a float are the float overloads; R_x = R_z = I, so R = R_y.
static void pitch_rotation(float pitch_deg, double R[9]) {
  const float th = static_cast<float>(M_PI * pitch_deg / 180.0);
  const double c = std::cos(th), s = std::sin(th);   // std::cos(float) -> float, widened
  const double Ry[9] = {c, 0, s, 0, 1, 0, -s, 0, c};
  std::memcpy(R, Ry, sizeof(Ry));
}
// rotate_pixel (spherical_surf.cpp:48-74): (row, col) ints -> sphere -> rotate -> (row, col) ints by truncation.
static void rotate_pixel_ref(int row, int col, const double R[9], int width, int height, int* out_row, int* out_col) {
  const double r0 = M_PI * row / height, r1 = 2 * M_PI * col / width;
  const double v[3] = {std::sin(r0) * std::cos(r1), std::sin(r0) * std::sin(r1), std::cos(r0)};
  const double w[3] = {R[0] * v[0] + R[1] * v[1] + R[2] * v[2], R[3] * v[0] + R[4] * v[1] + R[5] * v[2],
                       R[6] * v[0] + R[7] * v[1] + R[8] * v[2]};
  const double a = std::acos(w[2]);
  double b = std::atan2(w[1], w[0]);
  if (b < 0) b += M_PI * 2;
  *out_row = static_cast<int>(height * a / M_PI);
  *out_col = static_cast<int>(width * b / (2 * M_PI));
}
// rotate_keypoint (spherical_surf.cpp:110-123), in place on records whose first two floats are pt.x, pt.y.
void orc_rotate_keypoints(uint8_t* kp, size_t n, size_t stride, float pitch_deg, int width, int height) {
  double R[9];
  pitch_rotation(pitch_deg, R);
  for (size_t i = 0; i < n; ++i) {
    float px, py;
    std::memcpy(&px, kp + i * stride, 4);
    std::memcpy(&py, kp + i * stride + 4, 4);
    const int offset_i = static_cast<int>(py + height * 3 / 8);     // float + int, truncated (.cpp:116)
    int r, c;
    rotate_pixel_ref(offset_i, static_cast<int>(px), R, width, height, &r, &c);
    px = static_cast<float>(c); py = static_cast<float>(r);
    std::memcpy(kp + i * stride, &px, 4);
    std::memcpy(kp + i * stride + 4, &py, 4);
  }
}
// crop_rotated_image (spherical_surf.cpp:76-108): (H/4) x W band; pixels whose source falls outside stay 0 here
// (the reference leaves them uninitialised).
void orc_crop_rotated_image(const uint8_t* im, int im_h, int im_w, float pitch_deg, uint8_t* out) {
  double R[9];
  pitch_rotation(pitch_deg, R);
  std::memset(out, 0, static_cast<size_t>(im_h / 4) * im_w * 3);
  for (int i = 0; i < im_h / 4; ++i)
    for (int j = 0; j < im_w; ++j) {
      int r, c;
      rotate_pixel_ref(i + im_h * 3 / 8, j, R, im_w, im_h, &r, &c);
      if (r >= 0 && c >= 0 && r < im_h && c < im_w)
        std::memcpy(out + (static_cast<size_t>(i) * im_w + j) * 3, im + (static_cast<size_t>(r) * im_w + c) * 3, 3);
    }
}
// cube2equi_pixel (equi2cube_surf.cpp:19-76), in place on key-point records (pt in the 6S x S cube strip).
void orc_cube2equi_keypoints(uint8_t* kp, size_t n, size_t stride, int S, int im_w, int im_h) {
  for (size_t i = 0; i < n; ++i) {
    float cx, cy;
    std::memcpy(&cx, kp + i * stride, 4);
    std::memcpy(&cy, kp + i * stride + 4, 4);
    double v[3] = {0, 0, 0};
    if (cx < S) { v[0] = (S - 2.0 * cx) / S; v[1] = 1.0; v[2] = (S - 2.0 * cy) / S; }                                   // left
    else if (cx >= S && cx < 2 * S) { v[0] = -1.0; v[1] = (S - 2.0 * (cx - S)) / S; v[2] = (S - 2.0 * cy) / S; }         // front
    else if (cx >= 2 * S && cx < 3 * S) { v[0] = (2.0 * (cx - 2 * S) - S) / S; v[1] = -1.0; v[2] = (S - 2.0 * cy) / S; } // right
    else if (cx >= 3 * S && cx < 4 * S) { v[0] = 1.0; v[1] = (2.0 * (cx - 3 * S) - S) / S; v[2] = (S - 2.0 * cy) / S; }  // back
    else if (cx >= 4 * S && cx < 5 * S) { v[0] = (S - 2.0 * cy) / S; v[1] = (S - 2.0 * (cx - 4 * S)) / S; v[2] = 1.0; }  // top
    else if (cx >= 5 * S) { v[0] = (2.0 * cy - S) / S; v[1] = (S - 2.0 * (cx - 5 * S)) / S; v[2] = -1.0; }               // bottom
    const double nrm = std::sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
    const double a = std::acos(v[2] / nrm);
    double b = std::atan2(v[1] / nrm, v[0] / nrm);
    if (b < 0) b += M_PI * 2;
    const float ex = static_cast<float>(im_w * b / (2 * M_PI)), ey = static_cast<float>(im_h * a / M_PI);
    std::memcpy(kp + i * stride, &ex, 4);
    std::memcpy(kp + i * stride + 4, &ey, 4);
  }
}

// ---- ERP -> cubemap strip, equi2cube.cpp:12-302 -----------------------------------------------------
// face order of get_all (equi2cube.cpp:292-298): left, front, right, back, top, bottom.  `clamp` != 0
// clamps the source index into the image (the reference does not, equi2cube.cpp:47-50); the number of
// pixels where clamping changed anything is returned.
long orc_equi2cube(const uint8_t* im, int im_h, int im_w, int S, int clamp, uint8_t* out) {
  long clamped = 0;
  for (int face = 0; face < 6; ++face)
    for (int i = 0; i < S; ++i)
      for (int j = 0; j < S; ++j) {
        double v[3];
        switch (face) {
          case 0: v[0] = (S - 2.0 * j) / S; v[1] = 1.0; v[2] = (S - 2.0 * i) / S; break;           // left   :118-120
          case 1: v[0] = -1.0; v[1] = (S - 2.0 * j) / S; v[2] = (S - 2.0 * i) / S; break;          // front  :73-75
          case 2: v[0] = (2.0 * j - S) / S; v[1] = -1.0; v[2] = (S - 2.0 * i) / S; break;          // right  :163-165
          case 3: v[0] = 1.0; v[1] = (2.0 * j - S) / S; v[2] = (S - 2.0 * i) / S; break;           // back   :28-30
          case 4: v[0] = (S - 2.0 * i) / S; v[1] = (S - 2.0 * j) / S; v[2] = 1.0; break;           // top    :208-210
          default: v[0] = (2.0 * i - S) / S; v[1] = (S - 2.0 * j) / S; v[2] = -1.0; break;         // bottom :253-255
        }
        const double nrm = std::sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
        const double ux = v[0] / nrm, uy = v[1] / nrm, uz = v[2] / nrm;
        const double theta = std::acos(uz);
        double phi = std::atan2(uy, ux);
        if (phi < 0) phi += M_PI * 2;
        int row = static_cast<int>(im_h * theta / M_PI);
        int col = static_cast<int>(im_w * phi / (2 * M_PI));
        if (row < 0 || row >= im_h || col < 0 || col >= im_w) {
          ++clamped;
          if (clamp) { row = std::min(std::max(row, 0), im_h - 1); col = std::min(std::max(col, 0), im_w - 1); }
          else { continue; }
        }
        const uint8_t* s = im + (static_cast<size_t>(row) * im_w + col) * 3;
        uint8_t* d = out + (static_cast<size_t>(i) * 6 * S + static_cast<size_t>(face) * S + j) * 3;
        d[0] = s[0]; d[1] = s[1]; d[2] = s[2];
      }
  return clamped;
}

}  // extern "C"
