// TEST HARNESS (tests/ only): the product's HIP-free host code under AddressSanitizer + UndefinedBehaviorSanitizer.
// (GPU sanitizers are not available on this pool; the host-side solvers are header-only and compile with plain g++.)
// Exercises: LmSolver (all three modes, both translation parameterisations) on a synthetic quadratic model,
// ArmijoSearch / hermite_argmin, DepthStageSolver with a synthetic pass, the 8-point host math, the rotation helpers.
// Exit code 0 = no sanitizer report and all sanity checks hold.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>

#include "../../spherical_bundle_adjuster_amd/csrc/sba_depth_solver.hpp"
#include "../../spherical_bundle_adjuster_amd/csrc/sba_epipolar.hpp"
#include "../../spherical_bundle_adjuster_amd/csrc/sba_lm.hpp"
#include "../../spherical_bundle_adjuster_amd/csrc/sba_rotation.hpp"

#define REQUIRE(c) do { if (!(c)) { std::fprintf(stderr, "sanity check failed: %s (line %d)\n", #c, __LINE__); return 1; } } while (0)

int main() {
  std::mt19937_64 rng(7);
  std::normal_distribution<double> N(0.0, 1.0);
  sba_lm_options o;
  sba::lm_default_options(&o);

  // ---- LmSolver on f(x) = 1/2 |A (x - x*)|^2: normal equations H = A^T A, g = H (x - x*)
  for (int mode = 0; mode < 3; ++mode)
    for (int tp = 0; tp < 2; ++tp) {
      double A[36], H[36] = {0}, xs[6];
      for (double& v : A) v = N(rng);
      for (double& v : xs) v = 0.3 * N(rng);
      for (int i = 0; i < 6; ++i)
        for (int j = 0; j < 6; ++j)
          for (int k = 0; k < 6; ++k) H[6 * i + j] += A[6 * k + i] * A[6 * k + j];
      o.tran_param = tp;
      double rot[3] = {0.1, -0.2, 0.05}, tran[3] = {0.5, 0.5, 0.7};
      sba_lm_summary sum;
      const int rc = sba::lm_solve(mode, rot, tran, o, [&](const double* r, const double* t, sba_normal_eq* ne) {
        *ne = sba_normal_eq{};
        double x[6] = {r[0], r[1], r[2], t[0], t[1], t[2]}, dx[6];
        const bool rf = mode != SBA_MODE_TRAN, tf = mode != SBA_MODE_ROT;
        for (int i = 0; i < 6; ++i) dx[i] = ((i < 3 ? rf : tf) ? x[i] - xs[i] : 0.0);
        double cost = 0;
        for (int i = 0; i < 6; ++i) {
          const bool fi = i < 3 ? rf : tf;
          for (int j = 0; j < 6; ++j) {
            const bool fj = j < 3 ? rf : tf;
            if (fi && fj) { ne->H[6 * i + j] = H[6 * i + j]; ne->g[i] += H[6 * i + j] * dx[j]; cost += 0.5 * dx[i] * H[6 * i + j] * dx[j]; }
          }
        }
        ne->cost = cost + 1.0;
        return true;
      }, &sum);
      REQUIRE(rc == SBA_OK && sum.termination >= 1 && sum.termination <= 4 && sum.final_cost <= sum.initial_cost);
    }

  // ---- ArmijoSearch on phi(a) = f0 + g0 a + K a^p
  o.tran_param = 0;
  for (int p = 2; p <= 8; ++p) {
    const double f0 = 3.0, g0 = -1.0, K = std::pow(10.0, p);
    sba::ls::ArmijoSearch s;
    s.start(o, f0, g0, 1.0);
    int guard = 0;
    while (!s.done() && ++guard < 100) {
      const double a = s.query();
      s.feed(f0 + g0 * a + K * std::pow(a, p), g0 + K * p * std::pow(a, p - 1));
    }
    REQUIRE(s.done() && guard < 100);
    if (s.success()) { const double a = s.step_size(); REQUIRE(f0 + g0 * a + K * std::pow(a, p) <= f0 + 1e-4 * g0 * a); }
  }

  // ---- DepthStageSolver with a synthetic separable problem: cost(d) = 1/2 sum (d_i - c_i)^2, exact Newton passes
  {
    std::vector<double> d(40, 3.0), cand(40), target(40);
    for (double& v : target) v = std::fabs(N(rng)) * 2.0;
    sba::DepthStageSolver s;
    s.start(o);
    int guard = 0;
    while (!s.done() && ++guard < 1000) {
      const sba::DepthPassRequest rq = s.request();
      double out[sba::DEPTH_OUT_COUNT] = {0};
      for (size_t i = 0; i < d.size(); ++i) {
        const double g = d[i] - target[i], h = 1.0, sc = 0.5, Hs = sc * h * sc, Gs = sc * g;
        const double y = -Gs / (Hs + std::min(std::max(Hs, 1e-6), 1e32) / rq.radius), delta = sc * y;
        cand[i] = std::max(d[i] + rq.alpha * delta, 0.0);
        out[sba::DEPTH_OUT_COST] += 0.5 * g * g;
        out[sba::DEPTH_OUT_MODEL] += -Gs * y - 0.5 * Hs * y * y;
        out[sba::DEPTH_OUT_CAND_COST] += 0.5 * (cand[i] - target[i]) * (cand[i] - target[i]);
        out[sba::DEPTH_OUT_STEP2] += (cand[i] - d[i]) * (cand[i] - d[i]);
        out[sba::DEPTH_OUT_X2] += d[i] * d[i];
        out[sba::DEPTH_OUT_GDELTA] += g * delta;
        out[sba::DEPTH_OUT_CAND_GDELTA] += (cand[i] - target[i]) * delta;
        out[sba::DEPTH_OUT_GMAX] = std::max(out[sba::DEPTH_OUT_GMAX], std::fabs(d[i] - std::max(d[i] - g, 0.0)));
        out[sba::DEPTH_OUT_DMAX] = std::max(out[sba::DEPTH_OUT_DMAX], std::fabs(delta));
      }
      s.feed(out);
      if (s.take_candidate()) d = cand;
    }
    REQUIRE(s.done() && s.status() == SBA_OK && s.summary().final_cost <= s.summary().initial_cost);
  }

  // ---- 8-point host math on random unit-vector correspondences (incl. an empty and a tiny problem)
  for (int n : {0, 5, 40, 3000}) {
    std::vector<double> groups(static_cast<size_t>(sba::epi::kGroups) * sba::epi::kMom, 0.0);
    for (int i = 0; i < n; ++i) {
      double l[3] = {N(rng), N(rng), N(rng)}, r[3] = {N(rng), N(rng), N(rng)}, a[9];
      const double nl = std::sqrt(l[0] * l[0] + l[1] * l[1] + l[2] * l[2]), nr = std::sqrt(r[0] * r[0] + r[1] * r[1] + r[2] * r[2]);
      for (int p = 0; p < 3; ++p) for (int q = 0; q < 3; ++q) a[3 * p + q] = l[p] / nl * r[q] / nr;
      int k = 0;
      for (int p = 0; p < 9; ++p) for (int q = p; q < 9; ++q) groups[((i / 2) % 64) * sba::epi::kMom + k++] += a[p] * a[q];
    }
    const sba::epi::GuessResult g = sba::epi::initial_guess_from_groups(groups.data(), 80, 0.25, 3, 1);
    REQUIRE(g.num_candidates >= 0 && g.num_candidates <= 160);
  }

  // ---- rotation helpers at awkward angles
  for (double th : {0.0, 1e-9, 1.5e-8, 1e-3, 0.49, 0.51, 3.1, 3.14159265358979}) {
    const double w[3] = {th * 0.6, -th * 0.48, th * 0.64};
    double R[9], G[27], B[9], J[9];
    sba::rotation_and_derivatives(w, R, G);
    sba::rotation_and_derivatives(w, R, nullptr);
    sba::factored_frame(w, B, J);
    double det = R[0] * (R[4] * R[8] - R[5] * R[7]) - R[1] * (R[3] * R[8] - R[5] * R[6]) + R[2] * (R[3] * R[7] - R[4] * R[6]);
    REQUIRE(std::fabs(det - 1.0) < 1e-6);
  }
  std::puts("sanitize_main: ok");
  return 0;
}
