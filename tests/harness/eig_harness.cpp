// CPU harness for csrc/sba_epipolar.hpp's smallest_eigvec (tests/test_initial_guess_cpu.py): symmetric positive semi-definite
// 9 x 9 matrices with a prescribed spectrum -- well separated, CLUSTERED at the bottom (what subsets with outliers produce:
// lambda_2 / lambda_1 = 1.05 ... 1.6), nearly degenerate, rank-deficient -- against the cyclic Jacobi decomposition.
// Prints one line per case: name, matrices, fall-backs to Jacobi, worst |v - v_jacobi|, worst |lambda - lambda_jacobi| / trace.
#include <cstdio>
#include <random>
#include "sba_epipolar.hpp"

using namespace sba::epi;

static void random_orthogonal(std::mt19937_64& rng, double* Q) {      // Gram-Schmidt of a Gaussian matrix
  std::normal_distribution<double> nd;
  for (int i = 0; i < 81; ++i) Q[i] = nd(rng);
  for (int c = 0; c < 9; ++c) {
    for (int p = 0; p < c; ++p) {
      double dot = 0;
      for (int r = 0; r < 9; ++r) dot += Q[9 * r + c] * Q[9 * r + p];
      for (int r = 0; r < 9; ++r) Q[9 * r + c] -= dot * Q[9 * r + p];
    }
    double nn = 0;
    for (int r = 0; r < 9; ++r) nn += Q[9 * r + c] * Q[9 * r + c];
    nn = std::sqrt(nn);
    for (int r = 0; r < 9; ++r) Q[9 * r + c] /= nn;
  }
}

int main() {
  struct Case { const char* name; double l1, ratio; } cases[] = {
      {"separated", 1e-6, 1e3}, {"clustered_1.6", 5e-3, 1.6}, {"clustered_1.05", 7e-3, 1.05}, {"clustered_1.01", 7e-3, 1.01},
      {"near_degenerate_1e-6", 7e-3, 1.000001}, {"rank_deficient", 0.0, 0.0}};
  std::mt19937_64 rng(11);
  std::uniform_real_distribution<double> ud(0.0, 1.0);
  for (const Case& c : cases) {
    int fallbacks = 0, count = 300;
    double worst_v = 0, worst_l = 0;
    for (int m = 0; m < count; ++m) {
      double Q[81], w[9];
      random_orthogonal(rng, Q);
      w[0] = c.l1;
      w[1] = c.ratio > 0 ? c.l1 * c.ratio : 3e-3;
      for (int i = 2; i < 9; ++i) w[i] = w[i - 1] * (1.0 + 0.8 * ud(rng)) + 1e-3;
      double A[81], tr = 0;
      for (int i = 0; i < 9; ++i)
        for (int j = 0; j < 9; ++j) {
          double s = 0;
          for (int k = 0; k < 9; ++k) s += Q[9 * i + k] * w[k] * Q[9 * j + k];
          A[9 * i + j] = s;
        }
      for (int i = 0; i < 9; ++i) for (int j = 0; j < i; ++j) A[9 * i + j] = A[9 * j + i];     // exactly symmetric
      for (int i = 0; i < 9; ++i) tr += A[10 * i];
      double v[9], lam, wj[9], V[81];
      jacobi_eigen(9, A, wj, V);
      if (!smallest_eigvec(9, A, v, &lam)) { ++fallbacks; continue; }
      double dp = 0, dm = 0;
      for (int i = 0; i < 9; ++i) { dp = std::max(dp, std::fabs(v[i] - V[9 * i])); dm = std::max(dm, std::fabs(v[i] + V[9 * i])); }
      worst_v = std::max(worst_v, std::min(dp, dm));
      worst_l = std::max(worst_l, std::fabs(lam - wj[0]) / tr);
    }
    std::printf("%s %d %d %.3e %.3e\n", c.name, count, fallbacks, worst_v, worst_l);
  }
  return 0;
}
