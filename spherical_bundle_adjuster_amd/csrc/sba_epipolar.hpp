// Host side of the 8-point initial guess (reference spherical_bundle_adjuster.cpp:47-181).
//
// The reference runs 80 trials; each draws a random 25 % subset of the matches (random_shuffle on never-seeded
// rand(), .hpp:182-211 -- deterministic per libc, not portable), stacks the rows kron(left_i, right_i) into an
// m x 9 matrix A (.cpp:53-68), takes the right singular vector of the smallest singular value (.cpp:70-74),
// projects the 3x3 E onto rank 2 (.cpp:75-80), decomposes it into R1, R2, t (cv::decomposeEssentialMat, .cpp:85),
// converts both rotations to Euler angles (rot2euler, .cpp:25-45) and keeps those whose largest |angle| is below
// 1.57 (.cpp:101-115); the candidate closest to the others in the 20-80 % trimmed-mean sense wins (.cpp:162-180).
//
// Here the O(N) part is one device pass (sba_epipolar.hip) that accumulates A^T A separately for 64 interleaved
// groups of matches (group = (index / 2) % 64); a trial's subset is the union of a random quarter of the groups, so
// its A^T A is a sum of 16 small matrices and the null vector of A is the eigenvector of the smallest eigenvalue of
// A^T A.  Sampling whole groups instead of single matches is the documented deviation (the reference's own
// sampling cannot be reproduced across C libraries anyway); everything after the null vector follows the reference
// step by step.  All of it is tiny 9x9 / 3x3 host work.
//
// REFERENCE SAMPLING (small problems, round 3): the reference's own subsets can be reproduced.  random_array (.hpp:182-211)
// is std::iota + std::random_shuffle, and libstdc++'s random_shuffle(first, last) is
// `for i in 1..n-1: swap(a[i], a[std::rand() % (i + 1)])` on the process-wide rand() stream, which the reference never
// seeds: glibc's additive-feedback generator in its srand(1) state.  reference_trial_subsets() below draws in that very
// order (80 permutations of match indices, the first int(n * 0.25) entries each, .cpp:130-141) from a stream with
// exactly those values -- by default a PRIVATE re-implementation of glibc's generator (GlibcRand: identical output,
// pinned against the real rand() in tests/test_initial_guess_cpu.py), not the process's own rand(): a process that has
// initialised HIP no longer owns that stream (libhsa-runtime64 imports srand() and rand()), so borrowing it would make the
// subsets depend on what the ROCm runtime happened to draw.  The private stream starts where a never-seeded process
// that has drawn nothing starts, and runs on from call to call like the reference's would across image pairs;
// SBA_GUESS_RAND=libc switches to the process's rand() for integrators who want the matcher's draws (FLANN's kd-trees
// call rand()) to count as they do in the reference.  The device then accumulates A^T A per trial from the index lists
// (sba_epipolar.hip: epipolar_subset_moments_kernel).
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

// The small-matrix work of a trial is __host__ __device__: the host runs it for a single problem and as the cross-check of
// the batch, batch_guess_kernel (sba_epipolar.hip) runs the 80 trials of every pair of a batch on the device.  No FMA
// contraction (as sba_rotation.hpp / sba_lm.hpp), so both compilations round alike.
#ifndef SBA_HD
#if defined(__HIPCC__)
#define SBA_HD __host__ __device__
#else
#define SBA_HD
#endif
#endif
#if defined(__clang__)
#pragma STDC FP_CONTRACT OFF
#endif

namespace sba {
namespace epi {

constexpr int kGroups = 64;     // interleaved groups of matches: group(i) = (i / 2) % 64
constexpr int kMom = 45;        // upper triangle of the 9x9 A^T A

// ---- cyclic Jacobi eigen-decomposition of a symmetric n x n matrix (n <= 9) -------------------------------
// On return: w ascending eigenvalues, V columns = eigenvectors (row-major n x n).
SBA_HD inline void jacobi_eigen(int n, const double* A_in, double* w, double* V) {
  double A[81];
  for (int i = 0; i < n * n; ++i) A[i] = A_in[i];
  for (int i = 0; i < n; ++i)
    for (int j = 0; j < n; ++j) V[i * n + j] = (i == j) ? 1.0 : 0.0;
  for (int sweep = 0; sweep < 64; ++sweep) {
    double off = 0.0, diag = 0.0;
    for (int i = 0; i < n; ++i) {
      diag += A[i * n + i] * A[i * n + i];
      for (int j = i + 1; j < n; ++j) off += A[i * n + j] * A[i * n + j];
    }
    if (off <= 1e-34 * diag || off == 0.0) break;
    for (int p = 0; p < n - 1; ++p)
      for (int q = p + 1; q < n; ++q) {
        const double apq = A[p * n + q];
        if (apq == 0.0) continue;
        const double theta = (A[q * n + q] - A[p * n + p]) / (2.0 * apq);
        const double t = (theta >= 0 ? 1.0 : -1.0) / (std::fabs(theta) + std::sqrt(theta * theta + 1.0));
        const double c = 1.0 / std::sqrt(t * t + 1.0), s = t * c;
        for (int k = 0; k < n; ++k) {   // A <- A G
          const double akp = A[k * n + p], akq = A[k * n + q];
          A[k * n + p] = c * akp - s * akq;
          A[k * n + q] = s * akp + c * akq;
        }
        for (int k = 0; k < n; ++k) {   // A <- G^T A
          const double apk = A[p * n + k], aqk = A[q * n + k];
          A[p * n + k] = c * apk - s * aqk;
          A[q * n + k] = s * apk + c * aqk;
        }
        for (int k = 0; k < n; ++k) {   // V <- V G
          const double vkp = V[k * n + p], vkq = V[k * n + q];
          V[k * n + p] = c * vkp - s * vkq;
          V[k * n + q] = s * vkp + c * vkq;
        }
      }
  }
  int order[9];
  for (int i = 0; i < n; ++i) {            // ascending diagonal, insertion sort (what std::sort does below 16 elements)
    int j = i - 1;
    for (; j >= 0 && A[i * n + i] < A[order[j] * n + order[j]]; --j) order[j + 1] = order[j];
    order[j + 1] = i;
  }
  double Vs[81];
  for (int k = 0; k < n; ++k) {
    w[k] = A[order[k] * n + order[k]];
    for (int i = 0; i < n; ++i) Vs[i * n + k] = V[i * n + order[k]];
  }
  for (int i = 0; i < n * n; ++i) V[i] = Vs[i];
}

// Eigenvector of the SMALLEST eigenvalue of a symmetric positive semi-definite n x n matrix (n <= 9) -- all a trial
// needs from A^T A -- at a fraction of a full Jacobi decomposition (~4 k flops instead of ~50 k):
//   1. Cholesky of A + delta I (delta = 1e-13 trace) and 10 inverse iterations from a fixed start vector,
//   2. Rayleigh-quotient iteration (LU with partial pivoting of A - rho I) until |A x - rho x| <= 1e-15 trace,
//   3. verification: A - (rho - 1e-12 trace) I must be positive definite (its Cholesky succeeds), i.e. no
//      eigenvalue lies below rho -- otherwise the iteration settled on another eigenpair;
//   4. if 2. or 3. fail (close smallest eigenvalues): bisection for lambda_1 with Cholesky as the oracle, then inverse
//      iteration with the shift just below it, verified as in 3.
// Returns false whenever the steps are not conclusive; the caller then runs jacobi_eigen.  The sign is fixed by
// making the largest-magnitude component positive.
SBA_HD inline bool smallest_eigvec(int n, const double* A, double* v, double* lambda) {
  double tr = 0.0;
  for (int i = 0; i < n; ++i) tr += A[i * n + i];
  if (!(tr > 0.0) || !std::isfinite(tr)) return false;
  auto cholesky = [&](double shift, double* L) -> bool {   // A + shift I = L L^T
    for (int i = 0; i < n; ++i)
      for (int j = 0; j <= i; ++j) {
        double s = A[i * n + j] + (i == j ? shift : 0.0);
        for (int k = 0; k < j; ++k) s -= L[i * n + k] * L[j * n + k];
        if (i == j) {
          if (!(s > 0.0)) return false;
          L[i * n + i] = std::sqrt(s);
        } else {
          L[i * n + j] = s / L[j * n + j];
        }
      }
    return true;
  };
  auto normalise = [&](double* x) -> bool {
    double nn = 0.0;
    for (int i = 0; i < n; ++i) nn += x[i] * x[i];
    if (!(nn > 0.0) || !std::isfinite(nn)) return false;
    const double inv = 1.0 / std::sqrt(nn);
    for (int i = 0; i < n; ++i) x[i] *= inv;
    return true;
  };
  double L[81], x[9], y[9];
  if (!cholesky(1e-13 * tr, L)) return false;
  for (int i = 0; i < n; ++i) x[i] = 0.3 + 0.7 * ((0.6180339887498949 * (i + 1)) - std::floor(0.6180339887498949 * (i + 1)));
  normalise(x);
  for (int it = 0; it < 10; ++it) {
    for (int i = 0; i < n; ++i) {               // L z = x
      double t = x[i];
      for (int k = 0; k < i; ++k) t -= L[i * n + k] * y[k];
      y[i] = t / L[i * n + i];
    }
    for (int i = n - 1; i >= 0; --i) {           // L^T y = z
      double t = y[i];
      for (int k = i + 1; k < n; ++k) t -= L[k * n + i] * y[k];
      y[i] = t / L[i * n + i];
    }
    if (!normalise(y)) return false;
    for (int i = 0; i < n; ++i) x[i] = y[i];
  }
  auto rayleigh = [&](const double* q, double* Aq) {
    double rho = 0.0;
    for (int i = 0; i < n; ++i) {
      double t = 0.0;
      for (int k = 0; k < n; ++k) t += A[i * n + k] * q[k];
      Aq[i] = t;
      rho += q[i] * t;
    }
    return rho;
  };
  double Ax[9];
  double rho = rayleigh(x, Ax);
  bool converged = false;
  for (int it = 0; it < 8; ++it) {
    double res = 0.0;
    for (int i = 0; i < n; ++i) res += (Ax[i] - rho * x[i]) * (Ax[i] - rho * x[i]);
    if (std::sqrt(res) <= 1e-15 * tr) { converged = true; break; }
    double B[81], b[9];
    for (int i = 0; i < n; ++i) {
      b[i] = x[i];
      for (int k = 0; k < n; ++k) B[i * n + k] = A[i * n + k] - (i == k ? rho : 0.0);
    }
    for (int c = 0; c < n; ++c) {                // LU with partial pivoting, a vanishing pivot is nudged
      int piv = c;
      for (int r = c + 1; r < n; ++r)
        if (std::fabs(B[r * n + c]) > std::fabs(B[piv * n + c])) piv = r;
      if (piv != c) {
        for (int k = 0; k < n; ++k) { const double tmp = B[c * n + k]; B[c * n + k] = B[piv * n + k]; B[piv * n + k] = tmp; }
        const double tmp = b[c]; b[c] = b[piv]; b[piv] = tmp;
      }
      if (std::fabs(B[c * n + c]) < 1e-300 + 1e-18 * tr) B[c * n + c] = 1e-18 * tr + 1e-300;
      for (int r = c + 1; r < n; ++r) {
        const double f = B[r * n + c] / B[c * n + c];
        for (int k = c; k < n; ++k) B[r * n + k] -= f * B[c * n + k];
        b[r] -= f * b[c];
      }
    }
    for (int i = n - 1; i >= 0; --i) {
      double t = b[i];
      for (int k = i + 1; k < n; ++k) t -= B[i * n + k] * y[k];
      y[i] = t / B[i * n + i];
    }
    if (!normalise(y)) return false;
    for (int i = 0; i < n; ++i) x[i] = y[i];
    rho = rayleigh(x, Ax);
  }
  bool found = converged;
  if (!found) {
    double res = 0.0;
    for (int i = 0; i < n; ++i) res += (Ax[i] - rho * x[i]) * (Ax[i] - rho * x[i]);
    found = std::sqrt(res) <= 1e-13 * tr;
  }
  if (found) found = cholesky(-(rho - 1e-12 * tr), L);   // fails: an eigenvalue below rho -- not the smallest pair
  if (!found) {
    // Second tier.  Subsets with outliers have their smallest eigenvalues close together (lambda_2 / lambda_1 = 1.05 ... 1.6
    // on the synthetic pairs): ten unshifted inverse iterations do not separate them and the Rayleigh iteration then locks on
    // to a neighbour.  A - mu I is positive definite exactly when mu < lambda_1, so Cholesky is a bisection oracle for
    // lambda_1: bracket it between 0 (A + delta I factored above) and the smallest diagonal entry (a Rayleigh quotient), close
    // the bracket to 1e-7 trace, and iterate inversely with the shift just below lambda_1 -- the convergence factor is
    // (lambda_1 - mu) / (lambda_2 - mu).  Same residual test and same verification as the first tier; a matrix whose two
    // smallest eigenvalues agree to ~1e-7 trace still ends in the Jacobi decomposition.
    double lo = -1e-13 * tr, hi = A[0];
    for (int i = 1; i < n; ++i) hi = std::min(hi, A[i * n + i]);
    for (int it = 0; it < 60 && hi - lo > 1e-7 * tr; ++it) {
      const double mid = 0.5 * (lo + hi);
      if (cholesky(-mid, L)) lo = mid; else hi = mid;
    }
    if (!cholesky(-lo, L)) return false;
    for (int i = 0; i < n; ++i) x[i] = 0.3 + 0.7 * ((0.6180339887498949 * (i + 1)) - std::floor(0.6180339887498949 * (i + 1)));
    normalise(x);
    found = false;
    for (int it = 0; it < 24 && !found; ++it) {
      for (int i = 0; i < n; ++i) {               // L z = x
        double t = x[i];
        for (int k = 0; k < i; ++k) t -= L[i * n + k] * y[k];
        y[i] = t / L[i * n + i];
      }
      for (int i = n - 1; i >= 0; --i) {           // L^T y = z
        double t = y[i];
        for (int k = i + 1; k < n; ++k) t -= L[k * n + i] * y[k];
        y[i] = t / L[i * n + i];
      }
      if (!normalise(y)) return false;
      for (int i = 0; i < n; ++i) x[i] = y[i];
      rho = rayleigh(x, Ax);
      double res = 0.0;
      for (int i = 0; i < n; ++i) res += (Ax[i] - rho * x[i]) * (Ax[i] - rho * x[i]);
      found = std::sqrt(res) <= 1e-15 * tr;
    }
    if (!found) return false;
    if (!cholesky(-(rho - 1e-12 * tr), L)) return false;
  }
  int big = 0;
  for (int i = 1; i < n; ++i)
    if (std::fabs(x[i]) > std::fabs(x[big])) big = i;
  const double sgn = x[big] < 0.0 ? -1.0 : 1.0;
  for (int i = 0; i < n; ++i) v[i] = sgn * x[i];
  *lambda = rho;
  return true;
}

SBA_HD inline double det3(const double* M) {
  return M[0] * (M[4] * M[8] - M[5] * M[7]) - M[1] * (M[3] * M[8] - M[5] * M[6]) + M[2] * (M[3] * M[7] - M[4] * M[6]);
}
SBA_HD inline void mul3(const double* A, const double* B, double* C) {
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) C[3 * i + j] = A[3 * i] * B[j] + A[3 * i + 1] * B[3 + j] + A[3 * i + 2] * B[6 + j];
}

// SVD of a 3x3 matrix: E = U diag(w) Vt, w descending, U and Vt orthogonal (row-major).
SBA_HD inline void svd3(const double* E, double* U, double* w, double* Vt) {
  double EtE[9], lam[3], V[9];
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) EtE[3 * i + j] = E[i] * E[j] + E[3 + i] * E[3 + j] + E[6 + i] * E[6 + j];
  jacobi_eigen(3, EtE, lam, V);                 // ascending
  double Vd[9];                                  // columns reordered to descending singular values
  for (int k = 0; k < 3; ++k) {
    w[k] = std::sqrt(std::max(lam[2 - k], 0.0));
    for (int i = 0; i < 3; ++i) Vd[3 * i + k] = V[3 * i + (2 - k)];
  }
  // U columns: u_k = E v_k / w_k for the two large singular values, u_2 = u_0 x u_1
  double u[3][3];
  for (int k = 0; k < 2; ++k) {
    double n2 = 0;
    for (int i = 0; i < 3; ++i) {
      u[k][i] = E[3 * i] * Vd[k] + E[3 * i + 1] * Vd[3 + k] + E[3 * i + 2] * Vd[6 + k];
      n2 += u[k][i] * u[k][i];
    }
    const double inv = n2 > 0 ? 1.0 / std::sqrt(n2) : 0.0;
    for (int i = 0; i < 3; ++i) u[k][i] *= inv;
  }
  // re-orthogonalise u1 against u0 (guards the nearly-equal-singular-value case of an essential matrix)
  double dot = u[0][0] * u[1][0] + u[0][1] * u[1][1] + u[0][2] * u[1][2];
  double n2 = 0;
  for (int i = 0; i < 3; ++i) { u[1][i] -= dot * u[0][i]; n2 += u[1][i] * u[1][i]; }
  for (int i = 0; i < 3; ++i) u[1][i] /= std::sqrt(n2);
  u[2][0] = u[0][1] * u[1][2] - u[0][2] * u[1][1];
  u[2][1] = u[0][2] * u[1][0] - u[0][0] * u[1][2];
  u[2][2] = u[0][0] * u[1][1] - u[0][1] * u[1][0];
  // keep E v_2 = w_2 u_2 consistent in sign
  double ev2[3], s2 = 0;
  for (int i = 0; i < 3; ++i) {
    ev2[i] = E[3 * i] * Vd[2] + E[3 * i + 1] * Vd[5] + E[3 * i + 2] * Vd[8];
    s2 += ev2[i] * u[2][i];
  }
  if (s2 < 0) for (int i = 0; i < 3; ++i) u[2][i] = -u[2][i];
  for (int i = 0; i < 3; ++i)
    for (int k = 0; k < 3; ++k) { U[3 * i + k] = u[k][i]; Vt[3 * k + i] = Vd[3 * i + k]; }
}

// cv::decomposeEssentialMat (OpenCV calib3d, call site .cpp:85): SVD, flip U / Vt to det +1,
// R1 = U W Vt, R2 = U W^T Vt, t = last column of U.
SBA_HD inline void decompose_essential(const double* E, double* R1, double* R2, double* t) {
  double U[9], w[3], Vt[9];
  svd3(E, U, w, Vt);
  if (det3(U) < 0) for (int i = 0; i < 9; ++i) U[i] = -U[i];
  if (det3(Vt) < 0) for (int i = 0; i < 9; ++i) Vt[i] = -Vt[i];
  const double W[9] = {0, 1, 0, -1, 0, 0, 0, 0, 1}, Wt[9] = {0, -1, 0, 1, 0, 0, 0, 0, 1};
  double T[9];
  mul3(U, W, T); mul3(T, Vt, R1);
  mul3(U, Wt, T); mul3(T, Vt, R2);
  t[0] = U[2]; t[1] = U[5]; t[2] = U[8];
}

// rot2euler (.cpp:25-45): single-precision like the reference (float sy, float x/y/z, Vec3f).
SBA_HD inline void rot_to_euler(const double* R, float out[3]) {
  const float sy = static_cast<float>(std::sqrt(R[0] * R[0] + R[3] * R[3]));
  if (!(sy < 1e-6f)) {
    out[0] = static_cast<float>(std::atan2(R[7], R[8]));
    out[1] = static_cast<float>(std::atan2(-R[6], static_cast<double>(sy)));
    out[2] = static_cast<float>(std::atan2(R[3], R[0]));
  } else {
    out[0] = static_cast<float>(std::atan2(-R[5], R[4]));
    out[1] = static_cast<float>(std::atan2(-R[6], static_cast<double>(sy)));
    out[2] = 0.f;
  }
}

// max_vec (.cpp:14-22), including its behaviour on ties.
SBA_HD inline double max_vec(const float v[3]) {
  if (v[0] > v[1] && v[0] > v[2]) return v[0];
  if (v[1] > v[2]) return v[1];
  return v[2];
}

struct Candidate {
  float euler[3];
  float tran[3];
  int trial;
  int which;   // 1 = R1, 2 = R2
};

SBA_HD inline uint64_t splitmix64(uint64_t& s) {
  uint64_t z = (s += 0x9E3779B97F4A7C15ull);
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}

// The groups trial `trial` draws: the first `count` entries of a Fisher-Yates shuffle, seeded by (seed, trial), of the
// `n_items` candidate groups in `items` (nullptr: 0..n_items-1).  A longer draw extends a shorter one.
SBA_HD inline void trial_groups(uint64_t seed, int trial, int count, int* out, const int* items = nullptr, int n_items = kGroups) {
  uint64_t s = seed * 0x2545F4914F6CDD1Dull + static_cast<uint64_t>(trial) * 0xD6E8FEB86659FD93ull + 1;
  int perm[kGroups];
  for (int i = 0; i < n_items; ++i) perm[i] = items ? items[i] : i;
  for (int i = 0; i < count && i < n_items; ++i) {
    const int j = i + static_cast<int>(splitmix64(s) % static_cast<uint64_t>(n_items - i));
    const int tmp = perm[i]; perm[i] = perm[j]; perm[j] = tmp;
    out[i] = perm[i];
  }
}

// One trial from the summed moments (upper triangle, 45 values): E, its rank-2 projection, R1/R2/t, Euler angles.
// rows = number of matches behind the moments when it is known and below 9 (0 = at least 9): the reference takes the LAST
// row of vt from cv::SVDecomp of the rows x 9 matrix A (.cpp:70-73), and with fewer than 9 rows vt has only `rows` rows --
// its last one is the singular vector of the SMALLEST OF THE `rows` singular values, not a null vector of A; in terms of
// A^T A: the eigenvector of the rows-th largest eigenvalue.
SBA_HD inline void trial_from_moments(const double* mom45, float e1[3], float e2[3], float tv[3], bool* v1, bool* v2,
                               double* E_out /* 9, may be null */, int rows = 0) {
  double S[81], w[9], V[81];
  int k = 0;
  for (int a = 0; a < 9; ++a)
    for (int b = a; b < 9; ++b) { S[9 * a + b] = S[9 * b + a] = mom45[k]; ++k; }
  double E[9], lam_min;                                  // eigenvector of the smallest eigenvalue (.cpp:72-74)
  if (rows >= 1 && rows < 9) {
    jacobi_eigen(9, S, w, V);                            // ascending: the rows-th largest eigenvalue sits at index 9 - rows
    for (int i = 0; i < 9; ++i) E[i] = V[9 * i + (9 - rows)];
  } else if (!smallest_eigvec(9, S, E, &lam_min)) {
    jacobi_eigen(9, S, w, V);
    for (int i = 0; i < 9; ++i) E[i] = V[9 * i + 0];
  }
  // rank-2 correction: zero the smallest singular value (.cpp:75-80)
  double U[9], sv[3], Vt[9], T[9], Ec[9];
  svd3(E, U, sv, Vt);
  const double D[9] = {sv[0], 0, 0, 0, sv[1], 0, 0, 0, 0};
  mul3(U, D, T); mul3(T, Vt, Ec);
  if (E_out) for (int i = 0; i < 9; ++i) E_out[i] = Ec[i];
  double R1[9], R2[9], t[3];
  decompose_essential(Ec, R1, R2, t);
  rot_to_euler(R1, e1);
  rot_to_euler(R2, e2);
  for (int i = 0; i < 3; ++i) tv[i] = static_cast<float>(t[i]);
  const float a1[3] = {std::fabs(e1[0]), std::fabs(e1[1]), std::fabs(e1[2])};
  const float a2[3] = {std::fabs(e2[0]), std::fabs(e2[1]), std::fabs(e2[2])};
  *v1 = max_vec(a1) < 1.57;                              // .cpp:103-115
  *v2 = max_vec(a2) < 1.57;
}

// Consensus pick (.cpp:162-180): smallest 20-80 % trimmed mean of the distances to all candidates.
inline int consensus_pick(const std::vector<Candidate>& c) {
  const int r = static_cast<int>(c.size());
  if (r == 0) return -1;
  int best = 0;
  double best_d = 0;
  std::vector<double> d(r);
  const int lo = static_cast<int>(r * 0.2), hi = static_cast<int>(r * 0.8);
  for (int i = 0; i < r; ++i) {
    for (int j = 0; j < r; ++j) {
      const float dx = c[i].euler[0] - c[j].euler[0], dy = c[i].euler[1] - c[j].euler[1], dz = c[i].euler[2] - c[j].euler[2];
      d[j] = std::sqrt(static_cast<double>(dx * dx + dy * dy + dz * dz));
    }
    // only the 20-80 % middle is summed: select it, then sort just that part (same summation order as a full sort)
    if (lo < r) std::nth_element(d.begin(), d.begin() + lo, d.end());
    if (hi < r) std::nth_element(d.begin() + lo, d.begin() + hi, d.end());
    std::sort(d.begin() + lo, d.begin() + hi);
    double acc = 0.0;
    for (int j = lo; j < hi; ++j) acc += d[j];
    const double avg = acc / (static_cast<double>(hi - lo) * 1.0);    // 0/0 = NaN for r < 2, as in the reference
    if (i == 0 || avg < best_d) { best = i; best_d = avg; }          // std::min_element keeps the first minimum
  }
  return best;
}

struct GuessResult {
  float euler[3] = {0, 0, 0};   // R_vec_out
  float tran[3] = {0, 0, 0};    // T_vec_out
  int num_candidates = 0;
  int picked = -1;
  std::vector<Candidate> candidates;
};

// Occupancy of the groups: a row of A is kron(left, right) of two unit vectors, so it has unit length and the trace of
// a group's A^T A IS its number of matches.  With fewer than 256 matches some of the 64 groups are empty, and a blind
// draw of 16 groups could sum fewer than the 8 correspondences the linear system needs (the reference always takes
// floor(n / 4) individual matches, .cpp:132-141): draw among the NON-EMPTY groups -- a quarter of them, like the
// reference's quarter of the matches -- and keep drawing until the subset holds at least 8 matches (or every group).
// From 256 matches on every group is occupied and holds at least 4: exactly 16 groups.
struct GroupOccupancy {
  int nonempty[kGroups], count[kGroups], ne, take;
};
SBA_HD inline void group_occupancy(const double* groups, double fraction, GroupOccupancy* o) {
  int diag = 0;
  int diag_index[9];
  for (int a = 0; a < 9; ++a) { diag_index[a] = diag; diag += 9 - a; }
  o->ne = 0;
  for (int g = 0; g < kGroups; ++g) {
    double tr = 0;
    for (int a = 0; a < 9; ++a) tr += groups[g * kMom + diag_index[a]];
    o->count[g] = std::isfinite(tr) && tr > 0.5 ? static_cast<int>(std::floor(tr + 0.5)) : 0;
    if (o->count[g] > 0) o->nonempty[o->ne++] = g;
  }
  o->take = std::max(1, std::min(o->ne, static_cast<int>(o->ne * fraction)));
}

// Trial `trial` of a problem given by its group moments: draw the groups, sum their moments in ascending group order, solve.
struct TrialOut { Candidate c1, c2; bool v1, v2; };
SBA_HD inline void group_trial(const double* groups, const GroupOccupancy& occ, uint64_t seed, int trial, TrialOut* out) {
  TrialOut& o = *out;
  o.c1 = Candidate{}; o.c2 = Candidate{}; o.v1 = o.v2 = false;
  if (occ.ne == 0) return;
  int sel[kGroups];
  trial_groups(seed, trial, occ.ne, sel, occ.nonempty, occ.ne);
  int used = occ.take, matches = 0;
  for (int s = 0; s < used; ++s) matches += occ.count[sel[s]];
  while (matches < 8 && used < occ.ne) matches += occ.count[sel[used++]];
  for (int i = 1; i < used; ++i) {                  // ascending: fixed summation order
    const int v = sel[i];
    int j = i - 1;
    for (; j >= 0 && sel[j] > v; --j) sel[j + 1] = sel[j];
    sel[j + 1] = v;
  }
  double mom[kMom];
  for (int k = 0; k < kMom; ++k) mom[k] = 0.0;
  for (int s = 0; s < used; ++s)
    for (int k = 0; k < kMom; ++k) mom[k] += groups[sel[s] * kMom + k];
  float tv[3];
  trial_from_moments(mom, o.c1.euler, o.c2.euler, tv, &o.v1, &o.v2, nullptr);
  for (int i = 0; i < 3; ++i) o.c1.tran[i] = o.c2.tran[i] = tv[i];
  o.c1.trial = o.c2.trial = trial; o.c1.which = 1; o.c2.which = 2;
}

// groups: [64][45] moments.  trials / fraction / seed: 80, 0.25 in the reference.
// Trials are independent (the subset of trial k depends on (seed, k) only), so they run on `threads` host threads --
// what the reference's set_omp(num_proc) does for its loops -- and are collected in trial order: the result does not
// depend on the thread count.
inline GuessResult initial_guess_from_groups(const double* groups, int trials, double fraction, uint64_t seed,
                                             int threads = 1) {
  GuessResult res;
  GroupOccupancy occ;
  group_occupancy(groups, fraction, &occ);
  std::vector<TrialOut> out(static_cast<size_t>(std::max(trials, 0)));
  auto run = [&](int first, int last) {
    for (int trial = first; trial < last; ++trial) group_trial(groups, occ, seed, trial, &out[trial]);
  };
  // A trial is ~6 us and starting a thread costs tens of us: threads only pay with hundreds of trials each, so the
  // reference's 80 trials always run serially.
  const int nt = std::max(1, std::min(threads, trials / 256));
  if (nt <= 1) {
    run(0, trials);
  } else {
    std::vector<std::thread> pool;
    const int per = (trials + nt - 1) / nt;
    for (int t = 1; t < nt; ++t) pool.emplace_back(run, std::min(trials, t * per), std::min(trials, (t + 1) * per));
    run(0, std::min(trials, per));
    for (auto& th : pool) th.join();
  }
  for (int trial = 0; trial < trials; ++trial) {
    if (out[trial].v1) res.candidates.push_back(out[trial].c1);   // .cpp:148-157: T_vec is pushed with either rotation
    if (out[trial].v2) res.candidates.push_back(out[trial].c2);
  }
  res.num_candidates = static_cast<int>(res.candidates.size());
  res.picked = consensus_pick(res.candidates);
  if (res.picked >= 0) {
    std::memcpy(res.euler, res.candidates[res.picked].euler, sizeof(res.euler));
    std::memcpy(res.tran, res.candidates[res.picked].tran, sizeof(res.tran));
  }
  return res;
}

// ---- the reference's own subsets (random_array, .hpp:182-211; initial_guess, .cpp:130-141) -------------------------------
// sample_n = int(match_size * 0.25) like `int sample_n = match_size*0.25;` (.cpp:133).
inline int reference_sample_size(int n, double fraction = 0.25) { return static_cast<int>(n * fraction); }

// glibc's rand() / random() -- TYPE_3, the default: r[i] = r[i-3] + r[i-31] over 32-bit words, output r >> 1, state filled from
// the seed by the Lehmer generator 16807 x mod (2^31 - 1) and warmed up by 310 discarded steps (glibc stdlib/random_r.c:
// __srandom_r, __random_r) -- restated so that the library owns a stream with the reference's values.  seed 1 = the state of a
// process that never called srand().
class GlibcRand {
 public:
  explicit GlibcRand(unsigned seed = 1) { reseed(seed); }
  void reseed(unsigned seed) {
    if (seed == 0) seed = 1;
    int32_t word = static_cast<int32_t>(seed);
    r_[0] = static_cast<uint32_t>(word);
    for (int i = 1; i < 31; ++i) {
      const long hi = word / 127773, lo = word % 127773;           // 16807 * word mod 2147483647 without overflow
      long w = 16807 * lo - 2836 * hi;
      if (w < 0) w += 2147483647;
      word = static_cast<int32_t>(w);
      r_[i] = static_cast<uint32_t>(word);
    }
    f_ = 3; b_ = 0;
    for (int i = 0; i < 310; ++i) (void)next();
  }
  int next() {                                                       // what rand() returns: 0 .. RAND_MAX = 2^31 - 1
    const uint32_t sum = r_[f_] + r_[b_];
    r_[f_] = sum;
    f_ = f_ == 30 ? 0 : f_ + 1;
    b_ = b_ == 30 ? 0 : b_ + 1;
    return static_cast<int>(sum >> 1);
  }

 private:
  uint32_t r_[31];
  int f_ = 3, b_ = 0;
};

// out[trial * sample_n + k] = the k-th index trial `trial` draws.  Consumes (n - 1) values of `next_rand` per trial, in the
// order libstdc++'s std::random_shuffle(first, last) does:
//     for (i = first + 1; i != last; ++i) { j = first + std::rand() % ((i - first) + 1); if (i != j) iter_swap(i, j); }
// -- a fresh permutation of 0..n-1 per trial (random_array's constructor), its first sample_n entries used.
template <typename NextRand>
inline void reference_trial_subsets(int n, int trials, double fraction, int* out, NextRand next_rand) {
  const int sample_n = reference_sample_size(n, fraction);
  std::vector<int> perm(static_cast<size_t>(std::max(n, 0)));
  for (int trial = 0; trial < trials; ++trial) {
    for (int i = 0; i < n; ++i) perm[i] = i;                                  // std::iota
    int* a = perm.data();
    for (int i = 1; i < n; ++i) {                                             // std::random_shuffle, libstdc++
      const unsigned j = static_cast<unsigned>(next_rand()) % static_cast<unsigned>(i + 1);   // rand() >= 0: same as the int %
      const int vi = a[i], vj = a[j];       // i == j: swapping an element with itself is the identity
      a[i] = vj; a[j] = vi;
    }
    for (int k = 0; k < sample_n; ++k) out[static_cast<size_t>(trial) * sample_n + k] = perm[k];
  }
}

// The trials' own A^T A (moments: [trials][45], each over `rows` matches) -> candidates -> consensus, as
// initial_guess_from_groups does for group sums.
inline GuessResult initial_guess_from_trial_moments(const double* moments, int trials, int rows) {
  GuessResult res;
  for (int trial = 0; trial < trials; ++trial) {
    Candidate c1{}, c2{};
    bool v1 = false, v2 = false;
    float tv[3];
    trial_from_moments(moments + static_cast<size_t>(trial) * kMom, c1.euler, c2.euler, tv, &v1, &v2, nullptr, rows);
    for (int i = 0; i < 3; ++i) c1.tran[i] = c2.tran[i] = tv[i];
    c1.trial = c2.trial = trial; c1.which = 1; c2.which = 2;
    if (v1) res.candidates.push_back(c1);          // .cpp:148-157
    if (v2) res.candidates.push_back(c2);
  }
  res.num_candidates = static_cast<int>(res.candidates.size());
  res.picked = consensus_pick(res.candidates);
  if (res.picked >= 0) {
    std::memcpy(res.euler, res.candidates[res.picked].euler, sizeof(res.euler));
    std::memcpy(res.tran, res.candidates[res.picked].tran, sizeof(res.tran));
  }
  return res;
}

}  // namespace epi
}  // namespace sba
