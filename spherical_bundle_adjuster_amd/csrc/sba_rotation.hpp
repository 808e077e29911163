// Host-side angle-axis rotation, its parameter derivatives, and the map from the device's moment
// pack to the normal equations.
//
// The reference rotates every match with ceres::AngleAxisRotatePoint inside the functor
// (spherical_bundle_adjuster.cpp:857, :908, :965, :1019) and lets autodiff differentiate it, i.e.
// sin/cos/sqrt of the SAME angle are recomputed for every match.  The rotation vector is shared by all
// matches, so everything that depends only on it is built once per sweep on the host:
//
//   explicit-Jacobian kernel:   R(w) and G_j = dR/dw_j  ->  d(R p)/dw = [G_0 p | G_1 p | G_2 p]
//   factored kernel:            d(R p)/dw = -[R p]x J_l(w)   (J_l = left Jacobian of SO(3)),
//                               so with v = -d1 R x1 (already needed for the residual)
//                                   A = d e / d w = -[v]x J_l
//                               and  sum w A^T A = J_l^T (tr(M) I - M) J_l,  M = sum w v v^T
//                                    sum w A^T   = J_l^T [m]x,               m = sum w v
//                                    sum w A^T e = J_l^T vee(C - C^T),       C = sum w v e^T
//                               The device accumulates only M, m, C (no per-match Jacobian entries).
//
// Small-angle branch: AngleAxisRotatePoint switches to  p + w x p  when w.w <= DBL_EPSILON, whose
// derivative is -[p]x exactly.  Then R = I + [w]x, and A = -[x]x with x = -d1 x1 = B v,
// B = (I + [w]x)^-1: the same moments serve after the linear change of variables v -> B v, with
// J = I in place of J_l.
#pragma once
#include <cfloat>
#include <cmath>

// The same source serves the host (single-problem path, g++-built test harness) and the device (the batched path
// prepares every pair's state and converts every pair's moments in its own kernels).
#if defined(__HIPCC__)
#define SBA_HD __host__ __device__
#else
#define SBA_HD
#endif

// No FMA contraction in here: the per-sweep state and the moment -> normal-equation map are evaluated in several
// compilations (host g++, host clang, several device kernels with the flags known at compile time or not); with
// contraction left to the optimiser their last bits differ from one instantiation to the next.
#if defined(__clang__)
#pragma STDC FP_CONTRACT OFF
#endif

namespace sba {

// a = sin(th)/th, b = (1-cos th)/th^2, c = (th - sin th)/th^3, ap = (da/dth)/th, bp = (db/dth)/th.
// Power series in x = th^2 below 0.25 (the closed forms cancel catastrophically for small angles:
// bp loses ~2 eps / x^2), closed forms above, where they are accurate to a few eps.
//   a = sum (-x)^k/(2k+1)!   b = sum (-x)^k/(2k+2)!   c = sum (-x)^k/(2k+3)!
//   bp = (a - 2b)/x = sum_{k>=1} (-1)^k 2k x^(k-1)/(2k+2)!
SBA_HD inline void so3_coefficients(double x /* th^2 */, double* a, double* b, double* c, double* ap, double* bp) {
  if (x < 0.25) {
    // reciprocal factorials 1/(2k+1)!, 1/(2k+2)!, 1/(2k+3)! and the bp coefficients 2k/(2k+2)! as constants: the series
    // cost 4 multiply-adds per term instead of 4 divisions (this runs once per LM iteration per pair on ONE device
    // thread in batch_lm_kernel, where a double division is ~30 instructions)
    constexpr double kF1[11] = {1.0, 1.0 / 6, 1.0 / 120, 1.0 / 5040, 1.0 / 362880, 1.0 / 39916800, 1.0 / 6227020800.0,
                                1.0 / 1307674368000.0, 1.0 / 355687428096000.0, 1.0 / 121645100408832000.0,
                                1.0 / 51090942171709440000.0};
    constexpr double kF2[11] = {1.0 / 2, 1.0 / 24, 1.0 / 720, 1.0 / 40320, 1.0 / 3628800, 1.0 / 479001600, 1.0 / 87178291200.0,
                                1.0 / 20922789888000.0, 1.0 / 6402373705728000.0, 1.0 / 2432902008176640000.0,
                                1.0 / 1124000727777607680000.0};
    constexpr double kF3[11] = {1.0 / 6, 1.0 / 120, 1.0 / 5040, 1.0 / 362880, 1.0 / 39916800, 1.0 / 6227020800.0,
                                1.0 / 1307674368000.0, 1.0 / 355687428096000.0, 1.0 / 121645100408832000.0,
                                1.0 / 51090942171709440000.0, 1.0 / 25852016738884976640000.0};
    double sa = 0, sb = 0, sc = 0, sbp = 0;
    double pw = 1.0;      // (-x)^k
    double pwm = -1.0;    // (-1)^k x^(k-1), from k = 1
    for (int k = 0; k < 11; ++k) {
      sa += pw * kF1[k];
      sb += pw * kF2[k];
      sc += pw * kF3[k];
      if (k > 0) { sbp += pwm * ((2.0 * k) * kF2[k]); pwm *= -x; }
      pw *= -x;
    }
    *a = sa; *b = sb; *c = sc; *bp = sbp;
  } else {
    const double th = std::sqrt(x), s = std::sin(th), co = std::cos(th);
    *a = s / th;
    *b = (1.0 - co) / x;
    *c = (th - s) / (x * th);
    *bp = (*a - 2.0 * *b) / x;
  }
  *ap = *c - *b;   // (th cos th - sin th)/th^3 = c - b exactly
}

// R: row-major 3x3.  G: G[9*j + 3*r + c] = d R[r][c] / d w_j  (nullptr: not wanted -- the factored kernel needs R only).
SBA_HD inline void rotation_and_derivatives(const double w[3], double R[9], double* G) {
  const double th2 = w[0] * w[0] + w[1] * w[1] + w[2] * w[2];
  const double E[3][9] = {{0, 0, 0, 0, 0, -1, 0, 1, 0},     // [e_j]x, row-major
                                 {0, 0, 1, 0, 0, 0, -1, 0, 0},
                                 {0, -1, 0, 1, 0, 0, 0, 0, 0}};
  const double W[9] = {0, -w[2], w[1], w[2], 0, -w[0], -w[1], w[0], 0};
  if (th2 > DBL_EPSILON) {
    double a, b, c, ap, bp;
    so3_coefficients(th2, &a, &b, &c, &ap, &bp);
    // R = I + a [w]x + b ([w]x)^2 = I + a W + b (w w^T - th2 I)
    for (int r = 0; r < 3; ++r)
      for (int k = 0; k < 3; ++k)
        R[3 * r + k] = (r == k ? 1.0 : 0.0) + a * W[3 * r + k] + b * (w[r] * w[k] - (r == k ? th2 : 0.0));
    if (G == nullptr) return;
    for (int j = 0; j < 3; ++j)
      for (int r = 0; r < 3; ++r)
        for (int k = 0; k < 3; ++k) {
          const double I = (r == k) ? 1.0 : 0.0;
          const double ejw = (r == j ? w[k] : 0.0) + (k == j ? w[r] : 0.0);  // e_j w^T + w e_j^T
          G[9 * j + 3 * r + k] = ap * w[j] * W[3 * r + k] + a * E[j][3 * r + k] +
                                 bp * w[j] * (w[r] * w[k] - th2 * I) + b * (ejw - 2.0 * w[j] * I);
        }
  } else {
    for (int r = 0; r < 3; ++r)
      for (int k = 0; k < 3; ++k) R[3 * r + k] = (r == k ? 1.0 : 0.0) + W[3 * r + k];
    if (G == nullptr) return;
    for (int j = 0; j < 3; ++j)
      for (int i = 0; i < 9; ++i) G[9 * j + i] = E[j][i];
  }
}

// For the factored kernel: A = -[B v]x J with v = -d1 R x1.
//   th2 >  eps: B = I,              J = J_l(w) = I + b [w]x + c ([w]x)^2
//   th2 <= eps: B = (I + [w]x)^-1,  J = I
SBA_HD inline void factored_frame(const double w[3], double B[9], double J[9]) {
  const double th2 = w[0] * w[0] + w[1] * w[1] + w[2] * w[2];
  const double W[9] = {0, -w[2], w[1], w[2], 0, -w[0], -w[1], w[0], 0};
  for (int i = 0; i < 9; ++i) B[i] = J[i] = (i % 4 == 0) ? 1.0 : 0.0;
  if (th2 > DBL_EPSILON) {
    double a, b, c, ap, bp;
    so3_coefficients(th2, &a, &b, &c, &ap, &bp);
    for (int r = 0; r < 3; ++r)
      for (int k = 0; k < 3; ++k)
        J[3 * r + k] += b * W[3 * r + k] + c * (w[r] * w[k] - (r == k ? th2 : 0.0));
  } else {
    // (I + W)^-1 = (I - W + w w^T) / (1 + th2)   for skew W = [w]x
    const double s = 1.0 / (1.0 + th2);
    for (int r = 0; r < 3; ++r)
      for (int k = 0; k < 3; ++k) B[3 * r + k] = s * ((r == k ? 1.0 : 0.0) - W[3 * r + k] + w[r] * w[k]);
  }
}

// Device moment pack (factored kernel) -> normal-equation pack (SBA_PACK_* layout of sba_hip.h).
//   moment: [0..5] M upper 00 01 02 11 12 22, [6..14] C[k][l] = sum w v_k e_l, [15] sum w,
//           [16..18] m = sum w v, [19..21] sum w e, [22] cost, [23] n_outlier
SBA_HD inline void moments_to_normal_pack(bool rot_free, bool tran_free, const double B[9], const double J[9],
                                   const double* mom, double* pack) {
  for (int i = 0; i < 24; ++i) pack[i] = 0.0;
  pack[15] = mom[15];
  for (int i = 19; i < 24; ++i) pack[i] = mom[i];
  if (!rot_free) return;
  const double M[9] = {mom[0], mom[1], mom[2], mom[1], mom[3], mom[4], mom[2], mom[4], mom[5]};
  double BM[9], Mv[9], Cv[9], mv[3];
  for (int r = 0; r < 3; ++r)
    for (int k = 0; k < 3; ++k) {
      BM[3 * r + k] = B[3 * r] * M[k] + B[3 * r + 1] * M[3 + k] + B[3 * r + 2] * M[6 + k];
      Cv[3 * r + k] = B[3 * r] * mom[6 + k] + B[3 * r + 1] * mom[9 + k] + B[3 * r + 2] * mom[12 + k];
    }
  for (int r = 0; r < 3; ++r) {
    for (int k = 0; k < 3; ++k)
      Mv[3 * r + k] = BM[3 * r] * B[3 * k] + BM[3 * r + 1] * B[3 * k + 1] + BM[3 * r + 2] * B[3 * k + 2];
    mv[r] = B[3 * r] * mom[16] + B[3 * r + 1] * mom[17] + B[3 * r + 2] * mom[18];
  }
  // S = tr(Mv) I - Mv ;  H_aa = J^T S J
  const double tr = Mv[0] + Mv[4] + Mv[8];
  double S[9], SJ[9];
  for (int i = 0; i < 9; ++i) S[i] = (i % 4 == 0 ? tr : 0.0) - Mv[i];
  for (int r = 0; r < 3; ++r)
    for (int k = 0; k < 3; ++k) SJ[3 * r + k] = S[3 * r] * J[k] + S[3 * r + 1] * J[3 + k] + S[3 * r + 2] * J[6 + k];
  int idx = 0;
  for (int a = 0; a < 3; ++a)
    for (int b2 = a; b2 < 3; ++b2)
      pack[idx++] = J[a] * SJ[b2] + J[3 + a] * SJ[3 + b2] + J[6 + a] * SJ[6 + b2];
  // g_a = J^T vee(Cv - Cv^T)
  const double vc[3] = {Cv[5] - Cv[7], Cv[6] - Cv[2], Cv[1] - Cv[3]};
  for (int a = 0; a < 3; ++a) pack[16 + a] = J[a] * vc[0] + J[3 + a] * vc[1] + J[6 + a] * vc[2];
  if (tran_free) {
    // H_at = J^T [mv]x   (row = rot index, col = tran index)
    const double X[9] = {0, -mv[2], mv[1], mv[2], 0, -mv[0], -mv[1], mv[0], 0};
    for (int a = 0; a < 3; ++a)
      for (int c2 = 0; c2 < 3; ++c2)
        pack[6 + 3 * a + c2] = J[a] * X[c2] + J[3 + a] * X[3 + c2] + J[6 + a] * X[6 + c2];
  }
}

}  // namespace sba

#if defined(__clang__)
#pragma STDC FP_CONTRACT DEFAULT
#endif
