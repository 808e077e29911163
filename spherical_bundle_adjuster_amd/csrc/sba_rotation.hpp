// Host-side angle-axis rotation and its parameter derivatives.
//
// The reference rotates every match with ceres::AngleAxisRotatePoint inside the functor
// (spherical_bundle_adjuster.cpp:857, :908, :965, :1019) and lets autodiff differentiate it,
// i.e. sin/cos/sqrt of the SAME angle are recomputed for every match.  The rotation vector is
// shared by all matches, so here R(w) and the three constant matrices G_j = dR/dw_j are built
// once per sweep on the host; the device then needs no transcendental per match:
//      R(w) p           -> residual
//      [G_0 p G_1 p G_2 p] -> d(R p)/dw   (what Jet<double,3> would carry)
//
// Small-angle branch: AngleAxisRotatePoint switches to  p + w x p  when w.w <= DBL_EPSILON;
// differentiating that branch gives G_j = [e_j]x exactly, which is what is returned here.
#pragma once
#include <cfloat>
#include <cmath>

namespace sba {

// R: row-major 3x3.  G: G[9*j + 3*r + c] = d R[r][c] / d w_j.
inline void rotation_and_derivatives(const double w[3], double R[9], double G[27]) {
  const double th2 = w[0] * w[0] + w[1] * w[1] + w[2] * w[2];
  // [e_j]x, row-major
  static const double E[3][9] = {{0, 0, 0, 0, 0, -1, 0, 1, 0},
                                 {0, 0, 1, 0, 0, 0, -1, 0, 0},
                                 {0, -1, 0, 1, 0, 0, 0, 0, 0}};
  const double W[9] = {0, -w[2], w[1], w[2], 0, -w[0], -w[1], w[0], 0};
  if (th2 > DBL_EPSILON) {
    const double th = std::sqrt(th2);
    const double s = std::sin(th), c = std::cos(th);
    const double a = s / th;                 // sin(th)/th
    const double b = (1.0 - c) / th2;        // (1-cos(th))/th^2
    const double ap = (c - a) / th2;         // (da/dth)/th
    const double bp = (a - 2.0 * b) / th2;   // (db/dth)/th
    // R = I + a [w]x + b ([w]x)^2 = I + a W + b (w w^T - th2 I)
    for (int r = 0; r < 3; ++r)
      for (int k = 0; k < 3; ++k)
        R[3 * r + k] = (r == k ? 1.0 : 0.0) + a * W[3 * r + k] +
                       b * (w[r] * w[k] - (r == k ? th2 : 0.0));
    for (int j = 0; j < 3; ++j)
      for (int r = 0; r < 3; ++r)
        for (int k = 0; k < 3; ++k) {
          const double I = (r == k) ? 1.0 : 0.0;
          const double wwT = w[r] * w[k];
          const double ejw = (r == j ? w[k] : 0.0) + (k == j ? w[r] : 0.0);  // e_j w^T + w e_j^T
          G[9 * j + 3 * r + k] = ap * w[j] * W[3 * r + k] + a * E[j][3 * r + k] +
                                 bp * w[j] * (wwT - th2 * I) + b * (ejw - 2.0 * w[j] * I);
        }
  } else {
    for (int r = 0; r < 3; ++r)
      for (int k = 0; k < 3; ++k) R[3 * r + k] = (r == k ? 1.0 : 0.0) + W[3 * r + k];
    for (int j = 0; j < 3; ++j)
      for (int i = 0; i < 9; ++i) G[9 * j + i] = E[j][i];
  }
}

}  // namespace sba
