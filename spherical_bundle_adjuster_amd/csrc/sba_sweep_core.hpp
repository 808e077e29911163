// Device-side core of the residual + Jacobian sweep, shared by the single-problem kernels (sba_kernels.hip) and the
// batched kernels (sba_batch_kernels.hip): accumulator <-> pack slot maps, the DPP wave reduction, the 16-byte-per-lane vector
// loads with register double buffering, the Huber weight, and the per-correspondence accumulation (factored /
// explicit).  Everything is a template or a forceinline device function in an unnamed namespace: each translation unit
// gets its own copy, nothing is exported.
#pragma once
#include "sba_device.hpp"

#ifndef SBA_PARAMS_IN_LDS
#define SBA_PARAMS_IN_LDS 0
#endif
#ifndef SBA_NT_LOADS
#define SBA_NT_LOADS 1     // the once-read coordinate stream is loaded non-temporally (global_load ... nt):
                           // measured +10-12 % sweep bandwidth on MI355X (profiles/r01_tune_variants_10M.log)
#endif

namespace sba {
namespace {

constexpr int MODE_ROT = 0, MODE_TRAN = 1, MODE_RT = 2;
constexpr int DEPTH_UNIFORM = 0, DEPTH_PER_MATCH = 1;
constexpr int KIND_FACTORED = 0, KIND_EXPLICIT = 1;

// ---- accumulator <-> pack slot maps -------------------------------------------------------
// explicit pack = SBA_PACK_* of sba_hip.h; moment pack: [0..5] M, [6..14] C, [15] sw, [16..18] m,
// [19..21] sum w e, [22] cost, [23] n_outlier.  Slots 15 and 19..23 mean the same in both.
template <int MODE, int KIND> struct AccMap;
template <> struct AccMap<MODE_ROT, KIND_EXPLICIT> {   // haa[6] ga[3] cost nout
  static constexpr int N = 11;
  __host__ __device__ static constexpr int slot(int k) {
    return k < 6 ? k : (k < 9 ? 16 + (k - 6) : (k == 9 ? 22 : 23));
  }
};
template <> struct AccMap<MODE_ROT, KIND_FACTORED> {   // M[6] C[9] cost nout
  static constexpr int N = 17;
  __host__ __device__ static constexpr int slot(int k) { return k < 15 ? k : (k == 15 ? 22 : 23); }
};
template <int KIND> struct AccMap<MODE_TRAN, KIND> {   // sw gt[3] cost nout
  static constexpr int N = 6;
  __host__ __device__ static constexpr int slot(int k) {
    return k == 0 ? 15 : (k < 4 ? 19 + (k - 1) : (k == 4 ? 22 : 23));
  }
};
template <int KIND> struct AccMap<MODE_RT, KIND> {     // the full pack, either layout
  static constexpr int N = 24;
  __host__ __device__ static constexpr int slot(int k) { return k; }
};

// ---- wave64 sum via DPP; the total ends up in lane 63 --------------------------------------
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double dpp_f64(double v) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, ROW_MASK, 0xf, false);
  hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, ROW_MASK, 0xf, false);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double wave_sum_to_lane63(double v) {
  v += dpp_f64<0xB1, 0xf>(v);   // quad_perm [1,0,3,2]
  v += dpp_f64<0x4E, 0xf>(v);   // quad_perm [2,3,0,1]
  v += dpp_f64<0x141, 0xf>(v);  // row_half_mirror
  v += dpp_f64<0x140, 0xf>(v);  // row_mirror            -> every lane: its 16-lane row sum
  v += dpp_f64<0x142, 0xa>(v);  // row_bcast15 into rows 1,3
  v += dpp_f64<0x143, 0xc>(v);  // row_bcast31 into rows 2,3 -> lane 63: wave sum
  return v;
}

// ---- one 16-byte vector of correspondences per lane: 2 points (f64 planes) or 4 (f32 planes) ---
template <typename ST> struct Lanes;
template <> struct Lanes<double> { static constexpr int PPT = 2; typedef double2 vec; };
template <> struct Lanes<float> { static constexpr int PPT = 4; typedef float4 vec; };

template <typename ST, int DEPTH>
struct VecRegs {
  static constexpr int PPT = Lanes<ST>::PPT;
  typename Lanes<ST>::vec c[6];        // x1.x x1.y x1.z x2.x x2.y x2.z
  double2 d1[PPT / 2], d2[PPT / 2];    // per-match depths (always f64)
  template <typename V>
  static __device__ __forceinline__ V stream_load(const V* ptr) {
#if SBA_NT_LOADS
    typedef float f4 __attribute__((ext_vector_type(4)));
    const f4 r = __builtin_nontemporal_load(reinterpret_cast<const f4*>(ptr));
    return *reinterpret_cast<const V*>(&r);
#else
    return *ptr;
#endif
  }
  __device__ __forceinline__ void load(const Planes& pl, size_t p) {
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      c[k] = stream_load(reinterpret_cast<const typename Lanes<ST>::vec*>(pl.x1[k]) + p);
      c[3 + k] = stream_load(reinterpret_cast<const typename Lanes<ST>::vec*>(pl.x2[k]) + p);
    }
    if (DEPTH == DEPTH_PER_MATCH) {
#pragma unroll
      for (int h = 0; h < PPT / 2; ++h) {
        d1[h] = stream_load(reinterpret_cast<const double2*>(pl.d1) + p * (PPT / 2) + h);
        d2[h] = stream_load(reinterpret_cast<const double2*>(pl.d2) + p * (PPT / 2) + h);
      }
    }
  }
  __device__ __forceinline__ double coord(int k, int h) const {
    if (PPT == 2) return h == 0 ? static_cast<double>(c[k].x) : static_cast<double>(c[k].y);
    const float4& q = reinterpret_cast<const float4&>(c[k]);
    return h == 0 ? q.x : (h == 1 ? q.y : (h == 2 ? q.z : q.w));
  }
  __device__ __forceinline__ double depth1(int h) const { return (h & 1) ? d1[h >> 1].y : d1[h >> 1].x; }
  __device__ __forceinline__ double depth2(int h) const { return (h & 1) ? d2[h >> 1].y : d2[h >> 1].x; }
};

// ---- Huber: w = rho'(s), rho(s) ---------------------------------------------------------------
// Outlier region needs 1/sqrt(s): v_rsq_f64 seed + two Newton steps (f64 accuracy to ~2 ulp) instead of
// the library sqrt + divide (~40 instructions).  Inlier lanes discard the (possibly inf/nan) seed.
__device__ __forceinline__ void huber(double s, double delta, double delta2, double& w, double& rho,
                                      double& is_out) {
  double y = __builtin_amdgcn_rsq(s);
  const double hs = 0.5 * s;
  y = y * __builtin_fma(-hs * y, y, 1.5);
  y = y * __builtin_fma(-hs * y, y, 1.5);
  const bool out = s > delta2;
  w = out ? delta * y : 1.0;
  rho = out ? __builtin_fma(2.0 * delta, s * y, -delta2) : s;
  is_out = out ? 1.0 : 0.0;
}

// ---- one correspondence ------------------------------------------------------------------------
template <int MODE, int DEPTH, int KIND, bool LOSS>
__device__ __forceinline__ void accumulate(const SweepParams* __restrict__ P, double x, double y,
                                           double z, double u, double v, double q, double d1, double d2,
                                           bool valid, double* __restrict__ acc) {
  double r0 = P->Rn[0] * x + P->Rn[1] * y + P->Rn[2] * z;
  double r1 = P->Rn[3] * x + P->Rn[4] * y + P->Rn[5] * z;
  double r2 = P->Rn[6] * x + P->Rn[7] * y + P->Rn[8] * z;
  double e0, e1, e2;
  if (DEPTH == DEPTH_PER_MATCH) {
    r0 *= d1; r1 *= d1; r2 *= d1;
    e0 = r0 + __builtin_fma(d2, u, P->t[0]);
    e1 = r1 + __builtin_fma(d2, v, P->t[1]);
    e2 = r2 + __builtin_fma(d2, q, P->t[2]);
  } else {
    e0 = r0 + __builtin_fma(P->d2, u, P->t[0]);
    e1 = r1 + __builtin_fma(P->d2, v, P->t[1]);
    e2 = r2 + __builtin_fma(P->d2, q, P->t[2]);
  }
  const double s = e0 * e0 + e1 * e1 + e2 * e2;
  double w = 1.0, rho = s, is_out = 0.0;
  if (LOSS) huber(s, P->delta, P->delta2, w, rho, is_out);
  if (!valid) { w = 0.0; rho = 0.0; is_out = 0.0; }

  if (MODE == MODE_TRAN) {
    acc[0] += w;
    acc[1] = __builtin_fma(w, e0, acc[1]);
    acc[2] = __builtin_fma(w, e1, acc[2]);
    acc[3] = __builtin_fma(w, e2, acc[3]);
    acc[4] = __builtin_fma(0.5, rho, acc[4]);
    acc[5] += is_out;
    return;
  }

  if (KIND == KIND_FACTORED) {
    const double wr0 = w * r0, wr1 = w * r1, wr2 = w * r2;
    // M = sum w v v^T (upper)
    acc[0] = __builtin_fma(wr0, r0, acc[0]);
    acc[1] = __builtin_fma(wr0, r1, acc[1]);
    acc[2] = __builtin_fma(wr0, r2, acc[2]);
    acc[3] = __builtin_fma(wr1, r1, acc[3]);
    acc[4] = __builtin_fma(wr1, r2, acc[4]);
    acc[5] = __builtin_fma(wr2, r2, acc[5]);
    // C = sum w v e^T
    acc[6] = __builtin_fma(wr0, e0, acc[6]);
    acc[7] = __builtin_fma(wr0, e1, acc[7]);
    acc[8] = __builtin_fma(wr0, e2, acc[8]);
    acc[9] = __builtin_fma(wr1, e0, acc[9]);
    acc[10] = __builtin_fma(wr1, e1, acc[10]);
    acc[11] = __builtin_fma(wr1, e2, acc[11]);
    acc[12] = __builtin_fma(wr2, e0, acc[12]);
    acc[13] = __builtin_fma(wr2, e1, acc[13]);
    acc[14] = __builtin_fma(wr2, e2, acc[14]);
    if (MODE == MODE_ROT) {
      acc[15] = __builtin_fma(0.5, rho, acc[15]);
      acc[16] += is_out;
    } else {
      acc[15] += w;
      acc[16] += wr0; acc[17] += wr1; acc[18] += wr2;
      acc[19] = __builtin_fma(w, e0, acc[19]);
      acc[20] = __builtin_fma(w, e1, acc[20]);
      acc[21] = __builtin_fma(w, e2, acc[21]);
      acc[22] = __builtin_fma(0.5, rho, acc[22]);
      acc[23] += is_out;
    }
    return;
  }

  // KIND_EXPLICIT: A[r][j] = (Gn_j x1)[r]
  double A[3][3];
#pragma unroll
  for (int j = 0; j < 3; ++j) {
#pragma unroll
    for (int r = 0; r < 3; ++r) {
      double a = P->Gn[9 * j + 3 * r + 0] * x + P->Gn[9 * j + 3 * r + 1] * y +
                 P->Gn[9 * j + 3 * r + 2] * z;
      if (DEPTH == DEPTH_PER_MATCH) a *= d1;
      A[r][j] = a;
    }
  }
  double wA[3][3];
#pragma unroll
  for (int r = 0; r < 3; ++r)
#pragma unroll
    for (int j = 0; j < 3; ++j) wA[r][j] = w * A[r][j];
  int k = 0;
#pragma unroll
  for (int a = 0; a < 3; ++a)
#pragma unroll
    for (int b = a; b < 3; ++b) {
      acc[k] = __builtin_fma(wA[0][a], A[0][b],
               __builtin_fma(wA[1][a], A[1][b], __builtin_fma(wA[2][a], A[2][b], acc[k])));
      ++k;
    }
  if (MODE == MODE_ROT) {
#pragma unroll
    for (int a = 0; a < 3; ++a)
      acc[6 + a] = __builtin_fma(wA[0][a], e0,
                   __builtin_fma(wA[1][a], e1, __builtin_fma(wA[2][a], e2, acc[6 + a])));
    acc[9] = __builtin_fma(0.5, rho, acc[9]);
    acc[10] += is_out;
  } else {
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
      for (int c = 0; c < 3; ++c) acc[6 + 3 * a + c] += wA[c][a];   // sum w A^T
    acc[15] += w;
#pragma unroll
    for (int a = 0; a < 3; ++a)
      acc[16 + a] = __builtin_fma(wA[0][a], e0,
                    __builtin_fma(wA[1][a], e1, __builtin_fma(wA[2][a], e2, acc[16 + a])));
    acc[19] = __builtin_fma(w, e0, acc[19]);
    acc[20] = __builtin_fma(w, e1, acc[20]);
    acc[21] = __builtin_fma(w, e2, acc[21]);
    acc[22] = __builtin_fma(0.5, rho, acc[22]);
    acc[23] += is_out;
  }
}

template <int MODE, int DEPTH, typename ST, int KIND, bool LOSS, bool CHECK>
__device__ __forceinline__ void consume(const VecRegs<ST, DEPTH>& r, const SweepParams* __restrict__ P,
                                        size_t p, size_t n, double* __restrict__ acc) {
  constexpr int PPT = Lanes<ST>::PPT;
#pragma unroll
  for (int h = 0; h < PPT; ++h)
    accumulate<MODE, DEPTH, KIND, LOSS>(P, r.coord(0, h), r.coord(1, h), r.coord(2, h), r.coord(3, h),
                                        r.coord(4, h), r.coord(5, h),
                                        DEPTH == DEPTH_PER_MATCH ? r.depth1(h) : 1.0,
                                        DEPTH == DEPTH_PER_MATCH ? r.depth2(h) : 0.0,
                                        CHECK ? (p * PPT + h < n) : true, acc);
}

}  // namespace
}  // namespace sba
