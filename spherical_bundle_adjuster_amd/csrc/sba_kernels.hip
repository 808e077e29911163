// HIP kernels (gfx950 / CDNA4, wave64) of the spherical bundle-adjustment hot path.
//
// sweep kernels = the fused replacement of one Ceres residual+Jacobian evaluation over all residual
// blocks (reference spherical_bundle_adjuster.cpp:843-868, :891-919, :947-976 evaluated through
// AutoDiffCostFunction + HuberLoss(1.0), :887/:943/:1000).  Per correspondence i ("one evaluation"):
//
//       v  = d1 Rn x1_i            (Rn = -R, or -d1 R with uniform depths)
//       e  = t + d2 x2_i + v                                   residual            (.cpp:897-916)
//       s  = e.e ;  w = rho'(s) ;  rho(s)                      block-wise Huber (Ceres corrector with
//                                                              rho'' <= 0: scale e and J by sqrt(w))
//   KIND_FACTORED (default):  d e/d rot = A = -[v]x J_l(rot) with J_l constant per sweep, so the
//       normal equations factor through moments of v and e; the kernel accumulates
//           M = sum w v v^T (6), C = sum w v e^T (9), m = sum w v (3), sum w e (3), sum w, cost
//       and the host applies J_l (sba_rotation.hpp: moments_to_normal_pack).
//   KIND_EXPLICIT:  A = d1 [Gn_0 x1 | Gn_1 x1 | Gn_2 x1] formed per match (what Jet<double,3> carries),
//           acc += w A^T A, w A^T, w, w A^T e, w e, rho/2
//       kept as the independent cross-check of the factored form.
//   wave:   DPP butterfly over the 64 lanes
//   block:  4 waves through LDS -> partials[block][24]
//   grid:   finalize_kernel folds partials in a fixed order -> pack[24]  (deterministic)
//
// Memory: the coordinate planes are read exactly once, 16 B per lane per load instruction (1 KiB
// contiguous per wave instruction), next vector prefetched into registers while the current one is
// consumed.  No reuse, no MFMA: HBM-bound (48 B per evaluation with f64 planes and uniform depths,
// 64 B with per-match depths).  The wave-uniform R|t state (SweepParams) is a by-value kernel argument:
// scalar loads, operands stay in SGPRs (SBA_PARAMS_IN_LDS=1 stages it in LDS instead).
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

template <int MODE, int DEPTH, typename ST, int KIND, bool LOSS>
__global__ __launch_bounds__(kBlock) void sweep_kernel(Planes pl, SweepParams prm, SweepOut out) {
  double* __restrict__ partials = out.partials;
  constexpr int NACC = AccMap<MODE, KIND>::N;
  constexpr int PPT = Lanes<ST>::PPT;
  // One LDS object: [0, 48) the staged R|t state (SBA_PARAMS_IN_LDS) / the "I am last" word, then the
  // cross-wave scratch (4 x 24), reused by the fused final fold (kBlock/16 x 24).
  __shared__ double lds[48 + (kBlock / 16) * 24];
  double* wave_out = lds + 48;
  const int tid = threadIdx.x;
#if SBA_PARAMS_IN_LDS
  const SweepParams* P = reinterpret_cast<const SweepParams*>(lds);
  if (tid < 43) lds[tid] = reinterpret_cast<const double*>(&prm)[tid];
  __syncthreads();
#else
  const SweepParams* P = &prm;   // kernarg segment: scalar loads, operands stay in SGPRs
#endif
  const size_t n = prm.n;
  const size_t stride = static_cast<size_t>(gridDim.x) * kBlock;

  double acc[NACC];
#pragma unroll
  for (int k = 0; k < NACC; ++k) acc[k] = 0.0;

  // Full vectors: every lane valid, no masking in the hot loop.  The next vector's loads are issued
  // before the current one is consumed (register double buffer), so each wave keeps 6-8 KiB in flight
  // while its VALU works.
  const size_t nfull = n / PPT;
  size_t p = static_cast<size_t>(blockIdx.x) * kBlock + tid;
  VecRegs<ST, DEPTH> cur, nxt;
  if (p < nfull) cur.load(pl, p);
  while (p < nfull) {
    const size_t pn = p + stride;
    if (pn < nfull) nxt.load(pl, pn);
    consume<MODE, DEPTH, ST, KIND, LOSS, false>(cur, P, p, n, acc);
    cur = nxt;
    p = pn;
  }
  // Ragged tail (n % PPT != 0): one lane of the grid handles the last, partly valid vector
  // (the planes are zero-padded to a whole vector at upload).
  if (nfull * PPT != n && blockIdx.x == gridDim.x - 1 && tid == kBlock - 1) {
    cur.load(pl, nfull);
    consume<MODE, DEPTH, ST, KIND, LOSS, true>(cur, P, nfull, n, acc);
  }

  // wave -> lane 63 -> LDS -> block partial (pack layout, unused slots zero)
  const int lane = tid & 63, wave = tid >> 6;
#pragma unroll
  for (int k = 0; k < NACC; ++k) {
    const double s = wave_sum_to_lane63(acc[k]);
    if (lane == 63) wave_out[wave * 24 + AccMap<MODE, KIND>::slot(k)] = s;
  }
  if (NACC < 24 && tid < 24) {   // slots this mode does not produce
    bool used = false;
#pragma unroll
    for (int k = 0; k < NACC; ++k) used |= (AccMap<MODE, KIND>::slot(k) == tid);
    if (!used) {
#pragma unroll
      for (int wv = 0; wv < kBlock / 64; ++wv) wave_out[wv * 24 + tid] = 0.0;
    }
  }
  __syncthreads();
  // Block partial: one 256-byte row (24 sums + 8 zeros), stored by lanes 0..31 of wave 0 in ONE wave instruction
  // so that both 128-byte lines of the row are written whole.
  if (out.ticket == nullptr) {          // two-kernel mode: finalize_kernel folds the rows after the kernel boundary
    if (tid < kRow) {
      double s = 0.0;
      if (tid < 24) {
        s = wave_out[tid];
#pragma unroll
        for (int wv = 1; wv < kBlock / 64; ++wv) s += wave_out[wv * 24 + tid];
      }
      partials[static_cast<size_t>(blockIdx.x) * kRow + tid] = s;
    }
    return;
  }

  // ---- fused final reduction: the block that arrives last folds all rows (fixed order => the result does not
  // depend on which block that is) and publishes the pack.  Hand-off (cdna_hip_programming.md Guideline 16,
  // write-through form): sc1 row stores -> the storing wave's vmcnt(0) -> relaxed agent-scope ticket;
  // the reducer acquires at agent scope once, then reads.  Tickets are hierarchical: blocks with equal
  // blockIdx % 8 (observed to share an XCD; a speed assumption only) count on their own word, the last of each
  // group counts on the top word -- 8 short queues instead of one long one.
  if (tid < 64) {                        // wave 0
    bool last = false;
    if (tid < kRow) {
      double s = 0.0;
      if (tid < 24) {
        s = wave_out[tid];
#pragma unroll
        for (int wv = 1; wv < kBlock / 64; ++wv) s += wave_out[wv * 24 + tid];
      }
      __hip_atomic_store(&partials[static_cast<size_t>(blockIdx.x) * kRow + tid], s, __ATOMIC_RELAXED,
                         __HIP_MEMORY_SCOPE_AGENT);                                  // global_store ... sc1
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (tid == 0) {
      const unsigned g = blockIdx.x & 7u, ngroups = gridDim.x < 8u ? gridDim.x : 8u;
      const unsigned in_group = (gridDim.x - g + 7u) / 8u;                           // blocks b with b % 8 == g
      if (__hip_atomic_fetch_add(out.ticket + 16 * (1 + g), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) ==
          in_group - 1)
        last = __hip_atomic_fetch_add(out.ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == ngroups - 1;
      if (last) {
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
      lds[47] = last ? 1.0 : 0.0;
    }
  }
  __syncthreads();
  if (lds[47] == 0.0) return;
  {
    // 16 slot pairs x NG groups of blocks; every thread issues up to 12 independent 16-byte loads before it adds
    // (the rows were just written by other XCDs: each round trip costs ~2 us, so memory-level parallelism, not
    // instruction count, sets the length of this tail).
    const int nblocks = static_cast<int>(gridDim.x);
    constexpr int NG = kBlock / 16;                       // 16 groups of blocks
    const int pair = tid & 15, grp = tid >> 4;
    double sx = 0.0, sy = 0.0;
    if (pair < 12) {
      const double2* src = reinterpret_cast<const double2*>(partials) + pair;   // row stride = 16 double2
      for (int b0 = grp; b0 < nblocks; b0 += 12 * NG) {
        double2 v[12];
#pragma unroll
        for (int k = 0; k < 12; ++k) {
          const int b = b0 + k * NG;
          v[k] = b < nblocks ? src[static_cast<size_t>(b) * (kRow / 2)] : make_double2(0.0, 0.0);
        }
#pragma unroll
        for (int k = 0; k < 12; ++k) { sx += v[k].x; sy += v[k].y; }
      }
    }
    double* fold_lds = wave_out;                         // the cross-wave scratch is free again: NG x 24 doubles
    if (pair < 12) { fold_lds[grp * 24 + 2 * pair] = sx; fold_lds[grp * 24 + 2 * pair + 1] = sy; }
    __syncthreads();
    if (tid < 64) {                                      // wave 0 finishes alone: no further block barrier
      if (tid < 24) {
        double tot = fold_lds[tid];
#pragma unroll
        for (int g = 1; g < NG; ++g) tot += fold_lds[g * 24 + tid];
        out.pack_dev[tid] = tot;
        if (out.pack_host) out.pack_host[tid] = tot;
      }
      if (tid < 9) out.ticket[16 * tid] = 0;             // counters ready for the next launch on this stream
      if (out.pack_host) {
        // publish to the host: this wave's pack stores, a system-scope release, then the sequence number
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (tid == 0)
          __hip_atomic_store(reinterpret_cast<unsigned long long*>(out.pack_host + 24), out.seq, __ATOMIC_RELAXED,
                             __HIP_MEMORY_SCOPE_SYSTEM);
      }
    }
  }
}

// ---- batched sweep: many independent two-view problems ("pairs") in ONE launch (BASELINE config C5) ------------
// Block group g = blockIdx.x / bpp works on pair g with that pair's own R|t (params[g], wave-uniform address ->
// scalar loads), blocks j = blockIdx.x % bpp of the group grid-stride over the pair's vectors.  Rows of block
// partials are folded per pair by batch_finalize_kernel.  A pair with n == 0 (e.g. already converged) costs its
// blocks only the row store.
template <int MODE, int DEPTH, typename ST, int KIND, bool LOSS>
__global__ __launch_bounds__(kBlock) void batch_sweep_kernel(Planes pl, const SweepParams* __restrict__ params,
                                                            const PairDesc* __restrict__ desc, int bpp,
                                                            double* __restrict__ partials) {
  constexpr int NACC = AccMap<MODE, KIND>::N;
  constexpr int PPT = Lanes<ST>::PPT;
  __shared__ double lds[(kBlock / 64) * 24];
  double* wave_out = lds;
  const int tid = threadIdx.x;
  const unsigned pair = blockIdx.x / static_cast<unsigned>(bpp), j = blockIdx.x % static_cast<unsigned>(bpp);
  const SweepParams* __restrict__ P = params + pair;
  const size_t n = P->n, first = desc[pair].first_vec;
  const size_t stride = static_cast<size_t>(bpp) * kBlock;

  double acc[NACC];
#pragma unroll
  for (int k = 0; k < NACC; ++k) acc[k] = 0.0;
  const size_t nfull = n / PPT;
  size_t p = static_cast<size_t>(j) * kBlock + tid;
  VecRegs<ST, DEPTH> cur, nxt;
  if (p < nfull) cur.load(pl, first + p);
  while (p < nfull) {
    const size_t pn = p + stride;
    if (pn < nfull) nxt.load(pl, first + pn);
    consume<MODE, DEPTH, ST, KIND, LOSS, false>(cur, P, p, n, acc);
    cur = nxt;
    p = pn;
  }
  if (nfull * PPT != n && j == static_cast<unsigned>(bpp) - 1 && tid == kBlock - 1) {
    cur.load(pl, first + nfull);
    consume<MODE, DEPTH, ST, KIND, LOSS, true>(cur, P, nfull, n, acc);
  }
  const int lane = tid & 63, wave = tid >> 6;
#pragma unroll
  for (int k = 0; k < NACC; ++k) {
    const double s = wave_sum_to_lane63(acc[k]);
    if (lane == 63) wave_out[wave * 24 + AccMap<MODE, KIND>::slot(k)] = s;
  }
  if (NACC < 24 && tid < 24) {
    bool used = false;
#pragma unroll
    for (int k = 0; k < NACC; ++k) used |= (AccMap<MODE, KIND>::slot(k) == tid);
    if (!used) {
#pragma unroll
      for (int wv = 0; wv < kBlock / 64; ++wv) wave_out[wv * 24 + tid] = 0.0;
    }
  }
  __syncthreads();
  if (tid < kRow) {
    double s = 0.0;
    if (tid < 24) {
      s = wave_out[tid];
#pragma unroll
      for (int wv = 1; wv < kBlock / 64; ++wv) s += wave_out[wv * 24 + tid];
    }
    partials[static_cast<size_t>(blockIdx.x) * kRow + tid] = s;
  }
}

// ONE block folds every pair's bpp rows (fixed order) into packs[pair][24], on the device and -- when packs_host is
// given -- in mapped pinned host memory, followed by a system-scope release and the sequence number in
// packs_host[24 * num_pairs]: the host polls that word instead of queueing a D2H copy and synchronising the stream.
__global__ __launch_bounds__(1024) void batch_finalize_kernel(const double* __restrict__ partials, int bpp,
                                                              int num_pairs, double* __restrict__ packs,
                                                              double* __restrict__ packs_host,
                                                              unsigned long long seq) {
  const int items = num_pairs * 24;
  for (int it = threadIdx.x; it < items; it += 1024) {
    const int pair = it / 24, slot = it - pair * 24;
    const double* rows = partials + static_cast<size_t>(pair) * bpp * kRow + slot;
    double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
    int b = 0;
    for (; b + 3 < bpp; b += 4) {
      s0 += rows[static_cast<size_t>(b) * kRow];
      s1 += rows[static_cast<size_t>(b + 1) * kRow];
      s2 += rows[static_cast<size_t>(b + 2) * kRow];
      s3 += rows[static_cast<size_t>(b + 3) * kRow];
    }
    for (; b < bpp; ++b) s0 += rows[static_cast<size_t>(b) * kRow];
    const double tot = (s0 + s1) + (s2 + s3);
    packs[it] = tot;
    if (packs_host) packs_host[it] = tot;
  }
  if (!packs_host) return;
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");          // every thread: its host stores before the barrier
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0)
    __hip_atomic_store(reinterpret_cast<unsigned long long*>(packs_host + items), seq, __ATOMIC_RELAXED,
                       __HIP_MEMORY_SCOPE_SYSTEM);
}

// ---- batched step with the per-pair host work moved to the device ----------------------------------------------
// One thread per pair: SweepParams (rotation, its derivatives, the Huber constants) from the pair's (rot, tran, depths)
// in `state` (mapped host memory: 80 B per pair instead of the 344-byte SweepParams crossing PCIe), and for the factored
// kernel the frame (B, J) that batch_convert_finalize_kernel applies to the pair's moments.
__global__ __launch_bounds__(64) void batch_prepare_kernel(const BatchState* __restrict__ state, int num_pairs,
                                                           int depth_mode, double huber_delta, int with_frames,
                                                           SweepParams* __restrict__ params,
                                                           double* __restrict__ frames) {
  const int g = blockIdx.x * 64 + threadIdx.x;
  if (g >= num_pairs) return;
  const BatchState st = state[g];
  SweepParams prm;
  fill_sweep_params(st.n, depth_mode, st.rot, st.tran, st.d1, st.d2, huber_delta, &prm);
  params[g] = prm;
  if (with_frames) {
    double B[9], J[9];
    factored_frame(st.rot, B, J);
#pragma unroll
    for (int i = 0; i < 9; ++i) { frames[18 * g + i] = B[i]; frames[18 * g + 9 + i] = J[i]; }
  }
}

// Fold (as batch_finalize_kernel), then one thread per pair maps the moment pack to the SBA_PACK_* layout with the
// pair's own frame, and everything is published to the host.  convert: 1 = rot free, 2 = rot + tran free.
__global__ __launch_bounds__(1024) void batch_convert_finalize_kernel(const double* __restrict__ partials, int bpp,
                                                                      int num_pairs, const double* __restrict__ frames,
                                                                      int convert, double* __restrict__ packs,
                                                                      double* __restrict__ packs_host,
                                                                      unsigned long long seq) {
  const int items = num_pairs * 24;
  for (int it = threadIdx.x; it < items; it += 1024) {
    const int pair = it / 24, slot = it - pair * 24;
    const double* rows = partials + static_cast<size_t>(pair) * bpp * kRow + slot;
    double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
    int b = 0;
    for (; b + 3 < bpp; b += 4) {
      s0 += rows[static_cast<size_t>(b) * kRow];
      s1 += rows[static_cast<size_t>(b + 1) * kRow];
      s2 += rows[static_cast<size_t>(b + 2) * kRow];
      s3 += rows[static_cast<size_t>(b + 3) * kRow];
    }
    for (; b < bpp; ++b) s0 += rows[static_cast<size_t>(b) * kRow];
    packs[it] = (s0 + s1) + (s2 + s3);
  }
  __syncthreads();                                         // the raw packs of this block are visible to all its threads
  for (int pair = threadIdx.x; pair < num_pairs; pair += 1024) {
    double raw[24], out[24];
#pragma unroll
    for (int k = 0; k < 24; ++k) raw[k] = packs[pair * 24 + k];
    moments_to_normal_pack(true, convert == 2, frames + 18 * pair, frames + 18 * pair + 9, raw, out);
#pragma unroll
    for (int k = 0; k < 24; ++k) {
      packs[pair * 24 + k] = out[k];
      if (packs_host) packs_host[pair * 24 + k] = out[k];
    }
  }
  if (!packs_host) return;
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0)
    __hip_atomic_store(reinterpret_cast<unsigned long long*>(packs_host + items), seq, __ATOMIC_RELAXED,
                       __HIP_MEMORY_SCOPE_SYSTEM);
}

// ---- direct peer exchange over xGMI: the all-reduce of the 24-double pack without a collective library ----------
// Every rank owns an "inbox" in fine-grained device memory, mapped into all peers through HIP IPC:
//     inbox[parity][source rank][32]   (24 doubles of payload, word 31 = sequence number; 256 B per slot)
// One wave per rank: (1) store the local pack into slot [parity][my rank] of EVERY rank's inbox (peer stores travel
// over xGMI), (2) system-scope release, (3) store the sequence number into each of those slots, (4) poll the own
// inbox until all nranks slots carry this sequence number, (5) system-scope acquire, (6) sum the slots in RANK ORDER
// -- every rank adds the same numbers in the same order, so all ranks obtain the bit-identical result -- and
// (7) publish the pack to the host like publish_kernel.  Two parities: a rank can run at most one exchange ahead of
// the slowest (it needs everybody's slot of round s before it can finish round s), so round s+1 never overwrites a
// slot somebody still reads.  Spins are bounded; on a timeout word 25 of the host pack is set so the host reports
// SBA_ERR_COMM instead of waiting forever.
// The exchange as executed by ONE wave (lanes 0..23 hold the local pack in `v`); returns the all-reduced value for the
// lane and whether every source rank arrived in time.
__device__ __forceinline__ double peer_exchange_wave(double v, const PeerInboxes& px, unsigned long long seq,
                                                     unsigned long long spin_limit, int lane, bool* all_ok) {
  const unsigned parity = static_cast<unsigned>(seq & 1ull);
  const size_t my_slot = (static_cast<size_t>(parity) * kMaxPeers + px.rank) * 32;
  // (1) payload to every inbox (own included)
  if (lane < 24)
    for (int r = 0; r < px.nranks; ++r)
      __hip_atomic_store(px.inbox[r] + my_slot + lane, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  // (2) + (3): all payload stores of this wave are complete and visible before any sequence number is
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if (lane < px.nranks)
    __hip_atomic_store(reinterpret_cast<unsigned long long*>(px.inbox[lane] + my_slot + 31), seq, __ATOMIC_RELAXED,
                       __HIP_MEMORY_SCOPE_SYSTEM);
  // (4) lane r waits for source rank r
  bool ok = true;
  if (lane < px.nranks) {
    const unsigned long long* flag = reinterpret_cast<const unsigned long long*>(
        px.inbox[px.rank] + (static_cast<size_t>(parity) * kMaxPeers + lane) * 32 + 31);
    unsigned long long spins = 0;
    while (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != seq) {
      if (++spins > spin_limit) { ok = false; break; }
      __builtin_amdgcn_s_sleep(2);
    }
  }
  *all_ok = __ballot(!ok) == 0ull;
  // (5)
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");
  // (6) rank-ordered sum
  double tot = 0.0;
  if (lane < 24)
    for (int r = 0; r < px.nranks; ++r)
      tot += __hip_atomic_load(px.inbox[px.rank] + (static_cast<size_t>(parity) * kMaxPeers + r) * 32 + lane,
                               __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  return tot;
}

// (7) of the exchange / the single-GPU hand-over: lanes 0..23 of one wave store the pack into mapped pinned host
// memory, word 25 = error flag, system-scope release, then the sequence number the host polls.
__device__ __forceinline__ void publish_wave(double v, double* __restrict__ pack_host, unsigned long long host_seq,
                                             bool ok, int lane) {
  if (lane < 24) pack_host[lane] = v;
  if (lane == 0) reinterpret_cast<unsigned long long*>(pack_host)[25] = ok ? 0ull : 1ull;
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if (lane == 0)
    __hip_atomic_store(reinterpret_cast<unsigned long long*>(pack_host + 24), host_seq, __ATOMIC_RELAXED,
                       __HIP_MEMORY_SCOPE_SYSTEM);
}

// After a library collective (the all-reduced pack sits in device memory): one wave hands it to the host.
__global__ __launch_bounds__(64) void publish_kernel(const double* __restrict__ pack_dev,
                                                     double* __restrict__ pack_host, unsigned long long seq) {
  const int lane = threadIdx.x;
  publish_wave(lane < 24 ? pack_dev[lane] : 0.0, pack_host, seq, true, lane);
}

__global__ __launch_bounds__(64) void peer_exchange_kernel(const double* __restrict__ pack_local, PeerInboxes px,
                                                           unsigned long long seq, double* __restrict__ pack_out,
                                                           double* __restrict__ pack_host,
                                                           unsigned long long host_seq,
                                                           unsigned long long spin_limit) {
  const int lane = threadIdx.x;
  bool all_ok = true;
  const double tot = peer_exchange_wave(lane < 24 ? pack_local[lane] : 0.0, px, seq, spin_limit, lane, &all_ok);
  if (lane < 24) pack_out[lane] = tot;
  if (pack_host) publish_wave(tot, pack_host, host_seq, all_ok, lane);
}

// Fold partials[nblocks][kRow] in a fixed order.  1024 threads = 32 slots x 32 groups: group g sums blocks g, g+32, ...
// for its slot (4 independent chains so the loads pipeline), then the 32 groups are summed serially per slot.
// Independent of timing, so results are run-to-run identical.  Wave 0 then hands the pack on: with px.nranks > 0 it
// first runs the peer exchange (all-reduce across ranks), and with pack_host != nullptr it publishes the result to the
// host -- one launch after the sweep covers reduction, exchange and hand-over.
__global__ __launch_bounds__(1024) void finalize_kernel(const double* __restrict__ partials,
                                                        int nblocks, double* __restrict__ pack_out,
                                                        double* __restrict__ pack_host, unsigned long long seq,
                                                        PeerInboxes px, unsigned long long xseq,
                                                        unsigned long long spin_limit) {
  __shared__ double part[32][33];
  const int slot = threadIdx.x & 31, grp = threadIdx.x >> 5;
  double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
  if (slot < 24) {
    int b = grp;
    for (; b + 96 < nblocks; b += 128) {
      s0 += partials[static_cast<size_t>(b) * kRow + slot];
      s1 += partials[static_cast<size_t>(b + 32) * kRow + slot];
      s2 += partials[static_cast<size_t>(b + 64) * kRow + slot];
      s3 += partials[static_cast<size_t>(b + 96) * kRow + slot];
    }
    for (; b < nblocks; b += 32) s0 += partials[static_cast<size_t>(b) * kRow + slot];
  }
  part[grp][slot] = (s0 + s1) + (s2 + s3);
  __syncthreads();
  if (threadIdx.x >= 64) return;          // wave 0 finishes alone
  const int lane = threadIdx.x;
  double tot = 0.0;
  if (lane < 24) {
    tot = part[0][lane];
#pragma unroll
    for (int g = 1; g < 32; ++g) tot += part[g][lane];
  }
  bool ok = true;
  if (px.nranks > 0) tot = peer_exchange_wave(tot, px, xseq, spin_limit, lane, &ok);
  if (lane < 24) pack_out[lane] = tot;
  if (pack_host) publish_wave(tot, pack_host, seq, ok, lane);
}

// ---- layout conversion at upload time (once per problem, not per LM iteration) ---------------
template <typename ST>
__global__ void aos_to_planes_kernel(const double* __restrict__ aos, size_t n, size_t first,
                                     ST* __restrict__ px, ST* __restrict__ py, ST* __restrict__ pz) {
  const size_t i = static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i >= n) return;
  px[first + i] = static_cast<ST>(aos[3 * i + 0]);
  py[first + i] = static_cast<ST>(aos[3 * i + 1]);
  pz[first + i] = static_cast<ST>(aos[3 * i + 2]);
}
__global__ void d12_to_planes_kernel(const double* __restrict__ d12, size_t n, size_t first,
                                     double* __restrict__ d1, double* __restrict__ d2) {
  const size_t i = static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const double2 d = reinterpret_cast<const double2*>(d12)[i];
  d1[first + i] = d.x;
  d2[first + i] = d.y;
}
__global__ void planes_to_d12_kernel(const double* __restrict__ d1, const double* __restrict__ d2,
                                     size_t n, double* __restrict__ d12) {
  const size_t i = static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i >= n) return;
  reinterpret_cast<double2*>(d12)[i] = make_double2(d1[i], d2[i]);
}

// ---- pixel -> unit sphere (reference spherical_bundle_adjuster.cpp:271-298) ---------------------
//   lon = 2 pi (pt.x / W), colat = pi (pt.y / H);  v = (sin colat cos lon, sin colat sin lon, cos colat)
// pt.x / pt.y are the first two floats of each `stride_bytes`-byte key-point record.
__global__ void keypoints_to_sphere_kernel(const uint8_t* __restrict__ kp, size_t n, size_t stride_bytes,
                                           double im_w, double im_h, double* __restrict__ out_xyz) {
  const size_t i = static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float* rec = reinterpret_cast<const float*>(kp + i * stride_bytes);
  const double px = static_cast<double>(rec[0]), py = static_cast<double>(rec[1]);
  const double kPi = 3.14159265358979323846;
  const double lon = 2 * kPi * (px / im_w);
  const double colat = kPi * (py / im_h);
  const double sc = sin(colat), cc = cos(colat);
  out_xyz[3 * i + 0] = sc * cos(lon);
  out_xyz[3 * i + 1] = sc * sin(lon);
  out_xyz[3 * i + 2] = cc;
}

// Same map, fused with the upload: key-point records of BOTH images -> the six coordinate planes of a problem
// (no host-side cv::Point3d arrays in between).
template <typename ST>
__global__ void keypoints_to_planes_kernel(const uint8_t* __restrict__ kp_left, const uint8_t* __restrict__ kp_right,
                                           size_t n, size_t stride_bytes, double im_w, double im_h,
                                           ST* __restrict__ x1x, ST* __restrict__ x1y, ST* __restrict__ x1z,
                                           ST* __restrict__ x2x, ST* __restrict__ x2y, ST* __restrict__ x2z) {
  const size_t i = static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const double kPi = 3.14159265358979323846;
  const float* l = reinterpret_cast<const float*>(kp_left + i * stride_bytes);
  const float* r = reinterpret_cast<const float*>(kp_right + i * stride_bytes);
  const double lon1 = 2 * kPi * (static_cast<double>(l[0]) / im_w), col1 = kPi * (static_cast<double>(l[1]) / im_h);
  const double lon2 = 2 * kPi * (static_cast<double>(r[0]) / im_w), col2 = kPi * (static_cast<double>(r[1]) / im_h);
  const double s1 = sin(col1), s2 = sin(col2);
  x1x[i] = static_cast<ST>(s1 * cos(lon1)); x1y[i] = static_cast<ST>(s1 * sin(lon1)); x1z[i] = static_cast<ST>(cos(col1));
  x2x[i] = static_cast<ST>(s2 * cos(lon2)); x2y[i] = static_cast<ST>(s2 * sin(lon2)); x2z[i] = static_cast<ST>(cos(col2));
}

// ---- ERP -> cubemap strip (reference equi2cube.cpp:12-302) ----------------------------------------
// Output strip is S x 6S, faces left,front,right,back,top,bottom (equi2cube.cpp:292-298).  Per
// output pixel (i = row, j = column inside the face) the face-specific direction
// (equi2cube.cpp:28-30, 73-75, 118-120, 163-165, 208-210, 253-255) is normalised and mapped to a
// source pixel with truncation (equi2cube.cpp:40-50).  Each lane produces PIX consecutive output
// pixels so that stores are whole dwords; the gather side is byte loads (poor locality at the
// poles is inherent to the mapping).
__device__ __forceinline__ int erp_source_index(int face, int i, int j, int S, int im_h, int im_w) {
  const double s = static_cast<double>(S);
  const double a = (s - 2.0 * j) / s;   // (cube_size - 2 j) / cube_size
  const double b = (s - 2.0 * i) / s;   // (cube_size - 2 i) / cube_size
  const double an = (2.0 * j - s) / s;  // (2 j - cube_size) / cube_size
  const double bn = (2.0 * i - s) / s;
  double x, y, z;
  switch (face) {
    case 0: x = a;    y = 1.0;  z = b;    break;  // left   (.cpp:118-120)
    case 1: x = -1.0; y = a;    z = b;    break;  // front  (.cpp:73-75)
    case 2: x = an;   y = -1.0; z = b;    break;  // right  (.cpp:163-165)
    case 3: x = 1.0;  y = an;   z = b;    break;  // back   (.cpp:28-30)
    case 4: x = b;    y = a;    z = 1.0;  break;  // top    (.cpp:208-210)
    default: x = bn;  y = a;    z = -1.0; break;  // bottom (.cpp:253-255)
  }
  const double kPi = 3.14159265358979323846;
  const double nrm = sqrt(x * x + y * y + z * z);
  const double ux = x / nrm, uy = y / nrm, uz = z / nrm;
  const double theta = acos(uz);
  double phi = atan2(uy, ux);
  if (phi < 0) phi += kPi * 2;
  int row = static_cast<int>(im_h * theta / kPi);
  int col = static_cast<int>(im_w * phi / (2 * kPi));
  // The reference does not clamp (equi2cube.cpp:47-50); only the exact pole could leave the image.
  row = min(max(row, 0), im_h - 1);
  col = min(max(col, 0), im_w - 1);
  return row * im_w + col;
}

template <int PIX>
__global__ __launch_bounds__(256) void equi2cube_kernel(const uint8_t* __restrict__ erp, int im_h,
                                                        int im_w, int S, uint8_t* __restrict__ out,
                                                        size_t erp_stride, size_t out_stride, int batch,
                                                        int frames_per_block) {
  const int groups_per_row = (6 * S) / PIX;
  const size_t g = static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (g >= static_cast<size_t>(groups_per_row) * S) return;
  const int i = static_cast<int>(g / groups_per_row);
  const int c0 = static_cast<int>(g % groups_per_row) * PIX;   // strip column of the first pixel
  // The mapping depends only on (S, H, W): the f64 sqrt/acos/atan2 work is done once per output pixel and
  // reused for every frame of this block's slice of the batch; per frame only the gather and the store remain.
  size_t si[PIX];
#pragma unroll
  for (int k = 0; k < PIX; ++k) {
    const int c = c0 + k;
    const int face = c / S, j = c - face * S;
    si[k] = static_cast<size_t>(erp_source_index(face, i, j, S, im_h, im_w)) * 3;
  }
  const size_t o = (static_cast<size_t>(i) * 6 * S + c0) * 3;
  const int f0 = blockIdx.y * frames_per_block;
  const int f1 = min(batch, f0 + frames_per_block);
  for (int f = f0; f < f1; ++f) {
    const uint8_t* src = erp + static_cast<size_t>(f) * erp_stride;
    uint8_t* dst = out + static_cast<size_t>(f) * out_stride;
    uint8_t px[3 * PIX];
#pragma unroll
    for (int k = 0; k < PIX; ++k) {
      px[3 * k + 0] = src[si[k] + 0];
      px[3 * k + 1] = src[si[k] + 1];
      px[3 * k + 2] = src[si[k] + 2];
    }
    if (PIX == 4) {
      uint32_t* o32 = reinterpret_cast<uint32_t*>(dst + o);   // 12-byte group, 4-byte aligned
#pragma unroll
      for (int w = 0; w < 3; ++w)
        o32[w] = static_cast<uint32_t>(px[4 * w]) | (static_cast<uint32_t>(px[4 * w + 1]) << 8) |
                 (static_cast<uint32_t>(px[4 * w + 2]) << 16) | (static_cast<uint32_t>(px[4 * w + 3]) << 24);
    } else {
#pragma unroll
      for (int b = 0; b < 3 * PIX; ++b) dst[o + b] = px[b];
    }
  }
}

// ---- kernel table ------------------------------------------------------------------------------
typedef void (*SweepFn)(Planes, SweepParams, SweepOut);

template <int MODE, int DEPTH, typename ST, int KIND>
SweepFn pick_loss(bool loss) {
  return loss ? sweep_kernel<MODE, DEPTH, ST, KIND, true> : sweep_kernel<MODE, DEPTH, ST, KIND, false>;
}
template <int MODE, int DEPTH, typename ST>
SweepFn pick_kind(int kind, bool loss) {
  return kind == KIND_EXPLICIT ? pick_loss<MODE, DEPTH, ST, KIND_EXPLICIT>(loss)
                               : pick_loss<MODE, DEPTH, ST, KIND_FACTORED>(loss);
}
template <int MODE, int DEPTH>
SweepFn pick_store(int store, int kind, bool loss) {
  return store == 0 ? pick_kind<MODE, DEPTH, double>(kind, loss) : pick_kind<MODE, DEPTH, float>(kind, loss);
}
SweepFn pick(int mode, int depth, int store, int kind, bool loss) {
  switch (mode * 2 + depth) {
    case 0: return pick_store<MODE_ROT, DEPTH_UNIFORM>(store, kind, loss);
    case 1: return pick_store<MODE_ROT, DEPTH_PER_MATCH>(store, kind, loss);
    case 2: return pick_store<MODE_TRAN, DEPTH_UNIFORM>(store, KIND_FACTORED, loss);
    case 3: return pick_store<MODE_TRAN, DEPTH_PER_MATCH>(store, KIND_FACTORED, loss);
    case 4: return pick_store<MODE_RT, DEPTH_UNIFORM>(store, kind, loss);
    case 5: return pick_store<MODE_RT, DEPTH_PER_MATCH>(store, kind, loss);
  }
  return nullptr;
}

// ---- batch kernel table (same template axes as the single-problem sweep) ---------------------------------------
typedef void (*BatchFn)(Planes, const SweepParams*, const PairDesc*, int, double*);
template <int MODE, int DEPTH, typename ST, int KIND>
BatchFn bpick_loss(bool loss) {
  return loss ? batch_sweep_kernel<MODE, DEPTH, ST, KIND, true> : batch_sweep_kernel<MODE, DEPTH, ST, KIND, false>;
}
template <int MODE, int DEPTH, typename ST>
BatchFn bpick_kind(int kind, bool loss) {
  return kind == KIND_EXPLICIT ? bpick_loss<MODE, DEPTH, ST, KIND_EXPLICIT>(loss)
                               : bpick_loss<MODE, DEPTH, ST, KIND_FACTORED>(loss);
}
template <int MODE, int DEPTH>
BatchFn bpick_store(int store, int kind, bool loss) {
  return store == 0 ? bpick_kind<MODE, DEPTH, double>(kind, loss) : bpick_kind<MODE, DEPTH, float>(kind, loss);
}
BatchFn bpick(int mode, int depth, int store, int kind, bool loss) {
  switch (mode * 2 + depth) {
    case 0: return bpick_store<MODE_ROT, DEPTH_UNIFORM>(store, kind, loss);
    case 1: return bpick_store<MODE_ROT, DEPTH_PER_MATCH>(store, kind, loss);
    case 2: return bpick_store<MODE_TRAN, DEPTH_UNIFORM>(store, KIND_FACTORED, loss);
    case 3: return bpick_store<MODE_TRAN, DEPTH_PER_MATCH>(store, KIND_FACTORED, loss);
    case 4: return bpick_store<MODE_RT, DEPTH_UNIFORM>(store, kind, loss);
    case 5: return bpick_store<MODE_RT, DEPTH_PER_MATCH>(store, kind, loss);
  }
  return nullptr;
}

}  // namespace

int points_per_lane(int store) { return store == 0 ? 2 : 4; }

hipError_t batch_blocks_per_cu(int mode, int depth, int store, int kind, bool loss, int* blocks) {
  BatchFn fn = bpick(mode, depth, store, kind, loss);
  if (!fn) return hipErrorInvalidValue;
  return hipOccupancyMaxActiveBlocksPerMultiprocessor(blocks, reinterpret_cast<const void*>(fn), kBlock, 0);
}

hipError_t launch_batch_sweep(int mode, int depth, int store, int kind, bool loss, const Planes& pl,
                              const SweepParams* params, const PairDesc* desc, int num_pairs, int bpp,
                              double* partials, double* packs, double* packs_host, unsigned long long seq,
                              hipStream_t stream) {
  if (num_pairs <= 0) return hipSuccess;
  BatchFn fn = bpick(mode, depth, store, kind, loss);
  if (!fn) return hipErrorInvalidValue;
  hipLaunchKernelGGL(fn, dim3(static_cast<unsigned>(num_pairs) * bpp), dim3(kBlock), 0, stream, pl, params, desc,
                     bpp, partials);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(batch_finalize_kernel, dim3(1), dim3(1024), 0, stream, partials, bpp, num_pairs, packs,
                     packs_host, seq);
  return hipGetLastError();
}

hipError_t launch_batch_step(int mode, int depth, int store, int kind, double huber_delta, const Planes& pl,
                             const BatchState* state, SweepParams* params, double* frames, const PairDesc* desc,
                             int num_pairs, int bpp, double* partials, double* packs, double* packs_host,
                             unsigned long long seq, hipStream_t stream) {
  if (num_pairs <= 0) return hipSuccess;
  BatchFn fn = bpick(mode, depth, store, kind, huber_delta > 0.0);
  if (!fn) return hipErrorInvalidValue;
  const int convert = (kind == KIND_FACTORED && mode != MODE_TRAN) ? (mode == MODE_RT ? 2 : 1) : 0;
  hipLaunchKernelGGL(batch_prepare_kernel, dim3((num_pairs + 63) / 64), dim3(64), 0, stream, state, num_pairs, depth,
                     huber_delta, convert != 0 ? 1 : 0, params, frames);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(fn, dim3(static_cast<unsigned>(num_pairs) * bpp), dim3(kBlock), 0, stream, pl, params, desc, bpp,
                     partials);
  e = hipGetLastError();
  if (e != hipSuccess) return e;
  if (convert != 0)
    hipLaunchKernelGGL(batch_convert_finalize_kernel, dim3(1), dim3(1024), 0, stream, partials, bpp, num_pairs, frames,
                       convert, packs, packs_host, seq);
  else
    hipLaunchKernelGGL(batch_finalize_kernel, dim3(1), dim3(1024), 0, stream, partials, bpp, num_pairs, packs,
                       packs_host, seq);
  return hipGetLastError();
}

// Resident blocks per CU of the selected sweep kernel (the grid is sized to exactly one resident
// wave of blocks; the grid-stride loop spreads the vectors evenly over them).
hipError_t sweep_blocks_per_cu(int mode, int depth, int store, int kind, bool loss, int* blocks) {
  SweepFn fn = pick(mode, depth, store, kind, loss);
  if (!fn) return hipErrorInvalidValue;
  return hipOccupancyMaxActiveBlocksPerMultiprocessor(blocks, reinterpret_cast<const void*>(fn), kBlock, 0);
}

hipError_t launch_sweep(int mode, int depth, int store, int kind, const Planes& pl, const SweepParams& prm,
                        const SweepOut& out, int grid, hipStream_t stream) {
  if (grid <= 0) return hipSuccess;
  SweepFn fn = pick(mode, depth, store, kind, prm.delta > 0.0);
  if (!fn) return hipErrorInvalidValue;
  hipLaunchKernelGGL(fn, dim3(grid), dim3(kBlock), 0, stream, pl, prm, out);
  return hipGetLastError();
}

hipError_t launch_publish(const double* pack_dev, double* pack_host_dev, unsigned long long seq,
                          hipStream_t stream) {
  hipLaunchKernelGGL(publish_kernel, dim3(1), dim3(64), 0, stream, pack_dev, pack_host_dev, seq);
  return hipGetLastError();
}

hipError_t launch_peer_exchange(const double* pack_local, const PeerInboxes& px, unsigned long long xseq,
                                double* pack_out, double* pack_host_dev, unsigned long long host_seq,
                                unsigned long long spin_limit, hipStream_t stream) {
  hipLaunchKernelGGL(peer_exchange_kernel, dim3(1), dim3(64), 0, stream, pack_local, px, xseq, pack_out, pack_host_dev,
                     host_seq, spin_limit);
  return hipGetLastError();
}

hipError_t launch_finalize(const double* partials, int nblocks, double* pack_out, double* pack_host_dev,
                           unsigned long long seq, const PeerInboxes* px, unsigned long long xseq,
                           unsigned long long spin_limit, hipStream_t stream) {
  PeerInboxes none;
  for (auto& q : none.inbox) q = nullptr;
  none.nranks = 0;
  none.rank = 0;
  hipLaunchKernelGGL(finalize_kernel, dim3(1), dim3(1024), 0, stream, partials, nblocks, pack_out, pack_host_dev, seq,
                     px ? *px : none, xseq, spin_limit);
  return hipGetLastError();
}

hipError_t launch_aos_to_planes(const double* aos, size_t n, size_t first, void* px, void* py,
                                void* pz, int store, hipStream_t stream) {
  if (n == 0) return hipSuccess;
  const unsigned grid = static_cast<unsigned>((n + 255) / 256);
  if (store == 0)
    hipLaunchKernelGGL((aos_to_planes_kernel<double>), dim3(grid), dim3(256), 0, stream, aos, n, first,
                       static_cast<double*>(px), static_cast<double*>(py), static_cast<double*>(pz));
  else
    hipLaunchKernelGGL((aos_to_planes_kernel<float>), dim3(grid), dim3(256), 0, stream, aos, n, first,
                       static_cast<float*>(px), static_cast<float*>(py), static_cast<float*>(pz));
  return hipGetLastError();
}

hipError_t launch_d12_to_planes(const double* d12, size_t n, size_t first, double* d1, double* d2,
                                hipStream_t stream) {
  if (n == 0) return hipSuccess;
  const unsigned grid = static_cast<unsigned>((n + 255) / 256);
  hipLaunchKernelGGL(d12_to_planes_kernel, dim3(grid), dim3(256), 0, stream, d12, n, first, d1, d2);
  return hipGetLastError();
}

hipError_t launch_planes_to_d12(const double* d1, const double* d2, size_t n, double* d12,
                                hipStream_t stream) {
  if (n == 0) return hipSuccess;
  const unsigned grid = static_cast<unsigned>((n + 255) / 256);
  hipLaunchKernelGGL(planes_to_d12_kernel, dim3(grid), dim3(256), 0, stream, d1, d2, n, d12);
  return hipGetLastError();
}

hipError_t launch_keypoints_to_sphere(const uint8_t* kp, size_t n, size_t stride_bytes, double im_w,
                                      double im_h, double* out_xyz, hipStream_t stream) {
  if (n == 0) return hipSuccess;
  const unsigned grid = static_cast<unsigned>((n + 255) / 256);
  hipLaunchKernelGGL(keypoints_to_sphere_kernel, dim3(grid), dim3(256), 0, stream, kp, n, stride_bytes,
                     im_w, im_h, out_xyz);
  return hipGetLastError();
}

hipError_t launch_keypoints_to_planes(const uint8_t* kp_left, const uint8_t* kp_right, size_t n, size_t stride_bytes,
                                      double im_w, double im_h, void* const planes[6], int store, hipStream_t stream) {
  if (n == 0) return hipSuccess;
  const unsigned grid = static_cast<unsigned>((n + 255) / 256);
  if (store == 0)
    hipLaunchKernelGGL((keypoints_to_planes_kernel<double>), dim3(grid), dim3(256), 0, stream, kp_left, kp_right, n,
                       stride_bytes, im_w, im_h, static_cast<double*>(planes[0]), static_cast<double*>(planes[1]),
                       static_cast<double*>(planes[2]), static_cast<double*>(planes[3]), static_cast<double*>(planes[4]),
                       static_cast<double*>(planes[5]));
  else
    hipLaunchKernelGGL((keypoints_to_planes_kernel<float>), dim3(grid), dim3(256), 0, stream, kp_left, kp_right, n,
                       stride_bytes, im_w, im_h, static_cast<float*>(planes[0]), static_cast<float*>(planes[1]),
                       static_cast<float*>(planes[2]), static_cast<float*>(planes[3]), static_cast<float*>(planes[4]),
                       static_cast<float*>(planes[5]));
  return hipGetLastError();
}

hipError_t launch_equi2cube(const uint8_t* erp, int im_h, int im_w, int cube, int batch, uint8_t* out,
                            hipStream_t stream) {
  if (cube <= 0 || batch <= 0) return hipSuccess;
  const size_t erp_stride = static_cast<size_t>(im_h) * im_w * 3;
  const size_t out_stride = static_cast<size_t>(cube) * 6 * cube * 3;
  const bool wide = (6 * cube) % 4 == 0 && cube % 4 == 0;
  const size_t groups = wide ? static_cast<size_t>(6 * cube / 4) * cube : static_cast<size_t>(6 * cube) * cube;
  const unsigned gx = static_cast<unsigned>((groups + 255) / 256);
  // frames per block: amortise the index computation over the batch, but keep >= ~2048 blocks in flight
  int fpb = 1;
  while (fpb < 16 && fpb * 2 <= batch && static_cast<size_t>(gx) * ((batch + 2 * fpb - 1) / (2 * fpb)) >= 2048) fpb *= 2;
  const unsigned gy = static_cast<unsigned>((batch + fpb - 1) / fpb);
  if (wide)
    hipLaunchKernelGGL((equi2cube_kernel<4>), dim3(gx, gy), dim3(256), 0, stream, erp, im_h, im_w, cube, out,
                       erp_stride, out_stride, batch, fpb);
  else
    hipLaunchKernelGGL((equi2cube_kernel<1>), dim3(gx, gy), dim3(256), 0, stream, erp, im_h, im_w, cube, out,
                       erp_stride, out_stride, batch, fpb);
  return hipGetLastError();
}

}  // namespace sba
