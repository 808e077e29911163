// Device pass of the 8-point initial guess (reference spherical_bundle_adjuster.cpp:53-68: one row
// kron(left_i, right_i) of the N x 9 matrix A per match).  Instead of materialising A for 80 random subsets and
// running 80 SVDs of (N/4) x 9 matrices, one sweep accumulates A^T A separately for 64 interleaved groups of matches,
// group(i) = (i / 2) % 64: a lane always handles 2 consecutive matches -- ONE 16-byte vector of each f64 plane, so every
// wave load instruction covers 1 KiB contiguous like the sweep kernel's -- and keeps its own accumulators, so lane id ==
// group id and no cross-lane reduction is needed at all.  (Round 1 grouped by 4 matches: two 16-byte loads per lane at
// a 32-byte lane stride, i.e. every load instruction touched twice the cache lines it used; 5.0-5.4 TB/s.)
//
// Kronecker structure: entry ((p,q),(r,s)) of A^T A is sum l_p r_q l_r r_s = sum P_pr Q_qs with P = l l^T, Q = r r^T
// symmetric -- only 6 x 6 = 36 DISTINCT sums (the 45 entries of the upper triangle repeat 9 of them).  36 accumulators
// and 12 + 36 multiply-adds per match instead of 45 and 9 + 45; the fold kernel writes the 45-entry layout.
//
// 256-thread blocks: the four waves of a block stream different quads of matches for the SAME 64 groups and combine
// through LDS at the end (fixed order), so four times the loads are in flight per row of partials written.  The rows of
// all blocks are folded per (group, entry) by epipolar_fold_kernel in a fixed order.  Reads the same planes as the
// sweep kernel (48 B per match), once per problem -- not on the per-iteration path.
#include "sba_device.hpp"
#include "sba_epipolar.hpp"
#include "sba_pair_map.hpp"

namespace sba {
namespace {

constexpr int kMom = 45;     // upper triangle of the 9 x 9 matrix (what the host consumes)
constexpr int kSums = 36;    // distinct sums

__device__ __forceinline__ void add_match(double x, double y, double z, double u, double v, double w,
                                          double* __restrict__ acc) {
  // P = left left^T, Q = right right^T (upper triangles 00 01 02 11 12 22); row of A: left (x) right, .cpp:59-67
  const double P[6] = {x * x, x * y, x * z, y * y, y * z, z * z};
  const double Q[6] = {u * u, u * v, u * w, v * v, v * w, w * w};
#pragma unroll
  for (int a = 0; a < 6; ++a)
#pragma unroll
    for (int b = 0; b < 6; ++b) acc[a * 6 + b] = __builtin_fma(P[a], Q[b], acc[a * 6 + b]);
}

// which of the 36 sums is entry k of the 45 (upper triangle, row-major i <= j over i = 3 p + q, j = 3 r + s)
struct EntryMap { int src[kMom]; };
constexpr int sym6(int a, int b) { return a <= b ? (a == 0 ? b : (a == 1 ? 2 + b : 5)) : sym6(b, a); }
constexpr EntryMap make_entry_map() {
  EntryMap m{};
  int k = 0;
  for (int i = 0; i < 9; ++i)
    for (int j = i; j < 9; ++j) m.src[k++] = sym6(i / 3, j / 3) * 6 + sym6(i % 3, j % 3);
  return m;
}
__constant__ EntryMap kEntryMap = make_entry_map();

// two consecutive matches of one plane
template <typename ST>
__device__ __forceinline__ void load2(const void* plane, size_t vec, double out[2]);
template <>
__device__ __forceinline__ void load2<double>(const void* plane, size_t vec, double out[2]) {
  typedef float f4 __attribute__((ext_vector_type(4)));
  const f4 r = __builtin_nontemporal_load(reinterpret_cast<const f4*>(plane) + vec);
  const double2 a = *reinterpret_cast<const double2*>(&r);
  out[0] = a.x; out[1] = a.y;
}
template <>
__device__ __forceinline__ void load2<float>(const void* plane, size_t vec, double out[2]) {
  const float2 a = reinterpret_cast<const float2*>(plane)[vec];
  out[0] = a.x; out[1] = a.y;
}

// partials[block][sum][lane]  (sum-major so that the 64 lanes store 512 contiguous bytes per sum)
template <typename ST>
__global__ __launch_bounds__(256) void epipolar_moments_kernel(Planes pl, unsigned long long n,
                                                               double* __restrict__ partials) {
  __shared__ double red[2][kSums][64];
  double acc[kSums];
#pragma unroll
  for (int k = 0; k < kSums; ++k) acc[k] = 0.0;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const size_t nvec = (n + 1) / 2;
  const size_t stride = static_cast<size_t>(gridDim.x) * 256;
  // vector index = 64 * (something) + lane, so group(vector) = vector % 64 = lane for every wave of every block.
  // Two vectors ahead are in flight before the current one is consumed (register double buffer, like the sweep kernel).
  size_t q = (static_cast<size_t>(blockIdx.x) * 4 + wave) * 64 + lane;
  double cur[6][2], nx1[6][2], nx2[6][2];
  if (q < nvec) {
#pragma unroll
    for (int k = 0; k < 3; ++k) { load2<ST>(pl.x1[k], q, cur[k]); load2<ST>(pl.x2[k], q, cur[3 + k]); }
  }
  if (q + stride < nvec) {
#pragma unroll
    for (int k = 0; k < 3; ++k) { load2<ST>(pl.x1[k], q + stride, nx1[k]); load2<ST>(pl.x2[k], q + stride, nx1[3 + k]); }
  }
  while (q < nvec) {
    const size_t q2 = q + 2 * stride;
    if (q2 < nvec) {
#pragma unroll
      for (int k = 0; k < 3; ++k) { load2<ST>(pl.x1[k], q2, nx2[k]); load2<ST>(pl.x2[k], q2, nx2[3 + k]); }
    }
#pragma unroll
    for (int h = 0; h < 2; ++h)
      if (2 * q + h < n)   // the planes are zero-padded, so this only guards the count-exactness of a ragged tail
        add_match(cur[0][h], cur[1][h], cur[2][h], cur[3][h], cur[4][h], cur[5][h], acc);
#pragma unroll
    for (int k = 0; k < 6; ++k)
#pragma unroll
      for (int h = 0; h < 2; ++h) { cur[k][h] = nx1[k][h]; nx1[k][h] = nx2[k][h]; }
    q += stride;
  }
  // waves 2,3 -> LDS; waves 0,1 add them; wave 1 -> LDS; wave 0 adds and stores the block's row: a fixed order
  if (wave >= 2) {
#pragma unroll
    for (int k = 0; k < kSums; ++k) red[wave - 2][k][lane] = acc[k];
  }
  __syncthreads();
  if (wave < 2) {
#pragma unroll
    for (int k = 0; k < kSums; ++k) acc[k] += red[wave][k][lane];
  }
  __syncthreads();
  if (wave == 1) {
#pragma unroll
    for (int k = 0; k < kSums; ++k) red[0][k][lane] = acc[k];
  }
  __syncthreads();
  if (wave == 0) {
    double* row = partials + static_cast<size_t>(blockIdx.x) * kSums * 64;
#pragma unroll
    for (int k = 0; k < kSums; ++k) row[k * 64 + lane] = acc[k] + red[0][k][lane];
  }
}

// groups[lane][entry] = sum over blocks of partials[block][src(entry)][lane].  One 1024-thread block per entry: wave s
// folds blocks s, s+16, s+32, ... (four independent accumulators), wave 0 then adds the 16 segment sums in segment
// order -- a fixed order for a given grid, and 16 x 4 loads in flight per (entry, lane) instead of 4.
__global__ __launch_bounds__(1024) void epipolar_fold_kernel(const double* __restrict__ partials, int nblocks,
                                                             double* __restrict__ groups) {
  __shared__ double seg_sum[16][64];
  const int entry = blockIdx.x, lane = threadIdx.x & 63, seg = threadIdx.x >> 6;
  double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
  const size_t bs = static_cast<size_t>(kSums) * 64;
  const double* src = partials + static_cast<size_t>(kEntryMap.src[entry]) * 64 + lane;
  int b = seg;
  for (; b + 48 < nblocks; b += 64) {
    s0 += src[static_cast<size_t>(b) * bs];
    s1 += src[static_cast<size_t>(b + 16) * bs];
    s2 += src[static_cast<size_t>(b + 32) * bs];
    s3 += src[static_cast<size_t>(b + 48) * bs];
  }
  for (; b < nblocks; b += 16) s0 += src[static_cast<size_t>(b) * bs];
  seg_sum[seg][lane] = (s0 + s1) + (s2 + s3);
  __syncthreads();
  if (seg == 0) {
    double tot = seg_sum[0][lane];
#pragma unroll
    for (int k = 1; k < 16; ++k) tot += seg_sum[k][lane];
    groups[lane * kMom + entry] = tot;
  }
}

// ---- the same group moments for every pair of a batch (sba_batch_initial_guess) ------------------------------------------------
// One 512-thread block per pair: lane l of every wave streams the pair's vectors pr = 64 * k + l through the pair's tile map
// (sba_pair_map.hpp), so lane id == group id exactly as above (group = (match index within the pair / 2) % 64), and the
// eight waves combine through LDS in a fixed order (7,6 -> 5,4 -> 3,2 -> 1,0 -> 0).  groups[pair][lane][45].
template <typename ST>
__global__ __launch_bounds__(512) void batch_epipolar_moments_kernel(Planes pl, const PairDesc* __restrict__ desc,
                                                                     double* __restrict__ groups) {
  __shared__ double red[2][kSums][64];
  double acc[kSums];
#pragma unroll
  for (int k = 0; k < kSums; ++k) acc[k] = 0.0;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const PairDesc dsc = desc[blockIdx.x];
  const BatchPairMap<ST> map{dsc};
  const size_t n = dsc.n, nvec = (n + 1) / 2;
  constexpr size_t stride = 512;
  size_t q = threadIdx.x;
  double cur[6][2], nx1[6][2], nx2[6][2];
  if (q < nvec) {
    const size_t v = map(q);
#pragma unroll
    for (int k = 0; k < 3; ++k) { load2<ST>(pl.x1[k], v, cur[k]); load2<ST>(pl.x2[k], v, cur[3 + k]); }
  }
  if (q + stride < nvec) {
    const size_t v = map(q + stride);
#pragma unroll
    for (int k = 0; k < 3; ++k) { load2<ST>(pl.x1[k], v, nx1[k]); load2<ST>(pl.x2[k], v, nx1[3 + k]); }
  }
  while (q < nvec) {
    const size_t q2 = q + 2 * stride;
    if (q2 < nvec) {
      const size_t v = map(q2);
#pragma unroll
      for (int k = 0; k < 3; ++k) { load2<ST>(pl.x1[k], v, nx2[k]); load2<ST>(pl.x2[k], v, nx2[3 + k]); }
    }
#pragma unroll
    for (int h = 0; h < 2; ++h)
      if (2 * q + h < n) add_match(cur[0][h], cur[1][h], cur[2][h], cur[3][h], cur[4][h], cur[5][h], acc);
#pragma unroll
    for (int k = 0; k < 6; ++k)
#pragma unroll
      for (int h = 0; h < 2; ++h) { cur[k][h] = nx1[k][h]; nx1[k][h] = nx2[k][h]; }
    q += stride;
  }
  // waves (2s+1, 2s) hand their sums to waves (2s-1, 2s-2) through the two LDS slots, s = 3, 2, 1; then wave 1 -> wave 0
#pragma unroll
  for (int top = 6; top >= 2; top -= 2) {
    if (wave == top || wave == top + 1) {
#pragma unroll
      for (int k = 0; k < kSums; ++k) red[wave - top][k][lane] = acc[k];
    }
    __syncthreads();
    if (wave == top - 2 || wave == top - 1) {
#pragma unroll
      for (int k = 0; k < kSums; ++k) acc[k] += red[wave - (top - 2)][k][lane];
    }
    __syncthreads();
  }
  if (wave == 1) {
#pragma unroll
    for (int k = 0; k < kSums; ++k) red[0][k][lane] = acc[k];
  }
  __syncthreads();
  if (wave == 0) {
#pragma unroll
    for (int k = 0; k < kSums; ++k) red[1][k][lane] = acc[k] + red[0][k][lane];    // own column only: no barrier needed
    double* row = groups + (static_cast<size_t>(blockIdx.x) * 64 + lane) * kMom;
    for (int e = 0; e < kMom; ++e) row[e] = red[1][kEntryMap.src[e]][lane];
  }
}

// ---- the trials and the consensus pick of every pair of a batch, on the device ------------------------------------------------------
// One 256-thread block per pair, fed by the pair's 64 x 45 group moments (23 KB, L2-resident right after the pass above).
//   1. the moments are staged in LDS; thread 0: occupancy of the groups (epi::group_occupancy);
//   2. thread t runs trial t -- epi::group_trial, THE SOURCE the host runs (sba_epipolar.hpp is __host__ __device__, no FMA
//      contraction): draw the groups, sum their moments in ascending group order, null vector, rank-2 projection,
//      decomposeEssentialMat, Euler angles, validity;
//   3. thread 0 collects the valid candidates in trial order (R1 before R2, .cpp:148-157);
//   4. consensus pick (.cpp:162-180, epi::consensus_pick): wave w takes candidates w, w + 4, ...; its lanes compute the
//      distances to all r candidates, rank them (stable: value, then index) so that the 20-80 % middle can be summed in ascending
//      order -- the order the host's sort gives, hence the same bits -- lane 0 adds; thread 0 keeps the first minimum.
// At most kGuessMaxTrials trials (the reference runs 80); more are served by the host path.
constexpr int kGuessMaxCand = 2 * kGuessMaxTrials;

__global__ __launch_bounds__(256) void batch_guess_kernel(const double* __restrict__ groups, int trials, double fraction,
                                                          unsigned long long seed, BatchGuessOut* __restrict__ out) {
  __shared__ double g_s[epi::kGroups * epi::kMom];          // the pair's moments: every trial sums 16 of the 64 rows
  __shared__ epi::GroupOccupancy occ_s;
  // the trial records are dead once their candidates are collected; the consensus' sorted distances reuse the space
  constexpr size_t kSortedBytes = sizeof(double) * 2 * 4 * kGuessMaxCand;
  static_assert(sizeof(epi::TrialOut) * kGuessMaxTrials <= kSortedBytes, "trial records must fit the buffer they share");
  __shared__ alignas(16) unsigned char shared_buf[kSortedBytes];
  epi::TrialOut* trial_s = reinterpret_cast<epi::TrialOut*>(shared_buf);
  double (*sorted_s)[4][kGuessMaxCand] = reinterpret_cast<double (*)[4][kGuessMaxCand]>(shared_buf);
  __shared__ float ce[kGuessMaxCand][3], ct[kGuessMaxCand][3];
  __shared__ alignas(16) unsigned long long key_s[4][kGuessMaxCand];     // (bits of the squared distance) << 32 | index
  __shared__ double avg_s[kGuessMaxCand];
  __shared__ int r_s;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
#ifdef SBA_GUESS_PROFILE
  long long tk[6];
#define SBA_GTICK(i) tk[i] = wall_clock64()
#else
#define SBA_GTICK(i) do { } while (0)
#endif
  SBA_GTICK(0);
  {
    const double* g = groups + static_cast<size_t>(blockIdx.x) * epi::kGroups * epi::kMom;
    for (int k = tid; k < epi::kGroups * epi::kMom; k += 256) g_s[k] = g[k];
  }
  __syncthreads();
  if (tid == 0) epi::group_occupancy(g_s, fraction, &occ_s);
  __syncthreads();
  SBA_GTICK(1);
  if (tid < trials) {
    epi::TrialOut o;
    epi::group_trial(g_s, occ_s, seed, tid, &o);
    trial_s[tid] = o;
  }
  __syncthreads();
  SBA_GTICK(2);
  if (tid == 0) {
    int r = 0;
    for (int t = 0; t < trials; ++t) {
      const epi::TrialOut& o = trial_s[t];
      if (o.v1) { for (int i = 0; i < 3; ++i) { ce[r][i] = o.c1.euler[i]; ct[r][i] = o.c1.tran[i]; } ++r; }
      if (o.v2) { for (int i = 0; i < 3; ++i) { ce[r][i] = o.c2.euler[i]; ct[r][i] = o.c2.tran[i]; } ++r; }
    }
    r_s = r;
  }
  __syncthreads();
  SBA_GTICK(3);
  // Consensus.  Group gi = candidates 4 gi + wave, one per wave.  Per trip of the loop (all trip counts come from LDS:
  // block-uniform): A(gi) the wave's lanes write, for its candidate, the squared distance to every candidate j as the key
  // (float bits << 32 | j) -- non-negative floats order like their bit patterns, so key order IS the stable order (value,
  // then index) -- padded to an even count with all-ones; lane 0 adds up the sorted middle of group gi - 1 (C); barrier;
  // B(gi) every lane ranks its up to four keys against all of them (one 64-bit compare per pair, 16-byte LDS reads shared by
  // the four) and scatters the square roots into sorted order; barrier.  sorted_s is double-buffered by the parity of gi,
  // so C(gi - 1) and B(gi) never meet.
  const int r = r_s;
  const int lo = static_cast<int>(r * 0.2), hi = static_cast<int>(r * 0.8);
  const int r2 = (r + 1) & ~1, ngroups = (r + 3) / 4, nu = (r + 63) / 64;
  for (int gi = 0; gi <= ngroups; ++gi) {
    const int i = 4 * gi + wave;
    if (gi < ngroups && i < r)
      for (int j = lane; j < r2; j += 64) {
        unsigned long long key = ~0ull;
        if (j < r) {
          const float dx = ce[i][0] - ce[j][0], dy = ce[i][1] - ce[j][1], dz = ce[i][2] - ce[j][2];
          key = (static_cast<unsigned long long>(__float_as_uint(dx * dx + dy * dy + dz * dz)) << 32) | static_cast<unsigned>(j);
        }
        key_s[wave][j] = key;
      }
    if (gi > 0 && lane == 0 && i - 4 < r) {
      const double* sv = sorted_s[(gi - 1) & 1][wave];
      double acc = 0.0;
      int p = lo;
      for (; p + 8 <= hi; p += 8) {          // eight loads in flight, then the adds in ascending order
        const double v0 = sv[p], v1 = sv[p + 1], v2 = sv[p + 2], v3 = sv[p + 3], v4 = sv[p + 4], v5 = sv[p + 5], v6 = sv[p + 6], v7 = sv[p + 7];
        acc += v0; acc += v1; acc += v2; acc += v3; acc += v4; acc += v5; acc += v6; acc += v7;
      }
      for (; p < hi; ++p) acc += sv[p];
      avg_s[i - 4] = acc / (static_cast<double>(hi - lo) * 1.0);     // 0 / 0 = NaN for r < 2, as in the reference
    }
    __syncthreads();
    if (gi < ngroups && i < r) {
      unsigned long long kj[4] = {0, 0, 0, 0};
      int rank[4] = {0, 0, 0, 0};
#pragma unroll
      for (int u = 0; u < 4; ++u) kj[u] = (u < nu && lane + 64 * u < r) ? key_s[wave][lane + 64 * u] : 0ull;
      const ulonglong2* row = reinterpret_cast<const ulonglong2*>(key_s[wave]);
#pragma unroll 4
      for (int k2 = 0; k2 < r2 / 2; ++k2) {      // several 16-byte reads in flight per trip
        const ulonglong2 v = row[k2];
#pragma unroll
        for (int u = 0; u < 4; ++u)
          if (u < nu) rank[u] += (v.x < kj[u] ? 1 : 0) + (v.y < kj[u] ? 1 : 0);
      }
#pragma unroll
      for (int u = 0; u < 4; ++u)
        if (u < nu && lane + 64 * u < r)
          sorted_s[gi & 1][wave][rank[u]] = sqrt(static_cast<double>(__uint_as_float(static_cast<unsigned>(kj[u] >> 32))));
    }
    __syncthreads();
  }
  SBA_GTICK(4);
  if (tid == 0) {
    BatchGuessOut res{};
    res.num_candidates = r;
    res.status = r > 0 ? SBA_OK : SBA_ERR_NUMERIC;
    if (r > 0) {
      int best = 0;
      double best_d = avg_s[0];
      for (int i = 1; i < r; ++i)
        if (avg_s[i] < best_d) { best = i; best_d = avg_s[i]; }     // std::min_element keeps the first minimum
      for (int a = 0; a < 3; ++a) { res.euler[a] = ce[best][a]; res.tran[a] = ct[best][a]; }
    }
#ifdef SBA_GUESS_PROFILE
    // ticks of the 100 MHz wall clock: staging + occupancy, trials, collection, consensus, pick
    SBA_GTICK(5);
    res.euler[0] = static_cast<double>(tk[1] - tk[0]); res.euler[1] = static_cast<double>(tk[2] - tk[1]);
    res.euler[2] = static_cast<double>(tk[3] - tk[2]); res.tran[0] = static_cast<double>(tk[4] - tk[3]);
    res.tran[1] = static_cast<double>(tk[5] - tk[4]);
#endif
    out[blockIdx.x] = res;
  }
}

// ---- per-trial A^T A from index lists: the reference's own subsets (initial_guess, .cpp:130-141) ---------------------------
// Block t handles trial t: its 256 threads stride over the trial's `m` match indices (the first int(n * 0.25) entries of
// the reference's permutation, built on the host from the process's rand() stream), gather the six coordinates of each
// match from the planes, accumulate the 36 distinct sums, and fold them in a fixed order (lanes by butterfly, waves in
// wave order) into moments[t][45].  Meant for the reference's real problem sizes (thousands of matches: the whole pass
// is a few microseconds of scattered 8-byte loads); large problems use the streaming group pass above.
template <typename ST>
__device__ __forceinline__ double load1(const void* plane, size_t i) { return static_cast<double>(reinterpret_cast<const ST*>(plane)[i]); }

template <typename ST>
__global__ __launch_bounds__(256) void epipolar_subset_moments_kernel(Planes pl, unsigned long long n,
                                                                      const int* __restrict__ indices, int m,
                                                                      double* __restrict__ moments) {
  __shared__ double red[4][kSums];
  double acc[kSums];
#pragma unroll
  for (int k = 0; k < kSums; ++k) acc[k] = 0.0;
  const int* __restrict__ idx = indices + static_cast<size_t>(blockIdx.x) * m;
  for (int k = threadIdx.x; k < m; k += 256) {
    const unsigned long long i = static_cast<unsigned long long>(idx[k]);
    if (i < n)      // a list may only name resident matches (checked on the host too)
      add_match(load1<ST>(pl.x1[0], i), load1<ST>(pl.x1[1], i), load1<ST>(pl.x1[2], i),
                load1<ST>(pl.x2[0], i), load1<ST>(pl.x2[1], i), load1<ST>(pl.x2[2], i), acc);
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int k = 0; k < kSums; ++k) {
    double v = acc[k];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
    if (lane == 0) red[wave][k] = v;
  }
  __syncthreads();
  if (threadIdx.x < kMom) {
    const int src = kEntryMap.src[threadIdx.x];
    moments[static_cast<size_t>(blockIdx.x) * kMom + threadIdx.x] = ((red[0][src] + red[1][src]) + red[2][src]) + red[3][src];
  }
}

}  // namespace

// indices_dev: [trials][m] int32 match indices (< n); moments_dev: [trials][45].
hipError_t launch_epipolar_subset_moments(int store, const Planes& pl, size_t n, const int* indices_dev, int trials, int m,
                                          double* moments_dev, hipStream_t stream) {
  if (trials <= 0) return hipSuccess;
  if (store == 0)
    hipLaunchKernelGGL((epipolar_subset_moments_kernel<double>), dim3(trials), dim3(256), 0, stream, pl,
                       static_cast<unsigned long long>(n), indices_dev, m, moments_dev);
  else
    hipLaunchKernelGGL((epipolar_subset_moments_kernel<float>), dim3(trials), dim3(256), 0, stream, pl,
                       static_cast<unsigned long long>(n), indices_dev, m, moments_dev);
  return hipGetLastError();
}

// groups_dev: [num_pairs][64][45] doubles.
hipError_t launch_batch_epipolar_moments(int store, const Planes& pl, const PairDesc* desc, int num_pairs, double* groups_dev,
                                         hipStream_t stream) {
  if (num_pairs <= 0) return hipSuccess;
  if (store == 0)
    hipLaunchKernelGGL((batch_epipolar_moments_kernel<double>), dim3(num_pairs), dim3(512), 0, stream, pl, desc, groups_dev);
  else
    hipLaunchKernelGGL((batch_epipolar_moments_kernel<float>), dim3(num_pairs), dim3(512), 0, stream, pl, desc, groups_dev);
  return hipGetLastError();
}

hipError_t launch_batch_guess(const double* groups_dev, int num_pairs, int trials, double fraction, unsigned long long seed,
                              BatchGuessOut* out_dev, hipStream_t stream) {
  if (num_pairs <= 0) return hipSuccess;
  if (trials < 1 || trials > kGuessMaxTrials) return hipErrorInvalidValue;
  hipLaunchKernelGGL(batch_guess_kernel, dim3(num_pairs), dim3(256), 0, stream, groups_dev, trials, fraction, seed, out_dev);
  return hipGetLastError();
}

// groups_dev: [64][45] doubles; partials: [grid][36][64] doubles scratch (sized for 45 by the caller).
hipError_t launch_epipolar_moments(int store, const Planes& pl, size_t n, double* partials, int grid,
                                   double* groups_dev, hipStream_t stream) {
  if (grid > 0) {
    if (store == 0)
      hipLaunchKernelGGL((epipolar_moments_kernel<double>), dim3(grid), dim3(256), 0, stream, pl,
                         static_cast<unsigned long long>(n), partials);
    else
      hipLaunchKernelGGL((epipolar_moments_kernel<float>), dim3(grid), dim3(256), 0, stream, pl,
                         static_cast<unsigned long long>(n), partials);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
  }
  hipLaunchKernelGGL(epipolar_fold_kernel, dim3(kMom), dim3(1024), 0, stream, partials, grid, groups_dev);
  return hipGetLastError();
}

}  // namespace sba
