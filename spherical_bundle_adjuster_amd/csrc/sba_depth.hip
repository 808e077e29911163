// d-only stage on the device (reference spherical_bundle_adjuster.cpp:1004-1063, first stage of
// solve_problem, .cpp:196-197).
//
// The reference builds ONE Ceres problem of N residual blocks (5 residuals: the reprojection 3-vector and the
// two regularisers lambda*exp(-c*d), .cpp:1008-1029) over N private 2-parameter blocks with lower bound 0
// (.cpp:1060-1061) and no loss (.cpp:1059).  The Hessian is block diagonal, so the damped system of every LM
// iteration splits into N independent 2x2 solves -- but radius, accept/reject, Jacobi scaling and the convergence
// tests are global.  depth_step_kernel therefore does, for every match in one pass: residuals + 5x2 Jacobian at the
// current depths, the scaled damped 2x2 solve (the step delta), the projected candidate P(d + alpha delta), the
// candidate's cost and gradient, and contributes to the nine global reductions the host needs for Ceres' step logic
// and its projected Armijo line search (sba_stages.cpp: sba_problem_solve_depths; the problem is bounds-constrained,
// so Ceres line-searches every step).  alpha = 1 is the trust-region step and at the same time the line search's
// first trial; a contraction re-runs the pass with alpha < 1 and the stored diagonal (same delta, bit for bit).
// Two matches per lane, 16-byte accesses (1 KiB per wave instruction) -- HBM-bound: 8-12 loads + 4-8 stores of
// 8 B per match and iteration (96-128 B).
#include <cstdlib>

#include "sba_device.hpp"
#include "sba_pair_map.hpp"
#include "sba_resident.hpp"

namespace sba {
namespace {

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
  return v;
}
__device__ __forceinline__ double wave_max(double v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v = fmax(v, __shfl_xor(v, off));
  return v;
}

// Two consecutive matches per lane: 16-byte accesses on the f64 planes (coordinates with f64 storage, depths,
// scaling, diagonal, candidates), 8-byte on f32 coordinate planes; the once-read streams are loaded non-temporally.
template <typename ST> struct Pair;
template <> struct Pair<double> {
  static __device__ __forceinline__ void load(const void* plane, size_t pair, double out[2]) {
    typedef float f4 __attribute__((ext_vector_type(4)));
    const f4 r = __builtin_nontemporal_load(reinterpret_cast<const f4*>(plane) + pair);
    const double2 q = *reinterpret_cast<const double2*>(&r);
    out[0] = q.x; out[1] = q.y;
  }
};
template <> struct Pair<float> {
  static __device__ __forceinline__ void load(const void* plane, size_t pair, double out[2]) {
    const float2 q = reinterpret_cast<const float2*>(plane)[pair];
    out[0] = q.x; out[1] = q.y;
  }
};
__device__ __forceinline__ void store_pair_f64(double* plane, size_t pair, double a, double b) {
  reinterpret_cast<double2*>(plane)[pair] = make_double2(a, b);
}

__device__ __forceinline__ void store_pair_stream(double* plane, size_t pair, double a, double b) {
  typedef float f4 __attribute__((ext_vector_type(4)));
  const double2 q = make_double2(a, b);
  __builtin_nontemporal_store(*reinterpret_cast<const f4*>(&q), reinterpret_cast<f4*>(plane) + pair);
}

// Everything one lane reads for its two matches of one grid-stride step.  Loaded one step ahead of its use (register
// double buffer, like the sweep kernel): the ~300 VALU instructions per match (two exp at the current depths, two at
// the candidate, one reciprocal) then run under the next step's loads instead of after them.
template <typename ST>
struct DepthRegs {
  double X[2], Y[2], Z[2], U[2], V[2], W[2], A[2], B[2], S1[2], S2[2];
  __device__ __forceinline__ void load(const Planes& pl, const double* d1, const double* d2, const double* sc1,
                                       const double* sc2, bool load_scale, size_t pr) {
    Pair<ST>::load(pl.x1[0], pr, X); Pair<ST>::load(pl.x1[1], pr, Y); Pair<ST>::load(pl.x1[2], pr, Z);
    Pair<ST>::load(pl.x2[0], pr, U); Pair<ST>::load(pl.x2[1], pr, V); Pair<ST>::load(pl.x2[2], pr, W);
    Pair<double>::load(d1, pr, A); Pair<double>::load(d2, pr, B);
    if (load_scale) { Pair<double>::load(sc1, pr, S1); Pair<double>::load(sc2, pr, S2); }
  }
};

// One lane's share of a pass: its pairs of matches pr, pr + stride, ... -- residuals, Jacobian, damped 2x2 solve, projected
// candidate (stored to c1 / c2), candidate cost and gradient -- accumulated into the nine per-lane partial reductions
// r[DEPTH_OUT_*].  Shared by the grid-wide kernel (stride = grid size) and the resident single-block kernel (stride = 256).
// AHEAD = grid-stride steps whose loads are in flight while the current one is computed on.  A step is ~640 f64 VALU
// instructions per lane (~4 us per wave), so with one step ahead a wave's loads have long landed before it asks for the
// next ones -- would two steps ahead (+40 VGPRs: 227, still two waves per SIMD) feed the memory system better?  Measured:
// no -- 177.4 us per pass against 169.1 us (rocprofv3 averages over 54 launches, same box; profiles/r03_depth_ahead.log).
// One step ahead is the default; SBA_DEPTH_AHEAD=2 keeps the variant reachable.
// MAP: logical pair-of-matches index -> index into the planes (sba_pair_map.hpp: identity for a single problem; the batch's
// pair layout, contiguous or interleaved tiles, for a pair of a batch).
template <typename ST, int AHEAD, typename MAP = IdentityMap>
__device__ __forceinline__ void depth_stream(const Planes& pl, const double* __restrict__ d1, const double* __restrict__ d2,
                                             double* __restrict__ c1, double* __restrict__ c2, double* __restrict__ sc1,
                                             double* __restrict__ sc2, const DepthParams& P, size_t pr, size_t stride,
                                             double r[DEPTH_OUT_COUNT], const MAP map = MAP()) {
  static_assert(AHEAD == 1 || AHEAD == 2, "one or two steps of loads in flight");
  const size_t npairs = (P.n + 1) / 2;     // the planes are zero-padded to a whole vector (+ one spare)
  const bool load_scale = !P.first_iteration;
  double cost = 0, model = 0, cand_cost = 0, step2 = 0, x2n = 0, gdelta = 0, cand_gdelta = 0, gmax = 0, dmax = 0;
  DepthRegs<ST> cur, nxt, nx2;
  if (pr < npairs) cur.load(pl, d1, d2, sc1, sc2, load_scale, map(pr));
  if (AHEAD == 2 && pr + stride < npairs) nxt.load(pl, d1, d2, sc1, sc2, load_scale, map(pr + stride));
  while (pr < npairs) {
    const size_t pn = pr + stride, q = map(pr);           // q: where this lane's two matches sit in the planes
    if (AHEAD == 2) {
      if (pn + stride < npairs) nx2.load(pl, d1, d2, sc1, sc2, load_scale, map(pn + stride));
    } else {
      if (pn < npairs) nxt.load(pl, d1, d2, sc1, sc2, load_scale, map(pn));
    }
    double NA[2], NB[2];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const bool valid = 2 * pr + h < P.n;
      const double x = cur.X[h], y = cur.Y[h], z = cur.Z[h], u = cur.U[h], v = cur.V[h], w = cur.W[h];
      const double a = cur.A[h], b = cur.B[h];
      // q = R x1 ; e = b x2 - a q + t                                                   (.cpp:1008-1027)
      const double q0 = P.R[0] * x + P.R[1] * y + P.R[2] * z;
      const double q1 = P.R[3] * x + P.R[4] * y + P.R[5] * z;
      const double q2 = P.R[6] * x + P.R[7] * y + P.R[8] * z;
      const double e0 = b * u - a * q0 + P.t[0], e1 = b * v - a * q1 + P.t[1], e2 = b * w - a * q2 + P.t[2];
      const double r4 = P.lambda * exp(-P.c * a), r5 = P.lambda * exp(-P.c * b);       // .cpp:1028-1029
      // J = [[-q, x2], [-c r4, 0], [0, -c r5]]
      const double j41 = -P.c * r4, j52 = -P.c * r5;
      const double qq = q0 * q0 + q1 * q1 + q2 * q2, qx = q0 * u + q1 * v + q2 * w, xx = u * u + v * v + w * w;
      const double qe = q0 * e0 + q1 * e1 + q2 * e2, xe = u * e0 + v * e1 + w * e2;
      const double h11 = qq + j41 * j41, h12 = -qx, h22 = xx + j52 * j52;
      const double g1 = -qe + j41 * r4, g2 = xe + j52 * r5;
      double s1, s2;
      if (P.first_iteration) {
        s1 = P.jacobi_scaling ? 1.0 / (1.0 + sqrt(h11)) : 1.0;
        s2 = P.jacobi_scaling ? 1.0 / (1.0 + sqrt(h22)) : 1.0;
        cur.S1[h] = s1; cur.S2[h] = s2;
      } else {
        s1 = cur.S1[h]; s2 = cur.S2[h];
      }
      const double H11 = s1 * h11 * s1, H12 = s1 * h12 * s2, H22 = s2 * h22 * s2, G1 = s1 * g1, G2 = s2 * g2;
      // The LM diagonal.  Ceres keeps the squared column norms of the last accepted point across rejected steps and
      // line-search trials (reuse_diagonal_) -- but those passes run at that very point, so recomputing them here gives
      // the same bits and the two planes that used to carry them (16 B per match and pass) are gone.
      const double D1 = fmin(fmax(H11, P.min_diagonal), P.max_diagonal);
      const double D2 = fmin(fmax(H22, P.min_diagonal), P.max_diagonal);
      // damped 2x2 system (H + D / radius) y = -G; 1 / radius comes from the host, 1 / det is formed once
      const double A11 = __builtin_fma(D1, P.inv_radius, H11), A22 = __builtin_fma(D2, P.inv_radius, H22), A12 = H12;
      const double inv_det = 1.0 / (A11 * A22 - A12 * A12);
      const double y1 = (G2 * A12 - G1 * A22) * inv_det, y2 = (G1 * A12 - G2 * A11) * inv_det;
      const double dl1 = s1 * y1, dl2 = s2 * y2;                               // delta = step .* jacobi scaling
      // Plus(d, alpha * delta) + projection onto d >= 0 (alpha = 1: the trust-region candidate)
      const double na = fmax(a + P.alpha * dl1, 0.0), nb = fmax(b + P.alpha * dl2, 0.0);
      NA[h] = na; NB[h] = nb;
      const double f0 = nb * u - na * q0 + P.t[0], f1 = nb * v - na * q1 + P.t[1], f2 = nb * w - na * q2 + P.t[2];
      const double f4 = P.lambda * exp(-P.c * na), f5 = P.lambda * exp(-P.c * nb);
      // gradient at the candidate (for the slope of the line-search function at alpha)
      const double cg1 = -(q0 * f0 + q1 * f1 + q2 * f2) - P.c * f4 * f4, cg2 = (u * f0 + v * f1 + w * f2) - P.c * f5 * f5;
      if (valid) {     // the padding element of an odd-sized problem contributes nothing
        cost += 0.5 * (e0 * e0 + e1 * e1 + e2 * e2 + r4 * r4 + r5 * r5);
        // projected gradient norm of the bounded problem: |x - P(x - g)|_inf
        gmax = fmax(gmax, fmax(fabs(a - fmax(a - g1, 0.0)), fabs(b - fmax(b - g2, 0.0))));
        model += -(G1 * y1 + G2 * y2) - 0.5 * (H11 * y1 * y1 + 2.0 * H12 * y1 * y2 + H22 * y2 * y2);
        step2 += (na - a) * (na - a) + (nb - b) * (nb - b);
        x2n += a * a + b * b;
        cand_cost += 0.5 * (f0 * f0 + f1 * f1 + f2 * f2 + f4 * f4 + f5 * f5);
        gdelta += g1 * dl1 + g2 * dl2;
        cand_gdelta += cg1 * dl1 + cg2 * dl2;
        dmax = fmax(dmax, fmax(fabs(dl1), fabs(dl2)));
      } else {
        NA[h] = 0.0; NB[h] = 0.0;   // keep the padding zero
      }
    }
    if (P.stream_stores) { store_pair_stream(c1, q, NA[0], NA[1]); store_pair_stream(c2, q, NB[0], NB[1]); }
    else { store_pair_f64(c1, q, NA[0], NA[1]); store_pair_f64(c2, q, NB[0], NB[1]); }
    if (P.first_iteration) { store_pair_f64(sc1, q, cur.S1[0], cur.S1[1]); store_pair_f64(sc2, q, cur.S2[0], cur.S2[1]); }
    cur = nxt;
    if (AHEAD == 2) nxt = nx2;
    pr = pn;
  }
  r[0] = cost; r[1] = model; r[2] = cand_cost; r[3] = step2; r[4] = x2n; r[5] = gdelta; r[6] = cand_gdelta; r[7] = gmax; r[8] = dmax;
}
static_assert(DEPTH_OUT_COUNT == 9 && DEPTH_OUT_SUMS == 7, "depth_stream fills slots 0..6 (sums) and 7, 8 (maxima)");

// Block-level fold of the per-lane partials: lanes by butterfly, the four waves in wave order; res[0..8] valid on
// threads < DEPTH_OUT_COUNT after the call (which contains one barrier).
template <int NW = 4>
__device__ __forceinline__ double depth_block_fold(const double r[DEPTH_OUT_COUNT], double (*red)[DEPTH_OUT_COUNT]) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int k = 0; k < DEPTH_OUT_COUNT; ++k) {
    const double v = k < DEPTH_OUT_SUMS ? wave_sum(r[k]) : wave_max(r[k]);
    if (lane == 0) red[wave][k] = v;
  }
  __syncthreads();
  double s = 0.0;
  if (threadIdx.x < DEPTH_OUT_COUNT) {
    const bool is_max = threadIdx.x >= DEPTH_OUT_SUMS;
    s = red[0][threadIdx.x];
    for (int wv = 1; wv < NW; ++wv) s = is_max ? fmax(s, red[wv][threadIdx.x]) : s + red[wv][threadIdx.x];
  }
  return s;
}

// Two resident 256-thread blocks per CU: 187 VGPRs hold both matches' temporaries and the whole register double buffer.
// Holding the allocation to 3 / 4 blocks per CU (168 / 128 VGPRs, 84 / 244 B of scratch) was measured and is far worse:
// 231-279 / 499-517 us per pass against 185 us (profiles/r03_depth_tune.log) -- the spills sit in the hot loop.
template <typename ST, int AHEAD>
__global__ __launch_bounds__(256, 2) void depth_step_kernel(Planes pl, const double* __restrict__ d1,
                                                           const double* __restrict__ d2,
                                                           double* __restrict__ c1, double* __restrict__ c2,
                                                           double* __restrict__ sc1, double* __restrict__ sc2,
                                                           DepthParams P, double* __restrict__ partials) {
  __shared__ double red[4][DEPTH_OUT_COUNT];
  double r[DEPTH_OUT_COUNT];
  depth_stream<ST, AHEAD>(pl, d1, d2, c1, c2, sc1, sc2, P, static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x,
                          static_cast<size_t>(gridDim.x) * blockDim.x, r);
  const double s = depth_block_fold(r, red);
  if (threadIdx.x < DEPTH_OUT_COUNT) partials[static_cast<size_t>(blockIdx.x) * DEPTH_ROW + threadIdx.x] = s;
}

// ---- resident evaluator of the d-only stage of ONE small problem (sba_resident.hpp) ------------------------------------------
// One block, resident for the whole stage; per command one pass over all matches.  Command payload: [0] opcode,
// [1..19] R, t, lambda, c, radius, 1 / radius, min / max diagonal, alpha, [20] flags (bit 0 first_iteration, bit 1
// reuse_diagonal, bit 2 jacobi_scaling, bit 3 `flip`: the current depths are the b planes and the candidates go to the a
// planes), [21] n (as bits).  Loop shape as resident_sweep_kernel.
template <typename ST>
__global__ __launch_bounds__(256, 2) void resident_depth_kernel(Planes pl, unsigned long long n_resident, double* __restrict__ a1,
                                                               double* __restrict__ a2, double* __restrict__ b1,
                                                               double* __restrict__ b2, double* __restrict__ sc1,
                                                               double* __restrict__ sc2, const ResidentRecord* __restrict__ rec,
                                                               double* __restrict__ host_pack, unsigned long long first_cmd_seq,
                                                               unsigned long long first_pack_seq, unsigned long long idle_ticks) {
  __shared__ double red[4][DEPTH_OUT_COUNT];
  __shared__ double res_s[DEPTH_ROW];
  __shared__ double cmd_s[kResidentPayload];
  __shared__ int end_s;
  const int tid = threadIdx.x;
  if (tid == 0) end_s = 0;
  int trip = 0;
  for (; trip < kResidentMaxTrips; ++trip) {
    __syncthreads();                                   // B0: the previous pass's reductions are in res_s
    if (tid < 64) {
      if (trip > 0) resident_publish(host_pack, res_s, DEPTH_OUT_COUNT, first_pack_seq + static_cast<unsigned long long>(trip - 1));
      int end = 0;
      resident_wait_command(rec, first_cmd_seq + static_cast<unsigned long long>(trip), idle_ticks, cmd_s, &end);
      if (tid == 0 && end) end_s = end;
    }
    __syncthreads();                                   // B1
    if (end_s) break;
    const int op = static_cast<int>(cmd_s[0]);
    if (op != RESIDENT_OP_DEPTH) {
      if (tid == 0) end_s = op == RESIDENT_OP_QUIT ? RESIDENT_END_QUIT : RESIDENT_END_BAD_OP;
      break;
    }
    DepthParams P;
#pragma unroll
    for (int k = 0; k < 9; ++k) P.R[k] = cmd_s[1 + k];
#pragma unroll
    for (int k = 0; k < 3; ++k) P.t[k] = cmd_s[10 + k];
    P.lambda = cmd_s[13]; P.c = cmd_s[14]; P.radius = cmd_s[15]; P.inv_radius = cmd_s[16];
    P.min_diagonal = cmd_s[17]; P.max_diagonal = cmd_s[18]; P.alpha = cmd_s[19];
    const int flags = static_cast<int>(cmd_s[20]);
    P.first_iteration = flags & 1; P.reuse_diagonal = (flags >> 1) & 1; P.jacobi_scaling = (flags >> 2) & 1; P.stream_stores = 0;
    const bool flip = (flags >> 3) & 1;
    unsigned long long n_cmd;
    const double nd = cmd_s[21];
    __builtin_memcpy(&n_cmd, &nd, sizeof(n_cmd));
    P.n = n_cmd < n_resident ? n_cmd : n_resident;
    double r[DEPTH_OUT_COUNT];
    depth_stream<ST, 1>(pl, flip ? b1 : a1, flip ? b2 : a2, flip ? a1 : b1, flip ? a2 : b2, sc1, sc2, P, static_cast<size_t>(tid), 256, r);
    const double s = depth_block_fold(r, red);
    if (tid < DEPTH_OUT_COUNT) res_s[tid] = s;
  }
  __syncthreads();
  if (tid < 64) {
    // trip budget used up: the last trip's answer has not been published yet (answers go out at the top of the next trip)
    if (trip == kResidentMaxTrips) resident_publish(host_pack, res_s, DEPTH_OUT_COUNT, first_pack_seq + static_cast<unsigned long long>(trip - 1));
    resident_publish_end(host_pack, end_s ? end_s : RESIDENT_END_TRIPS);
  }
}

// ---- batched d-only stage (config C5: the first stage of solve_problem for every pair of a batch) --------------------------------
// One 512-thread block per pair (8 waves per CU, like two blocks of the single-problem kernel): thread 0 reads the pair's
// pass record (radius, step size, flags) from mapped host memory, the block runs the pair's pass with the pair's own
// rotation / translation, folds the nine reductions and publishes them to mapped host memory; the block that delivers the
// last pair stores the sequence word the host polls (as batch_step_kernel).  Every pair has its own trust region, line search
// and convergence -- B independent DepthStageSolvers advance in lock-step on the host (sba_batch.cpp).  flags: bit 0 first
// pass (store the Jacobi scaling), bit 1 keep the diagonal, bit 2 jacobi scaling on, bit 3 `flip` (the pair's current
// depths are in the work planes, its candidates go to the batch's depth planes); n == 0: the pair is finished, skip it.
template <typename ST>
__global__ __launch_bounds__(512, 2) void batch_depth_step_kernel(Planes pl, const PairDesc* __restrict__ desc,
                                                                 const BatchDepthConst* __restrict__ cst,
                                                                 const BatchDepthPass* __restrict__ pass, double lambda, double c,
                                                                 double min_diagonal, double max_diagonal,
                                                                 double* __restrict__ a1, double* __restrict__ a2,
                                                                 double* __restrict__ b1, double* __restrict__ b2,
                                                                 double* __restrict__ sc1, double* __restrict__ sc2,
                                                                 double* __restrict__ out_host, unsigned int* __restrict__ ticket,
                                                                 unsigned long long seq) {
  __shared__ double red[8][DEPTH_OUT_COUNT];
  __shared__ BatchDepthPass pass_s;
  const unsigned pair = blockIdx.x;
  const int tid = threadIdx.x;
  if (tid == 0) pass_s = pass[pair];
  __syncthreads();
  const BatchDepthPass ps = pass_s;
  const PairDesc dsc = desc[pair];
  const size_t n = ps.n < dsc.n ? ps.n : dsc.n;          // never past the pair's own matches, whatever the record says
  double s = 0.0;
  if (n > 0) {
    DepthParams P;
    const BatchDepthConst k = cst[pair];
#pragma unroll
    for (int i = 0; i < 9; ++i) P.R[i] = k.R[i];
#pragma unroll
    for (int i = 0; i < 3; ++i) P.t[i] = k.t[i];
    P.lambda = lambda; P.c = c; P.radius = ps.radius; P.inv_radius = 1.0 / ps.radius;
    P.min_diagonal = min_diagonal; P.max_diagonal = max_diagonal; P.alpha = ps.alpha;
    P.first_iteration = ps.flags & 1; P.reuse_diagonal = (ps.flags >> 1) & 1; P.jacobi_scaling = (ps.flags >> 2) & 1;
    P.stream_stores = 0; P.n = n;
    const bool flip = (ps.flags >> 3) & 1;
    double r[DEPTH_OUT_COUNT];
    depth_stream<ST, 1, BatchPairMap<ST>>(pl, flip ? b1 : a1, flip ? b2 : a2, flip ? a1 : b1, flip ? a2 : b2, sc1, sc2, P,
                                          static_cast<size_t>(tid), 512, r, BatchPairMap<ST>{dsc});
    s = depth_block_fold<8>(r, red);
  }
  if (tid >= 64) return;                  // wave 0 finishes alone
  if (tid < DEPTH_OUT_COUNT) host_store(out_host + static_cast<size_t>(pair) * DEPTH_ROW + tid, s);
  host_release();
  if (tid == 0 && __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT) == gridDim.x - 1) {
    __hip_atomic_store(ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");
    __hip_atomic_store(reinterpret_cast<unsigned long long*>(out_host + static_cast<size_t>(gridDim.x) * DEPTH_ROW), seq,
                       __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}

// Per-pair state of the d-only stage that outlives a launch (device memory): with a pass cap the one-launch kernel below hands
// a pair that is not done within its first passes over to the launches with dynamic shares (batch_depth_dyn_kernel ...).
struct BatchDepthDynState {
  DepthStageSolver solver;
  int flip;                 // the pair's current depths are in the work planes
  int passes;
};
static_assert(sizeof(DepthStageSolver) % 8 == 0 && sizeof(BatchDepthDynState) % 8 == 0, "copied as 8-byte words");
struct BatchDepthCont {     // all null: the kernel runs every pair to the end
  BatchDepthDynState* state;
  BatchDepthPass* req;      // the pair's next pass (flags: bit 0 first, bit 1 keep diagonal, bit 3 flip)
  int* done;
  unsigned char* finish;    // what batch_depth_finish_kernel still has to do for the pair: bit 0 copy back, bit 1 write `out`
};

// The whole d-only stage of a pair in ONE launch: the block that owns the pair keeps the pair's DepthStageSolver (the very
// class the host drives -- sba_depth_solver.hpp and sba_line_search.hpp are __host__ __device__) in LDS, thread 0 feeds it the
// nine reductions of the pass just run and takes the next request from it, the block runs that pass.  No host round trip
// per pass and no lock-step: a pair that converges after 9 passes frees its CU slot while its neighbour runs 36.  At the
// end the block copies the pair's depths back into the batch's own planes when they ended up in the work planes, writes
// them out in init_d layout (out != nullptr), and thread 0 delivers summary / status through the pair's mapped host record
// (the record of batch_lm_kernel; rot / tran are not used, d1 / d2 carry the refined depths of the pair's first two matches in
// image 1, pad_ the number of passes).
// Loop shape as batch_lm_kernel: the thread-0 region of a trip sits between two barriers of that trip, the trip bound is
// a counter every thread keeps, the exit decision is read from LDS after the second barrier.
template <typename ST>
__global__ __launch_bounds__(512, 2) void batch_depth_solve_kernel(Planes pl, const PairDesc* __restrict__ desc,
                                                                  const BatchDepthConst* __restrict__ cst, double lambda, double c,
                                                                  sba_lm_options opt, double* __restrict__ a1, double* __restrict__ a2,
                                                                  double* __restrict__ b1, double* __restrict__ b2,
                                                                  double* __restrict__ sc1, double* __restrict__ sc2,
                                                                  const unsigned long long* __restrict__ offsets,
                                                                  double* __restrict__ out, BatchLmIo* __restrict__ io,
                                                                  unsigned int* __restrict__ ticket,
                                                                  unsigned long long* __restrict__ seq_host, unsigned long long seq,
                                                                  int max_trips, BatchDepthCont cont) {
  __shared__ double red[8][DEPTH_OUT_COUNT];
  __shared__ double res_s[DEPTH_ROW];
  __shared__ double req_s[2];                 // radius, alpha of the next pass
  __shared__ int flags_s, done_s, flip_s, passes_s;
  __shared__ alignas(16) unsigned char solver_mem[sizeof(DepthStageSolver)];
  DepthStageSolver* solver = reinterpret_cast<DepthStageSolver*>(solver_mem);
  const unsigned pair = blockIdx.x;
  const int tid = threadIdx.x;
  const PairDesc dsc = desc[pair];
  const BatchPairMap<ST> map{dsc};
  DepthParams P;
  {
    const BatchDepthConst k = cst[pair];
#pragma unroll
    for (int i = 0; i < 9; ++i) P.R[i] = k.R[i];
#pragma unroll
    for (int i = 0; i < 3; ++i) P.t[i] = k.t[i];
  }
  P.lambda = lambda; P.c = c; P.min_diagonal = opt.min_lm_diagonal; P.max_diagonal = opt.max_lm_diagonal;
  P.jacobi_scaling = opt.jacobi_scaling ? 1 : 0; P.stream_stores = 0; P.n = dsc.n;
  if (tid == 0) {
    new (solver) DepthStageSolver();
    solver->start(opt);
    flip_s = 0;
    passes_s = 0;
  }
  for (int trip = 0; trip <= max_trips; ++trip) {      // the trip after the last pass only feeds its reductions (and breaks below)
    __syncthreads();                        // B0: the previous pass's reductions are in res_s
    if (tid == 0) {
      if (trip > 0) {
        solver->feed(res_s);
        if (solver->take_candidate()) flip_s ^= 1;
      }
      done_s = solver->done() ? 1 : 0;
      if (!done_s) {
        const DepthPassRequest& rq = solver->request();
        req_s[0] = rq.radius; req_s[1] = rq.alpha;
        flags_s = (rq.first ? 1 : 0) | (rq.keep_diagonal ? 2 : 0) | (flip_s ? 8 : 0);
        if (trip < max_trips) ++passes_s;
      }
    }
    __syncthreads();                        // B1: done_s / the next pass are visible to the block
    if (done_s || trip == max_trips) break; // both block-uniform: an LDS word read after B1, the trip counter
    const int fl = flags_s;
    P.radius = req_s[0]; P.inv_radius = 1.0 / P.radius; P.alpha = req_s[1];
    P.first_iteration = fl & 1; P.reuse_diagonal = (fl >> 1) & 1;
    const bool flip = (fl >> 3) & 1;
    double r[DEPTH_OUT_COUNT];
    depth_stream<ST, 1, BatchPairMap<ST>>(pl, flip ? b1 : a1, flip ? b2 : a2, flip ? a1 : b1, flip ? a2 : b2, sc1, sc2, P,
                                          static_cast<size_t>(tid), 512, r, map);
    const double s = depth_block_fold<8>(r, red);
    if (tid < DEPTH_OUT_COUNT) res_s[tid] = s;
  }
  __syncthreads();                          // flip_s is final; every store of the last pass has been issued by its thread
  const bool handed_over = !done_s && cont.state != nullptr;      // the pass cap ended the loop: the pair goes on in the dynamic launches
  if (!handed_over) {
    const bool fl = flip_s != 0;            // as batch_depth_finish_kernel, for this block's own pair
    const size_t npairs = (dsc.n + 1) / 2;
    if (fl || out)
      for (size_t pr = tid; pr < npairs; pr += 512) {
        const size_t q = map(pr);
        const double2 u = reinterpret_cast<const double2*>(fl ? b1 : a1)[q], v = reinterpret_cast<const double2*>(fl ? b2 : a2)[q];
        if (fl) { reinterpret_cast<double2*>(a1)[q] = u; reinterpret_cast<double2*>(a2)[q] = v; }
        if (out) {
          const size_t i = offsets[pair] + 2 * pr;
          reinterpret_cast<double2*>(out)[i] = make_double2(u.x, v.x);
          if (2 * pr + 1 < dsc.n) reinterpret_cast<double2*>(out)[i + 1] = make_double2(u.y, v.y);
        }
      }
  }
  if (tid == 0) {
    if (handed_over) {
      BatchDepthDynState* st = cont.state + pair;
      for (unsigned k = 0; k < sizeof(DepthStageSolver) / 8; ++k)
        reinterpret_cast<double*>(&st->solver)[k] = reinterpret_cast<const double*>(solver_mem)[k];
      st->flip = flip_s; st->passes = passes_s + 1;      // passes whose request has been issued: the pending one counts
      cont.req[pair] = BatchDepthPass{req_s[0], req_s[1], dsc.n, static_cast<unsigned>(flags_s), 0u};
    }
    if (cont.done) cont.done[pair] = handed_over ? 0 : 1;
    if (cont.finish) cont.finish[pair] = 0;            // a pair that ends here has done its own copy-back and output above
    BatchLmIo res{};
    res.summary = solver->summary();
    res.status = handed_over ? 1 : (solver->done() ? solver->status() : SBA_ERR_NUMERIC);   // 1: continues; else the bound ran out: never silent
    res.pad_ = passes_s;
    // init_d[0][0] and init_d[1][0] of the pair: what the reference's rot / tran stages use as the depths of EVERY match
    // (.cpp:941-942, :998-999).  Thread 0 stored both itself (its lane handles the pair's first two matches).
    if (dsc.n > 0) {
      const double2 u0 = reinterpret_cast<const double2*>(flip_s ? b1 : a1)[map(0)];
      res.d1 = u0.x; res.d2 = dsc.n > 1 ? u0.y : u0.x;
    }
    io[pair] = res;
    if (seq_host) {                         // completion as batch_lm_kernel: record in host memory, then a ticket
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      if (__hip_atomic_fetch_add(ticket, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT) == gridDim.x - 1) {
        __hip_atomic_store(ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");
        __hip_atomic_store(seq_host, seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      }
    }
  }
}

// After the stage: pairs whose result sits in the work planes (flip) are copied back into the batch's depth planes, and --
// when out != nullptr -- every pair's depths are written in init_d layout (double[total][2], pair g at offsets[g]).
template <typename ST>
__global__ __launch_bounds__(256) void batch_depth_finish_kernel(const PairDesc* __restrict__ desc, const unsigned char* __restrict__ flip,
                                                                double* __restrict__ a1, double* __restrict__ a2,
                                                                const double* __restrict__ b1, const double* __restrict__ b2,
                                                                const unsigned long long* __restrict__ offsets, double* __restrict__ out) {
  const unsigned pair = blockIdx.x;
  const PairDesc dsc = desc[pair];
  const BatchPairMap<ST> map{dsc};
  const bool fl = (flip[pair] & 1) != 0;       // bit 0: copy back; bit 1: write `out` (a pair with neither is already complete)
  if (!(flip[pair] & 2)) out = nullptr;
  if (!fl && !out) return;
  const size_t npairs = (dsc.n + 1) / 2;
  for (size_t pr = threadIdx.x; pr < npairs; pr += 256) {
    const size_t q = map(pr);
    double2 u = reinterpret_cast<const double2*>(fl ? b1 : a1)[q], v = reinterpret_cast<const double2*>(fl ? b2 : a2)[q];
    if (fl) { reinterpret_cast<double2*>(a1)[q] = u; reinterpret_cast<double2*>(a2)[q] = v; }
    if (out) {
      const size_t i = offsets[pair] + 2 * pr;
      reinterpret_cast<double2*>(out)[i] = make_double2(u.x, v.x);
      if (2 * pr + 1 < dsc.n) reinterpret_cast<double2*>(out)[i + 1] = make_double2(u.y, v.y);
    }
  }
}

// ---- the tail of the batched d-only stage with DYNAMIC shares (sba_device.hpp: BatchDynCtl) ---------------------------------------
// Per pass three launches, enqueued without waiting: batch_depth_dyn_kernel (the launch's blocks dealt out to the pairs that
// are still iterating: share j of S runs the pair's pass over vectors j * 512 + tid + k * 512 * S and leaves its nine
// reductions in partials[(slot * S + j) * 16]), batch_depth_dyn_feed_kernel (one wave per active pair: fold the rows in share
// order, feed the pair's DepthStageSolver -- in LDS for the step --, next request or result) and the compaction of the
// active list (batch_dyn_compact_kernel).
template <typename ST>
__global__ __launch_bounds__(512, 2) void batch_depth_dyn_kernel(Planes pl, const PairDesc* __restrict__ desc,
                                                                const BatchDepthConst* __restrict__ cst, const BatchDepthPass* __restrict__ req,
                                                                double lambda, double c, double min_diagonal, double max_diagonal,
                                                                int jacobi_scaling, double* __restrict__ a1, double* __restrict__ a2,
                                                                double* __restrict__ b1, double* __restrict__ b2, double* __restrict__ sc1,
                                                                double* __restrict__ sc2, const BatchDynCtl* __restrict__ ctl,
                                                                const unsigned int* __restrict__ active, int parity, int num_pairs,
                                                                double* __restrict__ partials) {
  __shared__ double red[8][DEPTH_OUT_COUNT];
  const int tid = threadIdx.x;
  const unsigned na = ctl->nactive[parity], G = gridDim.x;
  const unsigned S = dyn_shares(na, G);
  for (unsigned item = blockIdx.x; item < na * S; item += G) {
    const unsigned slot = item / S, j = item - slot * S;
    const unsigned pair = active[static_cast<size_t>(parity) * num_pairs + slot];
    const PairDesc dsc = desc[pair];
    const BatchDepthPass ps = req[pair];
    DepthParams P;
    {
      const BatchDepthConst k = cst[pair];
#pragma unroll
      for (int i = 0; i < 9; ++i) P.R[i] = k.R[i];
#pragma unroll
      for (int i = 0; i < 3; ++i) P.t[i] = k.t[i];
    }
    P.lambda = lambda; P.c = c; P.radius = ps.radius; P.inv_radius = 1.0 / ps.radius;
    P.min_diagonal = min_diagonal; P.max_diagonal = max_diagonal; P.alpha = ps.alpha;
    P.first_iteration = ps.flags & 1; P.reuse_diagonal = (ps.flags >> 1) & 1; P.jacobi_scaling = jacobi_scaling;
    P.stream_stores = 0; P.n = dsc.n;
    const bool flip = (ps.flags >> 3) & 1;
    double r[DEPTH_OUT_COUNT];
    depth_stream<ST, 1, BatchPairMap<ST>>(pl, flip ? b1 : a1, flip ? b2 : a2, flip ? a1 : b1, flip ? a2 : b2, sc1, sc2, P,
                                          static_cast<size_t>(j) * 512 + tid, static_cast<size_t>(S) * 512, r, BatchPairMap<ST>{dsc});
    const double s = depth_block_fold<8>(r, red);
    if (tid < DEPTH_OUT_COUNT) partials[static_cast<size_t>(item) * DEPTH_ROW + tid] = s;
    __syncthreads();                                  // `red` is reused by the next item
  }
}

template <typename ST>
__global__ __launch_bounds__(64) void batch_depth_dyn_feed_kernel(const PairDesc* __restrict__ desc, int num_pairs, int parity,
                                                                 unsigned sweep_grid, const BatchDynCtl* __restrict__ ctl,
                                                                 const unsigned int* __restrict__ active,
                                                                 const double* __restrict__ partials, BatchDepthDynState* __restrict__ state,
                                                                 BatchDepthPass* __restrict__ req, const double* __restrict__ a1,
                                                                 const double* __restrict__ b1, BatchLmIo* __restrict__ io,
                                                                 int* __restrict__ done, unsigned char* __restrict__ finish) {
  __shared__ alignas(16) unsigned char mem[sizeof(BatchDepthDynState)];
  __shared__ double res_s[DEPTH_ROW];
  BatchDepthDynState* st = reinterpret_cast<BatchDepthDynState*>(mem);
  const int lane = threadIdx.x;
  const unsigned na = ctl->nactive[parity];
  const unsigned S = dyn_shares(na, sweep_grid);
  for (unsigned slot = blockIdx.x; slot < na; slot += gridDim.x) {
    const unsigned pair = active[static_cast<size_t>(parity) * num_pairs + slot];
    for (unsigned k = lane; k < sizeof(BatchDepthDynState) / 8; k += 64)
      reinterpret_cast<double*>(mem)[k] = reinterpret_cast<const double*>(state + pair)[k];
    if (lane < DEPTH_OUT_COUNT) {                     // the pair's share rows, in share order: seven sums, two maxima
      const double* rows = partials + static_cast<size_t>(slot) * S * DEPTH_ROW + lane;
      double v = rows[0];
      for (unsigned j = 1; j < S; ++j) {
        const double w = rows[static_cast<size_t>(j) * DEPTH_ROW];
        v = lane < DEPTH_OUT_SUMS ? v + w : fmax(v, w);
      }
      res_s[lane] = v;
    }
    __syncthreads();
    if (lane == 0) {
      st->solver.feed(res_s);
      if (st->solver.take_candidate()) st->flip ^= 1;
      if (st->solver.done()) {
        const PairDesc dsc = desc[pair];
        BatchLmIo res{};
        res.summary = st->solver.summary();
        res.status = st->solver.status();
        res.pad_ = st->passes;
        if (dsc.n > 0) {
          const double2 u0 = reinterpret_cast<const double2*>(st->flip ? b1 : a1)[BatchPairMap<ST>{dsc}(0)];
          res.d1 = u0.x; res.d2 = dsc.n > 1 ? u0.y : u0.x;
        }
        io[pair] = res;                               // mapped host memory; read by the host after the last publication
        finish[pair] = static_cast<unsigned char>((st->flip ? 1 : 0) | 2);
        done[pair] = 1;
      } else {
        const DepthPassRequest& rq = st->solver.request();
        req[pair] = BatchDepthPass{rq.radius, rq.alpha, 0ull,
                                   static_cast<unsigned>((rq.first ? 1 : 0) | (rq.keep_diagonal ? 2 : 0) | (st->flip ? 8 : 0)), 0u};
        st->passes += 1;
      }
    }
    __syncthreads();
    for (unsigned k = lane; k < sizeof(BatchDepthDynState) / 8; k += 64)
      reinterpret_cast<double*>(state + pair)[k] = reinterpret_cast<const double*>(mem)[k];
    __syncthreads();                                  // mem / res_s are reused by the next slot
  }
}

// [nblocks][16] -> out[9]: sums of slots 0..6 and the maxima of slots 7, 8, in a fixed order.
// With host_out (mapped pinned memory) the nine results are also published to the host: stores, system-scope release,
// then the sequence number in host_out[24] -- the host polls that word (same protocol as finalize_kernel).
__global__ __launch_bounds__(64 * DEPTH_OUT_COUNT) void depth_finalize_kernel(const double* __restrict__ partials,
                                                                              int nblocks, double* __restrict__ out,
                                                                              double* __restrict__ host_out,
                                                                              unsigned long long seq, int gather_slot) {
  // One wave per result: lane l folds rows l, l+64, ... (four independent loads in flight), then a butterfly over the
  // wave.  The rows were just written by other CUs, so the length of this kernel is load round trips, not arithmetic.
  __shared__ double res[DEPTH_ROW];
  const int slot = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const bool is_max = slot >= DEPTH_OUT_SUMS;
  double v0 = 0.0, v1 = 0.0, v2 = 0.0, v3 = 0.0;
  int b = lane;
  for (; b + 192 < nblocks; b += 256) {
    const double a0 = partials[static_cast<size_t>(b) * DEPTH_ROW + slot], a1 = partials[static_cast<size_t>(b + 64) * DEPTH_ROW + slot];
    const double a2 = partials[static_cast<size_t>(b + 128) * DEPTH_ROW + slot], a3 = partials[static_cast<size_t>(b + 192) * DEPTH_ROW + slot];
    if (is_max) { v0 = fmax(v0, a0); v1 = fmax(v1, a1); v2 = fmax(v2, a2); v3 = fmax(v3, a3); }
    else { v0 += a0; v1 += a1; v2 += a2; v3 += a3; }
  }
  for (; b < nblocks; b += 64) {
    const double a0 = partials[static_cast<size_t>(b) * DEPTH_ROW + slot];
    v0 = is_max ? fmax(v0, a0) : v0 + a0;
  }
  const double s = is_max ? wave_max(fmax(fmax(v0, v1), fmax(v2, v3))) : wave_sum((v0 + v1) + (v2 + v3));
  if (lane == 0) res[slot] = s;
  __syncthreads();
  if (gather_slot >= 0) {
    // Sharded problem: `out` is the 24-double pack a SUM all-reduce follows on.  Slots 0..6 carry the seven sums; a
    // maximum cannot be summed, so every rank deposits its own two in slots 8 + rank and 16 + rank (zeros elsewhere)
    // -- after the all-reduce every rank holds all of them and takes the maxima on the host.
    if (threadIdx.x < 24) {
      const int t = static_cast<int>(threadIdx.x);
      out[t] = t < DEPTH_OUT_SUMS ? res[t] : (t == 8 + gather_slot ? res[DEPTH_OUT_GMAX] : (t == 16 + gather_slot ? res[DEPTH_OUT_DMAX] : 0.0));
    }
    return;
  }
  if (threadIdx.x < DEPTH_OUT_COUNT) {
    out[threadIdx.x] = res[threadIdx.x];
    if (host_out) host_store(host_out + threadIdx.x, res[threadIdx.x]);
  }
  if (host_out && threadIdx.x < 64) {     // wave 0 holds all nine host stores
    host_release();
    if (threadIdx.x == 0)
      __hip_atomic_store(reinterpret_cast<unsigned long long*>(host_out + 24), seq, __ATOMIC_RELAXED,
                         __HIP_MEMORY_SCOPE_SYSTEM);
  }
}

}  // namespace

typedef void (*DepthFn)(Planes, const double*, const double*, double*, double*, double*, double*, DepthParams, double*);
int depth_ahead() {      // SBA_DEPTH_AHEAD = 1 / 2: grid-stride steps of loads in flight (A/B; profiles/r03_depth_ahead.log)
  static const int ahead = [] { const char* e = std::getenv("SBA_DEPTH_AHEAD"); return e && e[0] == '2' ? 2 : 1; }();
  return ahead;
}
DepthFn depth_pick(int store) {
  if (depth_ahead() == 2) return store == 0 ? depth_step_kernel<double, 2> : depth_step_kernel<float, 2>;
  return store == 0 ? depth_step_kernel<double, 1> : depth_step_kernel<float, 1>;
}

hipError_t depth_blocks_per_cu(int store, int* blocks) {
  return hipOccupancyMaxActiveBlocksPerMultiprocessor(blocks, reinterpret_cast<const void*>(depth_pick(store)), 256, 0);
}

hipError_t launch_resident_depth(int store, const Planes& pl, size_t n, double* a1, double* a2, double* b1, double* b2,
                                 double* sc1, double* sc2, const ResidentRecord* rec_dev, double* host_pack_dev,
                                 unsigned long long first_cmd_seq, unsigned long long first_pack_seq,
                                 unsigned long long idle_ticks, hipStream_t stream) {
  if (store == 0)
    hipLaunchKernelGGL((resident_depth_kernel<double>), dim3(1), dim3(256), 0, stream, pl, static_cast<unsigned long long>(n), a1, a2,
                       b1, b2, sc1, sc2, rec_dev, host_pack_dev, first_cmd_seq, first_pack_seq, idle_ticks);
  else
    hipLaunchKernelGGL((resident_depth_kernel<float>), dim3(1), dim3(256), 0, stream, pl, static_cast<unsigned long long>(n), a1, a2,
                       b1, b2, sc1, sc2, rec_dev, host_pack_dev, first_cmd_seq, first_pack_seq, idle_ticks);
  return hipGetLastError();
}

hipError_t launch_batch_depth_step(int store, const Planes& pl, const PairDesc* desc, const BatchDepthConst* cst,
                                   const BatchDepthPass* pass_host_dev, int num_pairs, double lambda, double c, double min_diagonal,
                                   double max_diagonal, double* a1, double* a2, double* b1, double* b2, double* sc1, double* sc2,
                                   double* out_host_dev, unsigned int* ticket, unsigned long long seq, hipStream_t stream) {
  if (num_pairs <= 0) return hipSuccess;
  if (store == 0)
    hipLaunchKernelGGL((batch_depth_step_kernel<double>), dim3(num_pairs), dim3(512), 0, stream, pl, desc, cst, pass_host_dev, lambda, c,
                       min_diagonal, max_diagonal, a1, a2, b1, b2, sc1, sc2, out_host_dev, ticket, seq);
  else
    hipLaunchKernelGGL((batch_depth_step_kernel<float>), dim3(num_pairs), dim3(512), 0, stream, pl, desc, cst, pass_host_dev, lambda, c,
                       min_diagonal, max_diagonal, a1, a2, b1, b2, sc1, sc2, out_host_dev, ticket, seq);
  return hipGetLastError();
}

hipError_t launch_batch_depth_solve(int store, const Planes& pl, const PairDesc* desc, const BatchDepthConst* cst, int num_pairs,
                                    double lambda, double c, const sba_lm_options& opt, double* a1, double* a2, double* b1, double* b2,
                                    double* sc1, double* sc2, const unsigned long long* offsets_dev, double* out_dev, BatchLmIo* io,
                                    unsigned int* ticket, unsigned long long* seq_host_dev, unsigned long long seq,
                                    hipStream_t stream, int pass_cap, void* cont_state, BatchDepthPass* cont_req, int* cont_done,
                                    unsigned char* cont_finish) {
  if (num_pairs <= 0) return hipSuccess;
  int max_trips = batch_depth_pass_bound(opt);
  if (cont_state && pass_cap > 0 && pass_cap < max_trips) max_trips = pass_cap;
  const BatchDepthCont cont{static_cast<BatchDepthDynState*>(cont_state), cont_req, cont_done, cont_finish};
  if (store == 0)
    hipLaunchKernelGGL((batch_depth_solve_kernel<double>), dim3(num_pairs), dim3(512), 0, stream, pl, desc, cst, lambda, c, opt, a1, a2,
                       b1, b2, sc1, sc2, offsets_dev, out_dev, io, ticket, seq_host_dev, seq, max_trips, cont);
  else
    hipLaunchKernelGGL((batch_depth_solve_kernel<float>), dim3(num_pairs), dim3(512), 0, stream, pl, desc, cst, lambda, c, opt, a1, a2,
                       b1, b2, sc1, sc2, offsets_dev, out_dev, io, ticket, seq_host_dev, seq, max_trips, cont);
  return hipGetLastError();
}

size_t batch_depth_dyn_state_bytes() { return sizeof(BatchDepthDynState); }

// One pass of every pair that is still iterating (sweep + feed; the caller adds the compaction launch).
hipError_t launch_batch_depth_dyn_pass(int store, const Planes& pl, const PairDesc* desc, const BatchDepthConst* cst, int num_pairs,
                                       double lambda, double c, const sba_lm_options& opt, double* a1, double* a2, double* b1, double* b2,
                                       double* sc1, double* sc2, int parity, int sweep_grid, void* state, BatchDepthPass* req,
                                       const BatchDynCtl* ctl, const unsigned int* active, int* done, unsigned char* finish,
                                       double* partials, BatchLmIo* io, hipStream_t stream) {
  if (num_pairs <= 0) return hipSuccess;
  if (sweep_grid < 1) return hipErrorInvalidValue;
  const unsigned feed_grid = static_cast<unsigned>(num_pairs < 1024 ? num_pairs : 1024);
  if (store == 0) {
    hipLaunchKernelGGL((batch_depth_dyn_kernel<double>), dim3(static_cast<unsigned>(sweep_grid)), dim3(512), 0, stream, pl, desc, cst, req, lambda,
                       c, opt.min_lm_diagonal, opt.max_lm_diagonal, opt.jacobi_scaling ? 1 : 0, a1, a2, b1, b2, sc1, sc2, ctl, active, parity,
                       num_pairs, partials);
    hipLaunchKernelGGL((batch_depth_dyn_feed_kernel<double>), dim3(feed_grid), dim3(64), 0, stream, desc, num_pairs, parity,
                       static_cast<unsigned>(sweep_grid), ctl, active, partials, static_cast<BatchDepthDynState*>(state), req, a1, b1, io, done,
                       finish);
  } else {
    hipLaunchKernelGGL((batch_depth_dyn_kernel<float>), dim3(static_cast<unsigned>(sweep_grid)), dim3(512), 0, stream, pl, desc, cst, req, lambda,
                       c, opt.min_lm_diagonal, opt.max_lm_diagonal, opt.jacobi_scaling ? 1 : 0, a1, a2, b1, b2, sc1, sc2, ctl, active, parity,
                       num_pairs, partials);
    hipLaunchKernelGGL((batch_depth_dyn_feed_kernel<float>), dim3(feed_grid), dim3(64), 0, stream, desc, num_pairs, parity,
                       static_cast<unsigned>(sweep_grid), ctl, active, partials, static_cast<BatchDepthDynState*>(state), req, a1, b1, io, done,
                       finish);
  }
  return hipGetLastError();
}

hipError_t launch_batch_depth_finish(int store, const PairDesc* desc, const unsigned char* flip_dev, int num_pairs, double* a1,
                                     double* a2, const double* b1, const double* b2, const unsigned long long* offsets_dev,
                                     double* out_dev, hipStream_t stream) {
  if (num_pairs <= 0) return hipSuccess;
  if (store == 0)
    hipLaunchKernelGGL((batch_depth_finish_kernel<double>), dim3(num_pairs), dim3(256), 0, stream, desc, flip_dev, a1, a2, b1, b2,
                       offsets_dev, out_dev);
  else
    hipLaunchKernelGGL((batch_depth_finish_kernel<float>), dim3(num_pairs), dim3(256), 0, stream, desc, flip_dev, a1, a2, b1, b2,
                       offsets_dev, out_dev);
  return hipGetLastError();
}

hipError_t launch_depth_step(int store, const Planes& pl, const double* d1, const double* d2, double* c1,
                             double* c2, double* sc1, double* sc2,
                             const DepthParams& prm, double* partials, int grid, double* out, double* host_out,
                             unsigned long long seq, int gather_slot, hipStream_t stream) {
  if (grid > 0) {
    hipLaunchKernelGGL(depth_pick(store), dim3(grid), dim3(256), 0, stream, pl, d1, d2, c1, c2, sc1, sc2, prm, partials);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
  }
  hipLaunchKernelGGL(depth_finalize_kernel, dim3(1), dim3(64 * DEPTH_OUT_COUNT), 0, stream, partials, grid, out,
                     host_out, seq, gather_slot);
  return hipGetLastError();
}

}  // namespace sba
