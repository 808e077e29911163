// Batched kernels (BASELINE config C5): many independent two-view problems ("pairs") per launch -- the batched sweep,
// the per-pair preparation of the sweep state, and the fold + moment conversion + publication of all packs.
// Device core shared with the single-problem sweep: sba_sweep_core.hpp.
#include <new>

#include "sba_lm.hpp"
#include "sba_resident.hpp"
#include "sba_sweep_core.hpp"

namespace sba {
namespace {

// ---- batched sweep: many independent two-view problems ("pairs") in ONE launch (BASELINE config C5) ------------
// Block group g = blockIdx.x / bpp works on pair g with that pair's own R|t (params[g], wave-uniform address ->
// scalar loads), blocks j = blockIdx.x % bpp of the group grid-stride over the pair's vectors.  Rows of block
// partials are folded per pair by batch_finalize_kernel.  A pair with n == 0 (e.g. already converged) costs its
// blocks only the row store.
template <int MODE, int DEPTH, typename ST, int KIND, bool LOSS>
__device__ __forceinline__ void sweep_share_rows(const Planes& pl, const SweepParams* __restrict__ P, const PairDesc& dsc, unsigned j,
                                                 unsigned bpp, double* __restrict__ wave_out, double* __restrict__ row) {
  constexpr int NACC = AccMap<MODE, KIND>::N;
  constexpr int PPT = Lanes<ST>::PPT;
  const int tid = threadIdx.x;
  const size_t n = P->n;
  const size_t stride = static_cast<size_t>(bpp) * kBlock;
  static_assert(kBlock == kPairTile, "a block sweeps one 256-vector tile per trip");

  double acc[NACC];
#pragma unroll
  for (int k = 0; k < NACC; ++k) acc[k] = 0.0;
  const size_t nfull = n / PPT;
  // logical vector p = (j + k bpp) * 256 + tid lives at plane vector q = first + (j + k bpp) * tile_stride + tid
  size_t p = static_cast<size_t>(j) * kBlock + tid;
  size_t q = dsc.first_vec + static_cast<size_t>(j) * dsc.tile_stride + tid;
  const size_t qstride = static_cast<size_t>(bpp) * dsc.tile_stride;
  VecRegs<ST, DEPTH> cur, nxt;
  if (p < nfull) cur.load(pl, q);
  while (p < nfull) {
    const size_t pn = p + stride;
    q += qstride;
    if (pn < nfull) nxt.load(pl, q);
    consume<MODE, DEPTH, ST, KIND, LOSS, false>(cur, P, p, n, acc);
    cur = nxt;
    p = pn;
  }
  if (nfull * PPT != n && j == bpp - 1 && tid == kBlock - 1) {
    cur.load(pl, pair_vector(dsc, nfull));
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
    row[tid] = s;
  }
}

template <int MODE, int DEPTH, typename ST, int KIND, bool LOSS>
__global__ __launch_bounds__(kBlock) void batch_sweep_kernel(Planes pl, const SweepParams* __restrict__ params,
                                                            const PairDesc* __restrict__ desc, int bpp,
                                                            double* __restrict__ partials) {
  __shared__ double lds[(kBlock / 64) * 24];
  const unsigned pair = blockIdx.x / static_cast<unsigned>(bpp), j = blockIdx.x % static_cast<unsigned>(bpp);
  sweep_share_rows<MODE, DEPTH, ST, KIND, LOSS>(pl, params + pair, desc[pair], j, static_cast<unsigned>(bpp), lds,
                                                partials + static_cast<size_t>(blockIdx.x) * kRow);
}

// ---- DYNAMIC shares (sba_batch_solve, device-resident): the blocks of a launch are dealt out to the pairs that are still
// iterating.  ctl->nactive[parity] pairs are listed in active[parity * num_pairs ...]; with G = gridDim.x blocks every active
// pair gets S = G / nactive shares (1 when there are more pairs than blocks: a block then takes several pairs in turn), share
// j of S sweeps the pair's tiles j, j + S, ... exactly like block j of bpp = S above, and leaves its row at
// partials[(slot * S + j) * kRow].  Pairs that have converged cost nothing any more and their CUs work for the others.
template <int MODE, int DEPTH, typename ST, int KIND, bool LOSS>
__global__ __launch_bounds__(kBlock) void batch_sweep_dyn_kernel(Planes pl, const SweepParams* __restrict__ params,
                                                                const PairDesc* __restrict__ desc,
                                                                const BatchDynCtl* __restrict__ ctl,
                                                                const unsigned int* __restrict__ active, int parity, int num_pairs,
                                                                double* __restrict__ partials) {
  __shared__ double lds[(kBlock / 64) * 24];
  const unsigned na = ctl->nactive[parity], G = gridDim.x;
  const unsigned S = dyn_shares(na, G);
  for (unsigned item = blockIdx.x; item < na * S; item += G) {      // na comes from memory every block reads alike: block-uniform
    const unsigned slot = item / S, j = item - slot * S;
    const unsigned pair = active[static_cast<size_t>(parity) * num_pairs + slot];
    sweep_share_rows<MODE, DEPTH, ST, KIND, LOSS>(pl, params + pair, desc[pair], j, S, lds, partials + static_cast<size_t>(item) * kRow);
    __syncthreads();                                                // the LDS scratch is reused by the next item
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
  // `state` lives in mapped HOST memory.  Each lane reads its own 80-byte record with independent loads (one PCIe
  // round trip); staging the records through LDS with coalesced loads measured slower (11.8 vs 8.6 us for 256 pairs).
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
  // 256 pairs per pass: thread t < 256 folds pair t's bpp rows in block order (all twelve 16-byte loads of a row in
  // flight together -- the rows were just written by other CUs, a load is a full round trip), converts, and leaves the
  // pack in LDS; then all 1024 threads write the 256 x 24 doubles out with coalesced stores (512 B per wave
  // instruction to the mapped host buffer instead of one 8-byte PCIe write per lane).
  __shared__ double out_lds[256 * 24];
  for (int base = 0; base < num_pairs; base += 256) {
    const int count = min(256, num_pairs - base);
    if (static_cast<int>(threadIdx.x) < count) {
      const int pair = base + threadIdx.x;
      double raw[24], out[24];
#pragma unroll
      for (int k = 0; k < 24; ++k) raw[k] = 0.0;
      const double2* rows = reinterpret_cast<const double2*>(partials + static_cast<size_t>(pair) * bpp * kRow);
      for (int b = 0; b < bpp; ++b) {
        double2 v[12];
#pragma unroll
        for (int k = 0; k < 12; ++k) v[k] = rows[static_cast<size_t>(b) * (kRow / 2) + k];
#pragma unroll
        for (int k = 0; k < 12; ++k) { raw[2 * k] += v[k].x; raw[2 * k + 1] += v[k].y; }
      }
      moments_to_normal_pack(true, convert == 2, frames + 18 * pair, frames + 18 * pair + 9, raw, out);
#pragma unroll
      for (int k = 0; k < 24; ++k) out_lds[threadIdx.x * 24 + k] = out[k];
    }
    __syncthreads();
    for (int w = threadIdx.x; w < count * 24; w += 1024) {
      const double v = out_lds[w];
      packs[base * 24 + w] = v;
      if (packs_host) packs_host[base * 24 + w] = v;
    }
    __syncthreads();
  }
  if (!packs_host) return;
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0)
    __hip_atomic_store(reinterpret_cast<unsigned long long*>(packs_host + items), seq, __ATOMIC_RELAXED,
                       __HIP_MEMORY_SCOPE_SYSTEM);
}
// ---- one block = one pair: sweep the pair's matches with `prm` held in registers and fold the block's 24 sums --------
// (fold order identical to batch_sweep_kernel + batch_convert_finalize_kernel with bpp = 1).  Result in raw_s[24]
// (LDS), valid after the function returns on every thread (it ends with a barrier).
template <int MODE, int DEPTH, typename ST, int KIND, bool LOSS>
__device__ __forceinline__ void block_sweep_fold(const Planes& pl, const PairDesc& dsc, size_t n, const SweepParams& prm,
                                                 double* __restrict__ wave_out, double* __restrict__ raw_s) {
  constexpr int NACC = AccMap<MODE, KIND>::N;
  constexpr int PPT = Lanes<ST>::PPT;
  const int tid = threadIdx.x;
  const size_t nfull = n / PPT;
  double acc[NACC];
#pragma unroll
  for (int k = 0; k < NACC; ++k) acc[k] = 0.0;
  size_t p = tid, q = dsc.first_vec + tid;          // logical vector p = 256 k + tid sits at plane vector first + k tile_stride + tid
  VecRegs<ST, DEPTH> cur, nxt;
  if (p < nfull) cur.load(pl, q);
  while (p < nfull) {
    const size_t pn = p + kBlock;
    q += dsc.tile_stride;
    if (pn < nfull) nxt.load(pl, q);
    consume<MODE, DEPTH, ST, KIND, LOSS, false>(cur, &prm, p, n, acc);
    cur = nxt;
    p = pn;
  }
  if (nfull * PPT != n && tid == kBlock - 1) {
    cur.load(pl, pair_vector(dsc, nfull));
    consume<MODE, DEPTH, ST, KIND, LOSS, true>(cur, &prm, nfull, n, acc);
  }
  const int lane = tid & 63, wave = tid >> 6;
  if (tid < 24) raw_s[tid] = 0.0;       // slots no accumulator maps to stay zero
#pragma unroll
  for (int k = 0; k < NACC; ++k) {
    const double sw = wave_sum_to_lane63(acc[k]);
    if (lane == 63) wave_out[wave * 24 + AccMap<MODE, KIND>::slot(k)] = sw;
  }
  __syncthreads();
  if (tid < NACC) {
    const int slot = AccMap<MODE, KIND>::slot(tid);
    double sum = wave_out[slot];
#pragma unroll
    for (int wv = 1; wv < kBlock / 64; ++wv) sum += wave_out[wv * 24 + slot];
    raw_s[slot] = sum;
  }
  __syncthreads();
}

// ---- one batched evaluation step in ONE launch (bpp == 1) -----------------------------------------------------------
// What batch_prepare_kernel + batch_sweep_kernel + batch_convert_finalize_kernel do in three launches: thread 0 of block
// g reads pair g's 80-byte state from mapped host memory and builds the sweep state in LDS, the block sweeps and folds,
// thread 0 maps the moments to the SBA_PACK_* layout and stores the pack to device and mapped host memory; the block
// that delivers the last pack stores the sequence word the host polls.
template <int MODE, int DEPTH, typename ST, int KIND, bool LOSS>
__global__ __launch_bounds__(kBlock) void batch_step_kernel(Planes pl, const PairDesc* __restrict__ desc,
                                                           const BatchState* __restrict__ state, double huber_delta,
                                                           double* __restrict__ packs, double* __restrict__ packs_host,
                                                           unsigned int* __restrict__ ticket, unsigned long long seq) {
  __shared__ double wave_out[(kBlock / 64) * 24];
  __shared__ double raw_s[24];
  __shared__ SweepParams prm_s;
  __shared__ double frame_s[18];
  const int tid = threadIdx.x;
  const unsigned pair = blockIdx.x;
#ifdef SBA_STEP_PROFILE
  long long tk[6] = {0, 0, 0, 0, 0, 0};
  tk[0] = wall_clock64();
#endif
  const PairDesc dsc = desc[pair];
  const size_t n_pair = dsc.n;
  if (tid == 0) {
    const BatchState st = state[pair];
#ifdef SBA_STEP_PROFILE
    prm_s.n = st.n; __builtin_amdgcn_s_waitcnt(0); tk[1] = wall_clock64();
#endif
    fill_sweep_params(st.n, DEPTH, st.rot, st.tran, st.d1, st.d2, huber_delta, &prm_s, KIND == KIND_EXPLICIT);
    if (KIND == KIND_FACTORED && MODE != MODE_TRAN) factored_frame(st.rot, frame_s, frame_s + 9);
#ifdef SBA_STEP_PROFILE
    tk[2] = wall_clock64();
#endif
  }
  __syncthreads();
  const SweepParams prm = prm_s;
  // never read past the pair's own vectors, whatever the host wrote into the state record
  const size_t n = prm.n < n_pair ? prm.n : n_pair;
  block_sweep_fold<MODE, DEPTH, ST, KIND, LOSS>(pl, dsc, n, prm, wave_out, raw_s);
#ifdef SBA_STEP_PROFILE
  tk[3] = wall_clock64();
#endif
  if (KIND == KIND_FACTORED && MODE != MODE_TRAN) {
    if (tid == 0) {
      double pack[24];
      moments_to_normal_pack(true, MODE == MODE_RT, frame_s, frame_s + 9, raw_s, pack);
#pragma unroll
      for (int k = 0; k < 24; ++k) raw_s[k] = pack[k];
    }
    __syncthreads();
  }
#ifdef SBA_STEP_PROFILE
  tk[4] = wall_clock64();
  if (tid == 0) {   // ticks of the 100 MHz clock: state read / state build / sweep + fold / conversion, in pack slots 0..3
    raw_s[0] = static_cast<double>(tk[1] - tk[0]); raw_s[1] = static_cast<double>(tk[2] - tk[1]);
    raw_s[2] = static_cast<double>(tk[3] - tk[2]); raw_s[3] = static_cast<double>(tk[4] - tk[3]);
    raw_s[4] = static_cast<double>(tk[0] % 100000000);
  }
  __syncthreads();
#endif
  if (tid >= 64) return;                 // wave 0 finishes alone
  if (tid < 24) {
    const double v = raw_s[tid];
    packs[static_cast<size_t>(pair) * 24 + tid] = v;
    if (packs_host) host_store(packs_host + static_cast<size_t>(pair) * 24 + tid, v);
  }
  if (!packs_host) return;
  host_release();
  if (tid == 0 && __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT) == gridDim.x - 1) {
    __hip_atomic_store(ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#if !SBA_PUBLISH_WRITE_THROUGH
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");
#endif
    __hip_atomic_store(reinterpret_cast<unsigned long long*>(packs_host + static_cast<size_t>(gridDim.x) * 24), seq,
                       __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}

// Per-pair solver state that outlives a launch (device memory): the one-launch kernel below hands a pair that has not
// converged within its first sweeps over to the launches with dynamic shares further down.
struct BatchLmDynState {
  LmSolver solver;
  double depth[2];
};
static_assert(sizeof(LmSolver) % 8 == 0 && sizeof(BatchLmDynState) % 8 == 0, "copied as 8-byte words");
struct BatchLmCont {          // where batch_lm_kernel leaves the pairs it does not finish (all null: it runs every pair to the end)
  BatchLmDynState* state;
  SweepParams* params;
  double* frames;
  int* done;
};

__device__ __forceinline__ void dyn_next_sweep(int mode, int depth_mode, int kind, unsigned long long n, const LmSolver* solver,
                                               const double depth[2], double huber_delta, SweepParams* __restrict__ params,
                                               double* __restrict__ frames) {
  SweepParams prm;
  fill_sweep_params(n, depth_mode, solver->query_rot(), solver->query_tran(), depth[0], depth[1], huber_delta, &prm, kind == KIND_EXPLICIT);
  *params = prm;
  if (kind == KIND_FACTORED && mode != MODE_TRAN) factored_frame(solver->query_rot(), frames, frames + 9);
}

// ---- the whole per-pair solve in ONE launch ---------------------------------------------------------------------
// With one block per pair (bpp == 1: at least one pair per CU, config C5) a pair's sweep is reduced completely inside
// its block, so nothing about its Levenberg-Marquardt iteration needs another block -- or the host.  Block g runs pair
// g's entire solve: thread 0 owns an LmSolver (the same __host__ __device__ source the host path runs, csrc/sba_lm.hpp)
// in LDS; per iteration it turns the solver's query point into the sweep state (fill_sweep_params, factored_frame),
// all 256 threads sweep the pair's matches with that state held in registers, the block folds its 24 sums, thread 0
// maps the moments to normal equations (moments_to_normal_pack) and feeds the solver.  No host round trip, no
// prepare / convert launches, no lock-step between pairs: a block leaves when its pair has converged.  The loop is
// bounded by the solver itself (max_num_iterations + 1 evaluations, every code path of LmSolver::feed either advances
// the iteration counter or terminates).
template <int MODE, int DEPTH, typename ST, int KIND, bool LOSS>
__global__ __launch_bounds__(kBlock) void batch_lm_kernel(Planes pl, const PairDesc* __restrict__ desc,
                                                         BatchLmIo* __restrict__ io, sba_lm_options opt,
                                                         unsigned int* __restrict__ ticket,
                                                         unsigned long long* __restrict__ seq_host,
                                                         unsigned long long seq, int sweep_cap, BatchLmCont cont) {
  __shared__ double wave_out[(kBlock / 64) * 24];
  __shared__ double raw_s[24];
  __shared__ SweepParams prm_s;
  __shared__ double frame_s[18];
  __shared__ int done_s;
  __shared__ alignas(16) unsigned char solver_mem[sizeof(LmSolver)];
  LmSolver* solver = reinterpret_cast<LmSolver*>(solver_mem);
  const int tid = threadIdx.x;
  const unsigned pair = blockIdx.x;
  const PairDesc dsc = desc[pair];
  const size_t n = dsc.n;
  __shared__ double depth_s[2];
  if (tid == 0) {
    // io lives in mapped HOST memory: one read of the pair's record here, one write of its result at the end
    const BatchLmIo in = io[pair];
    depth_s[0] = in.d1; depth_s[1] = in.d2;
    new (solver) LmSolver();
    solver->start(MODE, in.rot, in.tran, opt);
  }
#ifdef SBA_LM_PROFILE
  long long tk_prep = 0, tk_sweep = 0, tk_conv = 0, tk_feed = 0, tk0 = 0;
#define SBA_TICK(acc) do { const long long now_ = wall_clock64(); acc += now_ - tk0; tk0 = now_; } while (0)
#else
#define SBA_TICK(acc) do { } while (0)
#endif
  // The loop is block-uniform BY CONSTRUCTION, not by what one compiler happens to emit:
  //   * the trip count is bounded by a counter every thread keeps (`trip`), not by state only thread 0 sees;
  //   * the ONE thread-0 region of a trip sits between two barriers of that same trip (B0 ... B1) -- there is no
  //     thread-0-only code between the sweep's last barrier and the back edge, nor between the back edge and B0, so no
  //     divergent region is adjacent to the back edge for a jump-threading pass to splice into a private inner loop
  //     (that is what once left the other 255 threads at a barrier without lane 0: the kernel never finished);
  //   * every thread takes the exit decision from the LDS word `done_s`, read after B1.
  // tests/test_isa_checks_cpu.py remains as the second line: all four barriers at loop depth 1 in every instantiation,
  // with the Makefile's flags and with the profiling variants.  The solver needs at most max_num_iterations + 1 sweeps.
  // sweep_cap > 0 (with `cont`): at most that many sweeps in this launch; a pair that needs more is handed over below.
  const int full_trips = opt.max_num_iterations + 8;
  const int max_trips = sweep_cap > 0 && sweep_cap < full_trips ? sweep_cap : full_trips;
  for (int trip = 0; trip <= max_trips; ++trip) {    // the trip after the last sweep only feeds its sums (and breaks below)
    __syncthreads();                        // B0: the previous trip's sums (raw_s) are complete
    if (tid == 0) {
#ifdef SBA_LM_PROFILE
      if (trip > 0) SBA_TICK(tk_sweep); else tk0 = wall_clock64();   // profiling ticks live in THIS region too: none at the back edge
#endif
      if (trip > 0) {
        double pack[24];
        if (KIND == KIND_FACTORED && MODE != MODE_TRAN)
          moments_to_normal_pack(true, MODE == MODE_RT, frame_s, frame_s + 9, raw_s, pack);
        else
          for (int k = 0; k < 24; ++k) pack[k] = raw_s[k];
        sba_normal_eq ne;
        expand_pack(MODE, pack, &ne);
        SBA_TICK(tk_conv);
        solver->feed(ne);
        SBA_TICK(tk_feed);
      }
      done_s = solver->done() ? 1 : 0;
      if (!done_s) {
        fill_sweep_params(n, DEPTH, solver->query_rot(), solver->query_tran(), depth_s[0], depth_s[1], opt.huber_delta, &prm_s,
                          KIND == KIND_EXPLICIT);
        if (KIND == KIND_FACTORED && MODE != MODE_TRAN) factored_frame(solver->query_rot(), frame_s, frame_s + 9);
      }
      SBA_TICK(tk_prep);
    }
    __syncthreads();                        // B1: done_s / the next sweep state are visible to the block
    if (done_s || trip == max_trips) break;      // both block-uniform: an LDS word read after B1, the trip counter
    const SweepParams prm = prm_s;        // LDS broadcast -> registers, held across the sweep
    block_sweep_fold<MODE, DEPTH, ST, KIND, LOSS>(pl, dsc, n, prm, wave_out, raw_s);   // two barriers, the last one at its end
  }
  if (tid == 0) {
    // The sweep cap ended the loop: the last trip fed the last sweep's sums and, the pair being not done, left the state of its
    // next sweep in prm_s / frame_s.  Solver, depths and that state go to device memory for the launches with dynamic shares
    // (batch_sweep_dyn_kernel / batch_lm_dyn_feed_kernel).
    bool handed_over = false;
    if (!done_s && cont.state) {
      BatchLmDynState* out = cont.state + pair;
      for (unsigned k = 0; k < sizeof(LmSolver) / 8; ++k) reinterpret_cast<double*>(&out->solver)[k] = reinterpret_cast<const double*>(solver_mem)[k];
      out->depth[0] = depth_s[0]; out->depth[1] = depth_s[1];
      cont.params[pair] = prm_s;
      if (KIND == KIND_FACTORED && MODE != MODE_TRAN)
        for (int k = 0; k < 18; ++k) cont.frames[18 * pair + k] = frame_s[k];
      handed_over = true;
    }
    if (cont.done) cont.done[pair] = handed_over ? 0 : 1;
    BatchLmIo res;
    for (int a = 0; a < 3; ++a) { res.rot[a] = solver->rot()[a]; res.tran[a] = solver->tran()[a]; }
    res.d1 = depth_s[0]; res.d2 = depth_s[1];
    res.summary = solver->summary();
#ifdef SBA_LM_PROFILE
    // ticks of the 100 MHz wall clock spent by thread 0 in: preparation / waiting for the sweep / conversion / LM feed
    res.summary.initial_cost = static_cast<double>(tk_prep); res.summary.final_cost = static_cast<double>(tk_sweep);
    res.summary.final_gradient_max_norm = static_cast<double>(tk_conv); res.summary.final_radius = static_cast<double>(tk_feed);
#endif
    res.status = solver->done() ? solver->status() : SBA_ERR_NUMERIC;   // max_trips ran out: cannot happen, but never silent
    res.pad_ = handed_over ? 1 : 0;                                       // 1: the pair continues in the dynamic launches
    io[pair] = res;
    // Completion: this block's record is in host memory (system-scope release + vmcnt(0)) before it takes a ticket; the
    // block that takes the last ticket re-arms the counter and stores the sequence word the host polls -- no stream
    // synchronise (that alone cost ~50 us of a 0.6 ms solve).
    if (seq_host) {
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

// ---- resident evaluator of ONE small problem (sba_resident.hpp) ---------------------------------------------------------
// One block stays resident for a whole solve stage: per trip wave 0 publishes the previous sweep's sums to the host and
// waits (bounded in time) for the next command in the mapped record; the block then sweeps all matches with the
// commanded state -- the same block_sweep_fold the one-launch batch kernels use, same fold order -- and loops.  The loop
// has the shape of batch_lm_kernel's: the one divergent (wave-0) region of a trip sits between two barriers of that trip,
// the trip count is bounded by a counter every thread keeps, the exit decision is read from LDS after a barrier.
template <int MODE, int DEPTH, typename ST, int KIND, bool LOSS>
__global__ __launch_bounds__(kBlock) void resident_sweep_kernel(Planes pl, unsigned long long n_resident,
                                                               const ResidentRecord* __restrict__ rec,
                                                               double* __restrict__ host_pack,
                                                               unsigned long long first_cmd_seq,
                                                               unsigned long long first_pack_seq,
                                                               unsigned long long idle_ticks) {
  __shared__ double wave_out[(kBlock / 64) * 24];
  __shared__ double raw_s[24];
  __shared__ double cmd_s[kResidentPayload];
  __shared__ int end_s;
  const int tid = threadIdx.x;
  if (tid == 0) end_s = 0;
  int trip = 0;
  for (; trip < kResidentMaxTrips; ++trip) {
    __syncthreads();                                   // B0: the previous trip's sums are complete
    if (tid < 64) {                                    // wave 0: answer, then wait for the next command
      if (trip > 0) resident_publish(host_pack, raw_s, 24, first_pack_seq + static_cast<unsigned long long>(trip - 1));
      int end = 0;
      resident_wait_command(rec, first_cmd_seq + static_cast<unsigned long long>(trip), idle_ticks, cmd_s, &end);
      if (tid == 0 && end) end_s = end;
    }
    __syncthreads();                                   // B1: command (or the end reason) visible to the block
    if (end_s) break;
    const int op = static_cast<int>(cmd_s[0]);
    if (op != RESIDENT_OP_SWEEP) {
      if (tid == 0) end_s = op == RESIDENT_OP_QUIT ? RESIDENT_END_QUIT : RESIDENT_END_BAD_OP;
      break;
    }
    SweepParams prm;
#pragma unroll
    for (int k = 0; k < 42; ++k) reinterpret_cast<double*>(&prm)[k] = cmd_s[1 + k];
    unsigned long long n_cmd;
    const double nd = cmd_s[43];
    __builtin_memcpy(&n_cmd, &nd, sizeof(n_cmd));
    const size_t n = n_cmd < n_resident ? n_cmd : n_resident;      // never past the resident matches, whatever the record says
    prm.n = n;
    block_sweep_fold<MODE, DEPTH, ST, KIND, LOSS>(pl, PairDesc{0ull, n, kPairTile, 0ull}, n, prm, wave_out, raw_s);
  }
  __syncthreads();
  if (tid < 64) {
    // trip budget used up: the last trip's answer has not been published yet (answers go out at the top of the next trip)
    if (trip == kResidentMaxTrips) resident_publish(host_pack, raw_s, 24, first_pack_seq + static_cast<unsigned long long>(trip - 1));
    resident_publish_end(host_pack, end_s ? end_s : RESIDENT_END_TRIPS);
  }
}

// ---- device-resident per-pair LM with dynamic shares: the kernels around batch_sweep_dyn_kernel ------------------------------
// Per LM iteration three launches, all enqueued by the host without waiting: the sweep above, batch_lm_dyn_feed_kernel (one
// wave per active pair: fold the pair's share rows in share order, feed the pair's LmSolver -- copied into LDS for the step
// and back --, write the next sweep state or the result) and batch_dyn_compact_kernel (one block: the pairs that are not done
// yet, in order, become the other parity's active list; publishes their number to the host when asked).  Kernel boundaries
// are the only synchronisation: no block ever waits for another.
__global__ __launch_bounds__(64) void batch_lm_dyn_init_kernel(int mode, int depth_mode, int kind, const PairDesc* __restrict__ desc,
                                                               const BatchLmIo* __restrict__ io, sba_lm_options opt, int num_pairs,
                                                               BatchLmDynState* __restrict__ state, SweepParams* __restrict__ params,
                                                               double* __restrict__ frames, BatchDynCtl* __restrict__ ctl,
                                                               unsigned int* __restrict__ active, int* __restrict__ done) {
  __shared__ alignas(16) unsigned char mem[sizeof(BatchLmDynState)];
  BatchLmDynState* st = reinterpret_cast<BatchLmDynState*>(mem);
  const unsigned pair = blockIdx.x;
  const int lane = threadIdx.x;
  if (lane == 0) {
    const BatchLmIo in = io[pair];           // mapped host memory: the pair's start point
    new (&st->solver) LmSolver();
    st->solver.start(mode, in.rot, in.tran, opt);
    st->depth[0] = in.d1; st->depth[1] = in.d2;
    dyn_next_sweep(mode, depth_mode, kind, desc[pair].n, &st->solver, st->depth, opt.huber_delta, params + pair, frames + 18 * pair);
    done[pair] = 0;
    active[pair] = pair;
    if (pair == 0) { ctl->nactive[0] = static_cast<unsigned>(num_pairs); ctl->nactive[1] = 0u; }
  }
  __syncthreads();
  for (unsigned k = lane; k < sizeof(BatchLmDynState) / 8; k += 64)
    reinterpret_cast<double*>(state + pair)[k] = reinterpret_cast<const double*>(mem)[k];
}

__global__ __launch_bounds__(64) void batch_lm_dyn_feed_kernel(int mode, int depth_mode, int kind, const PairDesc* __restrict__ desc,
                                                               sba_lm_options opt, int num_pairs, int parity, unsigned sweep_grid,
                                                               const BatchDynCtl* __restrict__ ctl,
                                                               const unsigned int* __restrict__ active,
                                                               const double* __restrict__ partials,
                                                               BatchLmDynState* __restrict__ state, SweepParams* __restrict__ params,
                                                               double* __restrict__ frames, BatchLmIo* __restrict__ io,
                                                               int* __restrict__ done) {
  __shared__ alignas(16) unsigned char mem[sizeof(BatchLmDynState)];
  __shared__ double raw_s[24];
  BatchLmDynState* st = reinterpret_cast<BatchLmDynState*>(mem);
  const int lane = threadIdx.x;
  const unsigned na = ctl->nactive[parity];
  const unsigned S = dyn_shares(na, sweep_grid);
  for (unsigned slot = blockIdx.x; slot < na; slot += gridDim.x) {
    const unsigned pair = active[static_cast<size_t>(parity) * num_pairs + slot];
    for (unsigned k = lane; k < sizeof(BatchLmDynState) / 8; k += 64)
      reinterpret_cast<double*>(mem)[k] = reinterpret_cast<const double*>(state + pair)[k];
    if (lane < 24) {                                  // the pair's share rows, in share order
      const double* rows = partials + static_cast<size_t>(slot) * S * kRow + lane;
      double sum = rows[0];
      for (unsigned j = 1; j < S; ++j) sum += rows[static_cast<size_t>(j) * kRow];
      raw_s[lane] = sum;
    }
    __syncthreads();
    if (lane == 0) {
      double pack[24];
      if (kind == KIND_FACTORED && mode != MODE_TRAN)
        moments_to_normal_pack(true, mode == MODE_RT, frames + 18 * pair, frames + 18 * pair + 9, raw_s, pack);
      else
        for (int k = 0; k < 24; ++k) pack[k] = raw_s[k];
      sba_normal_eq ne;
      expand_pack(mode, pack, &ne);
      st->solver.feed(ne);
      if (st->solver.done()) {
        BatchLmIo res;
        for (int a = 0; a < 3; ++a) { res.rot[a] = st->solver.rot()[a]; res.tran[a] = st->solver.tran()[a]; }
        res.d1 = st->depth[0]; res.d2 = st->depth[1];
        res.summary = st->solver.summary();
        res.status = st->solver.status();
        res.pad_ = 0;
        io[pair] = res;                               // mapped host memory; the host reads it after the last publication
        done[pair] = 1;
      } else {
        dyn_next_sweep(mode, depth_mode, kind, desc[pair].n, &st->solver, st->depth, opt.huber_delta, params + pair, frames + 18 * pair);
      }
    }
    __syncthreads();
    for (unsigned k = lane; k < sizeof(LmSolver) / 8; k += 64)
      reinterpret_cast<double*>(state + pair)[k] = reinterpret_cast<const double*>(mem)[k];
    __syncthreads();                                  // mem / raw_s are reused by the next slot
  }
}

__global__ __launch_bounds__(256) void batch_dyn_identity_kernel(BatchDynCtl* __restrict__ ctl, unsigned int* __restrict__ active,
                                                                 int num_pairs) {
  const unsigned i = blockIdx.x * 256u + threadIdx.x;
  if (i < static_cast<unsigned>(num_pairs)) active[i] = i;
  if (i == 0) { ctl->nactive[0] = static_cast<unsigned>(num_pairs); ctl->nactive[1] = 0u; }
}

// One block: active[parity] minus the pairs that are done -> active[parity ^ 1], order kept.  host_words (may be null):
// mapped host memory, [0] <- pairs still active, then [1] <- seq (the host polls [1]).
__global__ __launch_bounds__(256) void batch_dyn_compact_kernel(BatchDynCtl* __restrict__ ctl, unsigned int* __restrict__ active,
                                                                const int* __restrict__ done, int num_pairs, int parity,
                                                                unsigned long long* __restrict__ host_words,
                                                                unsigned long long seq) {
  __shared__ unsigned int count_s[256];
  __shared__ unsigned int total_s;
  const unsigned tid = threadIdx.x;
  const unsigned na = ctl->nactive[parity];
  const unsigned per = (na + 255u) / 256u;
  const unsigned lo = min(na, tid * per), hi = min(na, lo + per);
  const unsigned int* src = active + static_cast<size_t>(parity) * num_pairs;
  unsigned int* dst = active + static_cast<size_t>(parity ^ 1) * num_pairs;
  unsigned cnt = 0;
  for (unsigned i = lo; i < hi; ++i) cnt += done[src[i]] ? 0u : 1u;
  count_s[tid] = cnt;
  __syncthreads();
  if (tid == 0) {
    unsigned run = 0;
    for (int t = 0; t < 256; ++t) { const unsigned c = count_s[t]; count_s[t] = run; run += c; }
    total_s = run;
  }
  __syncthreads();
  unsigned o = count_s[tid];
  for (unsigned i = lo; i < hi; ++i) {
    const unsigned pair = src[i];
    if (!done[pair]) dst[o++] = pair;
  }
  if (tid == 0) {
    ctl->nactive[parity ^ 1] = total_s;
    if (host_words) {
      __hip_atomic_store(host_words, static_cast<unsigned long long>(total_s), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __hip_atomic_store(host_words + 1, seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
  }
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

typedef void (*BatchLmFn)(Planes, const PairDesc*, BatchLmIo*, sba_lm_options, unsigned int*, unsigned long long*,
                          unsigned long long, int, BatchLmCont);
template <int MODE, int DEPTH, typename ST, int KIND>
BatchLmFn lpick_loss(bool loss) {
  return loss ? batch_lm_kernel<MODE, DEPTH, ST, KIND, true> : batch_lm_kernel<MODE, DEPTH, ST, KIND, false>;
}
template <int MODE, int DEPTH, typename ST>
BatchLmFn lpick_kind(int kind, bool loss) {
  return kind == KIND_EXPLICIT ? lpick_loss<MODE, DEPTH, ST, KIND_EXPLICIT>(loss)
                               : lpick_loss<MODE, DEPTH, ST, KIND_FACTORED>(loss);
}
template <int MODE, int DEPTH>
BatchLmFn lpick_store(int store, int kind, bool loss) {
  return store == 0 ? lpick_kind<MODE, DEPTH, double>(kind, loss) : lpick_kind<MODE, DEPTH, float>(kind, loss);
}
BatchLmFn lpick(int mode, int depth, int store, int kind, bool loss) {
  switch (mode * 2 + depth) {
    case 0: return lpick_store<MODE_ROT, DEPTH_UNIFORM>(store, kind, loss);
    case 1: return lpick_store<MODE_ROT, DEPTH_PER_MATCH>(store, kind, loss);
    case 2: return lpick_store<MODE_TRAN, DEPTH_UNIFORM>(store, KIND_FACTORED, loss);
    case 3: return lpick_store<MODE_TRAN, DEPTH_PER_MATCH>(store, KIND_FACTORED, loss);
    case 4: return lpick_store<MODE_RT, DEPTH_UNIFORM>(store, kind, loss);
    case 5: return lpick_store<MODE_RT, DEPTH_PER_MATCH>(store, kind, loss);
  }
  return nullptr;
}

typedef void (*ResidentFn)(Planes, unsigned long long, const ResidentRecord*, double*, unsigned long long, unsigned long long,
                           unsigned long long);
template <int MODE, int DEPTH, typename ST, int KIND>
ResidentFn rpick_loss(bool loss) {
  return loss ? resident_sweep_kernel<MODE, DEPTH, ST, KIND, true> : resident_sweep_kernel<MODE, DEPTH, ST, KIND, false>;
}
template <int MODE, int DEPTH, typename ST>
ResidentFn rpick_kind(int kind, bool loss) {
  return kind == KIND_EXPLICIT ? rpick_loss<MODE, DEPTH, ST, KIND_EXPLICIT>(loss)
                               : rpick_loss<MODE, DEPTH, ST, KIND_FACTORED>(loss);
}
template <int MODE, int DEPTH>
ResidentFn rpick_store(int store, int kind, bool loss) {
  return store == 0 ? rpick_kind<MODE, DEPTH, double>(kind, loss) : rpick_kind<MODE, DEPTH, float>(kind, loss);
}
ResidentFn rpick(int mode, int depth, int store, int kind, bool loss) {
  switch (mode * 2 + depth) {
    case 0: return rpick_store<MODE_ROT, DEPTH_UNIFORM>(store, kind, loss);
    case 1: return rpick_store<MODE_ROT, DEPTH_PER_MATCH>(store, kind, loss);
    case 2: return rpick_store<MODE_TRAN, DEPTH_UNIFORM>(store, KIND_FACTORED, loss);
    case 3: return rpick_store<MODE_TRAN, DEPTH_PER_MATCH>(store, KIND_FACTORED, loss);
    case 4: return rpick_store<MODE_RT, DEPTH_UNIFORM>(store, kind, loss);
    case 5: return rpick_store<MODE_RT, DEPTH_PER_MATCH>(store, kind, loss);
  }
  return nullptr;
}

typedef void (*BatchStepFn)(Planes, const PairDesc*, const BatchState*, double, double*, double*, unsigned int*,
                            unsigned long long);
template <int MODE, int DEPTH, typename ST, int KIND>
BatchStepFn spick_loss(bool loss) {
  return loss ? batch_step_kernel<MODE, DEPTH, ST, KIND, true> : batch_step_kernel<MODE, DEPTH, ST, KIND, false>;
}
template <int MODE, int DEPTH, typename ST>
BatchStepFn spick_kind(int kind, bool loss) {
  return kind == KIND_EXPLICIT ? spick_loss<MODE, DEPTH, ST, KIND_EXPLICIT>(loss)
                               : spick_loss<MODE, DEPTH, ST, KIND_FACTORED>(loss);
}
template <int MODE, int DEPTH>
BatchStepFn spick_store(int store, int kind, bool loss) {
  return store == 0 ? spick_kind<MODE, DEPTH, double>(kind, loss) : spick_kind<MODE, DEPTH, float>(kind, loss);
}
BatchStepFn spick(int mode, int depth, int store, int kind, bool loss) {
  switch (mode * 2 + depth) {
    case 0: return spick_store<MODE_ROT, DEPTH_UNIFORM>(store, kind, loss);
    case 1: return spick_store<MODE_ROT, DEPTH_PER_MATCH>(store, kind, loss);
    case 2: return spick_store<MODE_TRAN, DEPTH_UNIFORM>(store, KIND_FACTORED, loss);
    case 3: return spick_store<MODE_TRAN, DEPTH_PER_MATCH>(store, KIND_FACTORED, loss);
    case 4: return spick_store<MODE_RT, DEPTH_UNIFORM>(store, kind, loss);
    case 5: return spick_store<MODE_RT, DEPTH_PER_MATCH>(store, kind, loss);
  }
  return nullptr;
}

typedef void (*BatchDynFn)(Planes, const SweepParams*, const PairDesc*, const BatchDynCtl*, const unsigned int*, int, int, double*);
template <int MODE, int DEPTH, typename ST, int KIND>
BatchDynFn dpick_loss(bool loss) {
  return loss ? batch_sweep_dyn_kernel<MODE, DEPTH, ST, KIND, true> : batch_sweep_dyn_kernel<MODE, DEPTH, ST, KIND, false>;
}
template <int MODE, int DEPTH, typename ST>
BatchDynFn dpick_kind(int kind, bool loss) {
  return kind == KIND_EXPLICIT ? dpick_loss<MODE, DEPTH, ST, KIND_EXPLICIT>(loss) : dpick_loss<MODE, DEPTH, ST, KIND_FACTORED>(loss);
}
template <int MODE, int DEPTH>
BatchDynFn dpick_store(int store, int kind, bool loss) {
  return store == 0 ? dpick_kind<MODE, DEPTH, double>(kind, loss) : dpick_kind<MODE, DEPTH, float>(kind, loss);
}
BatchDynFn dpick(int mode, int depth, int store, int kind, bool loss) {
  switch (mode * 2 + depth) {
    case 0: return dpick_store<MODE_ROT, DEPTH_UNIFORM>(store, kind, loss);
    case 1: return dpick_store<MODE_ROT, DEPTH_PER_MATCH>(store, kind, loss);
    case 2: return dpick_store<MODE_TRAN, DEPTH_UNIFORM>(store, KIND_FACTORED, loss);
    case 3: return dpick_store<MODE_TRAN, DEPTH_PER_MATCH>(store, KIND_FACTORED, loss);
    case 4: return dpick_store<MODE_RT, DEPTH_UNIFORM>(store, kind, loss);
    case 5: return dpick_store<MODE_RT, DEPTH_PER_MATCH>(store, kind, loss);
  }
  return nullptr;
}

}  // namespace

hipError_t launch_batch_step_fused(int mode, int depth, int store, int kind, double huber_delta, const Planes& pl,
                                   const BatchState* state, const PairDesc* desc, int num_pairs, double* packs,
                                   double* packs_host, unsigned int* ticket, unsigned long long seq, hipStream_t stream) {
  if (num_pairs <= 0) return hipSuccess;
  BatchStepFn fn = spick(mode, depth, store, kind, huber_delta > 0.0);
  if (!fn) return hipErrorInvalidValue;
  hipLaunchKernelGGL(fn, dim3(static_cast<unsigned>(num_pairs)), dim3(kBlock), 0, stream, pl, desc, state, huber_delta,
                     packs, packs_host, ticket, seq);
  return hipGetLastError();
}

hipError_t launch_batch_lm(int mode, int depth, int store, int kind, const Planes& pl, const PairDesc* desc,
                           BatchLmIo* io, const sba_lm_options& opt, int num_pairs, unsigned int* ticket,
                           unsigned long long* seq_host_dev, unsigned long long seq, hipStream_t stream, int sweep_cap,
                           void* cont_state, SweepParams* cont_params, double* cont_frames, int* cont_done) {
  if (num_pairs <= 0) return hipSuccess;
  BatchLmFn fn = lpick(mode, depth, store, kind, opt.huber_delta > 0.0);
  if (!fn) return hipErrorInvalidValue;
  const BatchLmCont cont{static_cast<BatchLmDynState*>(cont_state), cont_params, cont_frames, cont_done};
  hipLaunchKernelGGL(fn, dim3(static_cast<unsigned>(num_pairs)), dim3(kBlock), 0, stream, pl, desc, io, opt, ticket,
                     seq_host_dev, seq, cont_state ? sweep_cap : 0, cont);
  return hipGetLastError();
}

// After a capped batch_lm_kernel: the pairs it handed over become the first active list (parity 1).
hipError_t launch_batch_dyn_first_list(BatchDynCtl* ctl, unsigned int* active, const int* done, int num_pairs, hipStream_t stream) {
  if (num_pairs <= 0) return hipSuccess;
  hipLaunchKernelGGL(batch_dyn_identity_kernel, dim3((static_cast<unsigned>(num_pairs) + 255u) / 256u), dim3(256), 0, stream, ctl, active,
                     num_pairs);
  hipLaunchKernelGGL(batch_dyn_compact_kernel, dim3(1), dim3(256), 0, stream, ctl, active, done, num_pairs, 0,
                     static_cast<unsigned long long*>(nullptr), 0ull);
  return hipGetLastError();
}

hipError_t launch_resident_sweep(int mode, int depth, int store, int kind, bool loss, const Planes& pl, size_t n,
                                 const ResidentRecord* rec_dev, double* host_pack_dev, unsigned long long first_cmd_seq,
                                 unsigned long long first_pack_seq, unsigned long long idle_ticks, hipStream_t stream) {
  ResidentFn fn = rpick(mode, depth, store, kind, loss);
  if (!fn) return hipErrorInvalidValue;
  hipLaunchKernelGGL(fn, dim3(1), dim3(kBlock), 0, stream, pl, static_cast<unsigned long long>(n), rec_dev, host_pack_dev,
                     first_cmd_seq, first_pack_seq, idle_ticks);
  return hipGetLastError();
}

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

// The batched sweep kernel alone (params already prepared on the device), no fold: timing primitive.
hipError_t launch_batch_sweep_only(int mode, int depth, int store, int kind, bool loss, const Planes& pl,
                                   const SweepParams* params, const PairDesc* desc, int num_pairs, int bpp,
                                   double* partials, hipStream_t stream) {
  if (num_pairs <= 0) return hipSuccess;
  BatchFn fn = bpick(mode, depth, store, kind, loss);
  if (!fn) return hipErrorInvalidValue;
  hipLaunchKernelGGL(fn, dim3(static_cast<unsigned>(num_pairs) * bpp), dim3(kBlock), 0, stream, pl, params, desc,
                     bpp, partials);
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

hipError_t launch_batch_dyn_compact(BatchDynCtl* ctl, unsigned int* active, const int* done, int num_pairs, int parity,
                                    unsigned long long* host_words, unsigned long long seq, hipStream_t stream) {
  if (num_pairs <= 0) return hipSuccess;
  hipLaunchKernelGGL(batch_dyn_compact_kernel, dim3(1), dim3(256), 0, stream, ctl, active, done, num_pairs, parity, host_words, seq);
  return hipGetLastError();
}

size_t batch_lm_dyn_state_bytes() { return sizeof(BatchLmDynState); }

hipError_t launch_batch_lm_dyn_init(int mode, int depth, int kind, const PairDesc* desc, const BatchLmIo* io, const sba_lm_options& opt,
                                    int num_pairs, void* state, SweepParams* params, double* frames, BatchDynCtl* ctl,
                                    unsigned int* active, int* done, hipStream_t stream) {
  if (num_pairs <= 0) return hipSuccess;
  hipLaunchKernelGGL(batch_lm_dyn_init_kernel, dim3(static_cast<unsigned>(num_pairs)), dim3(64), 0, stream, mode, depth, kind, desc, io, opt,
                     num_pairs, static_cast<BatchLmDynState*>(state), params, frames, ctl, active, done);
  return hipGetLastError();
}

// One LM iteration of every pair that is still iterating: sweep (sweep_grid blocks dealt to the active pairs), feed, compaction.
// host_words / seq: as batch_dyn_compact_kernel (null: no publication after this iteration).
hipError_t launch_batch_lm_dyn_pass(int mode, int depth, int store, int kind, const Planes& pl, const PairDesc* desc,
                                    const sba_lm_options& opt, int num_pairs, int parity, int sweep_grid, void* state,
                                    SweepParams* params, double* frames, BatchDynCtl* ctl, unsigned int* active, int* done,
                                    double* partials, BatchLmIo* io, unsigned long long* host_words, unsigned long long seq,
                                    hipStream_t stream) {
  if (num_pairs <= 0) return hipSuccess;
  BatchDynFn fn = dpick(mode, depth, store, kind, opt.huber_delta > 0.0);
  if (!fn || sweep_grid < 1) return hipErrorInvalidValue;
  hipLaunchKernelGGL(fn, dim3(static_cast<unsigned>(sweep_grid)), dim3(kBlock), 0, stream, pl, params, desc, ctl, active, parity, num_pairs,
                     partials);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return e;
  const unsigned feed_grid = static_cast<unsigned>(num_pairs < 1024 ? num_pairs : 1024);
  hipLaunchKernelGGL(batch_lm_dyn_feed_kernel, dim3(feed_grid), dim3(64), 0, stream, mode, depth, kind, desc, opt, num_pairs, parity,
                     static_cast<unsigned>(sweep_grid), ctl, active, partials, static_cast<BatchLmDynState*>(state), params, frames, io,
                     done);
  e = hipGetLastError();
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(batch_dyn_compact_kernel, dim3(1), dim3(256), 0, stream, ctl, active, done, num_pairs, parity, host_words, seq);
  return hipGetLastError();
}

}  // namespace sba
