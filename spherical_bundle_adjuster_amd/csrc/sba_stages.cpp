// Entry points of the stages either side of the sweep (include/sba_hip.h): the d-only stage (reference
// spherical_bundle_adjuster.cpp:1004-1063) and the 8-point initial guess (.cpp:47-181).
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <mutex>

#include "sba_depth_solver.hpp"
#include "sba_epipolar.hpp"
#include "sba_lm.hpp"
#include "sba_problem.hpp"
#include "sba_rotation.hpp"

using sba::shim::allreduce_buffer;
using sba::shim::allreduce_pack;
using sba::shim::fetch_pack_raw;

extern "C" {

// ---- callers / data formats either side of the path ----------------------------------------------
int sba_problem_solve_depths(sba_problem* p, const double rot[3], const double tran[3], double lambda,
                             double c, const sba_lm_options* opt, double* d12_out, sba_lm_summary* summary) {
  if (!p || !rot || !tran) return sba::set_error(SBA_ERR_INVALID_ARG, "null argument");
  SBA_REFUSE_POISONED(p);
  if (!p->uploaded) return sba::set_error(SBA_ERR_NOT_UPLOADED, "no correspondences uploaded");
  if (!p->has_d12 && p->n > 0) return sba::set_error(SBA_ERR_INVALID_ARG, "the d-only stage needs per-match depths uploaded");
  // Sharded problem: the nine global reductions of every pass are all-reduced over the attached transport (the two
  // maxima travel as one slot per rank each, so at most 8 shards), and every rank replays the same step logic.
  const bool collective = p->comm != nullptr || p->hook != nullptr || p->peer_ready;
  if (collective && (p->shard_count < 1 || p->shard_count > 8 || p->shard_rank < 0 || p->shard_rank >= p->shard_count))
    return sba::set_error(SBA_ERR_UNSUPPORTED, "d-only stage over a transport needs 1..8 shards (sba_problem_set_shard); have %d/%d",
                p->shard_rank, p->shard_count);
  SBA_TRY_HIP(hipSetDevice(p->device));
  sba_lm_options o;
  if (opt) o = *opt; else sba::lm_default_options(&o);
  sba_lm_summary local;
  sba_lm_summary* sum = summary ? summary : &local;
  std::memset(sum, 0, sizeof(*sum));
  const auto t_start = std::chrono::steady_clock::now();
  const size_t n = p->n, elems = std::max<size_t>(p->plane_elems, 2);

  // work planes: candidate depths, Jacobi scaling, LM diagonal (6 x n doubles), block partials, results.  Zeroed once:
  // the kernel writes whole pairs only, and a candidate plane becomes the problem's depth plane when a step is accepted
  // -- its padding must be zeros like the uploaded planes' (a later per-match sweep loads it in its ragged tail).
  // The scratch lives in the handle and is reused while it fits (three hipMalloc / hipFree pairs per call were a tenth of the
  // stage at the reference's problem sizes); a handle that gets poisoned simply keeps it.
  const size_t max_grid_rows = static_cast<size_t>(p->num_cus) * 16 + 1;
  const size_t need = (4 * elems + (max_grid_rows + 1) * sba::DEPTH_ROW) * sizeof(double);
  if (p->depth_scratch_bytes < need || p->depth_scratch_bytes > 4 * need + (size_t(1) << 20)) {
    if (p->depth_scratch) SBA_TRY_HIP(hipFree(p->depth_scratch));
    p->depth_scratch = nullptr; p->depth_scratch_bytes = 0;
    SBA_TRY_HIP(hipMalloc(&p->depth_scratch, need));
    p->depth_scratch_bytes = need;
  }
  double* work = static_cast<double*>(p->depth_scratch);
  SBA_TRY_HIP(hipMemsetAsync(work, 0, 4 * elems * sizeof(double), p->stream));
  double *c1 = work, *c2 = work + elems, *sc1 = work + 2 * elems, *sc2 = work + 3 * elems;
  // one resident wave of blocks (occupancy of the kernel, SBA_DEPTH_BLOCKS_PER_CU caps it), grid-stride inside
  int& occ = p->depth_occ[p->store];
  if (occ == 0) {
    SBA_TRY_HIP(sba::depth_blocks_per_cu(p->store, &occ));
    occ = std::max(1, occ);
  }
  int cap = 8;
  if (const char* env = std::getenv("SBA_DEPTH_BLOCKS_PER_CU")) { const int v = std::atoi(env); if (v >= 1 && v <= 16) cap = v; }
  const int grid = static_cast<int>(std::min<size_t>(((n + 1) / 2 + 255) / 256,
                                                     static_cast<size_t>(p->num_cus) * std::max(1, std::min(occ, cap))));
  double *out_dev = work + 4 * elems, *partials = out_dev + sba::DEPTH_ROW;     // [DEPTH_ROW] results, then [grid][DEPTH_ROW] rows

  sba::Planes pl;
  for (int k = 0; k < 3; ++k) { pl.x1[k] = p->coord[k]; pl.x2[k] = p->coord[3 + k]; }
  pl.d1 = p->dplane[0]; pl.d2 = p->dplane[1];
  sba::DepthParams prm;
  double G[27];
  sba::rotation_and_derivatives(rot, prm.R, G);
  for (int i = 0; i < 3; ++i) prm.t[i] = tran[i];
  prm.lambda = lambda; prm.c = c;
  prm.min_diagonal = o.min_lm_diagonal; prm.max_diagonal = o.max_lm_diagonal;
  prm.jacobi_scaling = o.jacobi_scaling; prm.n = n;
  // Plain stores for the candidate planes: 167.4 us per pass against 181.3 us with non-temporal stores (rocprofv3
  // averages over 54 launches at 10^7 matches, same box; WRITE_SIZE is 160.8 MB either way -- profiles/r03_depth_stores.md).
  prm.stream_stores = 0;
  if (const char* env = std::getenv("SBA_DEPTH_NT_STORES")) prm.stream_stores = std::atoi(env) != 0 ? 1 : 0;

  double* cur1 = p->dplane[0];
  double* cur2 = p->dplane[1];
  int rc_final = SBA_OK;
  double out[sba::DEPTH_OUT_COUNT] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
  // One device pass at the current depths: the step delta of the damped system, the candidate P(d + alpha delta)
  // into (c1, c2), and the nine reductions -- all-reduced when the problem is sharded.
  auto pass = [&](const sba::DepthPassRequest& rq) -> int {
    prm.radius = rq.radius; prm.inv_radius = 1.0 / rq.radius; prm.alpha = rq.alpha;
    prm.first_iteration = rq.first ? 1 : 0; prm.reuse_diagonal = rq.keep_diagonal ? 1 : 0;
    if (collective) {
      SBA_TRY_HIP(sba::launch_depth_step(p->store, pl, cur1, cur2, c1, c2, sc1, sc2, prm, partials, grid,
                                         p->pack_dev, nullptr, 0, p->shard_rank, p->stream));
      int rc = allreduce_pack(p);
      if (rc) return rc;
      double raw[SBA_PACK_SIZE];
      rc = fetch_pack_raw(p, raw);
      if (rc) return rc;
      for (int k = 0; k < sba::DEPTH_OUT_SUMS; ++k) out[k] = raw[k];
      out[sba::DEPTH_OUT_GMAX] = out[sba::DEPTH_OUT_DMAX] = 0.0;
      for (int r = 0; r < p->shard_count; ++r) {
        out[sba::DEPTH_OUT_GMAX] = std::max(out[sba::DEPTH_OUT_GMAX], raw[8 + r]);
        out[sba::DEPTH_OUT_DMAX] = std::max(out[sba::DEPTH_OUT_DMAX], raw[16 + r]);
      }
      return SBA_OK;
    }
    if (p->publish) {
      // the finalize kernel publishes the results itself; the host polls the sequence word (see fetch_pack_raw)
      const unsigned long long seq = ++p->seq;
      SBA_TRY_HIP(sba::launch_depth_step(p->store, pl, cur1, cur2, c1, c2, sc1, sc2, prm, partials, grid,
                                         out_dev, p->pack_host_dev, seq, -1, p->stream));
      const int rc = sba::wait_for_sequence(reinterpret_cast<volatile unsigned long long*>(p->pack_host + 24), seq,
                                            p->stream, "d-only pass", &p->poisoned);
      if (rc) return rc;
    } else {
      SBA_TRY_HIP(sba::launch_depth_step(p->store, pl, cur1, cur2, c1, c2, sc1, sc2, prm, partials, grid,
                                         out_dev, nullptr, 0, -1, p->stream));
      SBA_TRY_HIP(hipMemcpyAsync(p->pack_host, out_dev, sizeof(out), hipMemcpyDeviceToHost, p->stream));
      { const int _rc = sba::stream_wait(p->stream, "stream synchronisation", &p->poisoned); if (_rc) return _rc; }
    }
    std::memcpy(out, p->pack_host, sizeof(out));
    return SBA_OK;
  };

  // Small, unsharded problem: the whole stage in ONE launch -- the problem as a batch of one pair through
  // batch_depth_solve_kernel: the DepthStageSolver below runs on the device next to its passes (same source), the kernel
  // copies the result back into the problem's planes itself.  SBA_SMALL_ONE_LAUNCH=0 keeps the resident evaluator.
  if (p->small_rec != nullptr && p->publish && !collective && n > 0 && n <= p->one_launch_max_n_depth && sba::shim::small_one_launch(true)) {
    sba::shim::SmallRecord* rec = static_cast<sba::shim::SmallRecord*>(p->small_rec);
    sba::shim::SmallRecord* rec_dev = static_cast<sba::shim::SmallRecord*>(p->small_rec_dev);
    rec->desc = sba::PairDesc{0ull, n, sba::kPairTile, 0ull};
    for (int i = 0; i < 9; ++i) rec->cst.R[i] = prm.R[i];
    for (int i = 0; i < 3; ++i) rec->cst.t[i] = tran[i];
    rec->io = sba::BatchLmIo{};
    rec->io.status = SBA_ERR_NUMERIC;
    const unsigned long long seq = ++p->small_seq;
    SBA_TRY_HIP(sba::launch_batch_depth_solve(p->store, pl, &rec_dev->desc, &rec_dev->cst, 1, lambda, c, o, p->dplane[0], p->dplane[1], c1, c2,
                                              sc1, sc2, nullptr, nullptr, &rec_dev->io, p->ticket, const_cast<unsigned long long*>(&rec_dev->seq),
                                              seq, p->stream));
    const int wrc = sba::wait_for_sequence(&rec->seq, seq, p->stream, "one-launch d-only stage", &p->poisoned);
    if (wrc) return wrc;
    *sum = rec->io.summary;
    sum->seconds_total = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_start).count();
    rc_final = rec->io.status;
    if (d12_out && n > 0) {
      sba::DeviceBuffer aos(&p->poisoned);
      SBA_TRY_HIP(aos.alloc(2 * n * sizeof(double)));
      SBA_TRY_HIP(sba::launch_planes_to_d12(p->dplane[0], p->dplane[1], n, aos.as<double>(), p->stream));
      SBA_TRY_HIP(hipMemcpyAsync(d12_out, aos.ptr, 2 * n * sizeof(double), hipMemcpyDeviceToHost, p->stream));
      { const int _rc = sba::stream_wait(p->stream, "stream synchronisation", &p->poisoned); if (_rc) return _rc; }
    }
    if (rc_final != SBA_OK) return sba::set_error(rc_final, "d-only stage failed: non-finite cost or 5 consecutive invalid steps");
    return SBA_OK;
  }
  // Otherwise (SBA_SMALL_ONE_LAUNCH=0): one resident single-block kernel serves every pass of the stage (sba_resident.hpp); the
  // step logic below is the same, a pass is a command instead of two launches.  The planes alternate as in the launch
  // path: `flip` says that the work planes currently hold the depths and the problem's planes take the candidates.
  sba::shim::ResidentSession session(p);
  bool flip = false;
  if (sba::shim::resident_eligible(p, true)) {
    const int rc = session.start_depth(p->dplane[0], p->dplane[1], c1, c2, sc1, sc2);
    if (rc) return rc;
  }
  auto resident_pass = [&](const sba::DepthPassRequest& rq) -> int {
    double payload[22];
    payload[0] = static_cast<double>(sba::RESIDENT_OP_DEPTH);
    for (int k = 0; k < 9; ++k) payload[1 + k] = prm.R[k];
    for (int k = 0; k < 3; ++k) payload[10 + k] = prm.t[k];
    payload[13] = prm.lambda; payload[14] = prm.c; payload[15] = rq.radius; payload[16] = 1.0 / rq.radius;
    payload[17] = prm.min_diagonal; payload[18] = prm.max_diagonal; payload[19] = rq.alpha;
    payload[20] = static_cast<double>((rq.first ? 1 : 0) | (rq.keep_diagonal ? 2 : 0) | (prm.jacobi_scaling ? 4 : 0) | (flip ? 8 : 0));
    const unsigned long long n_bits = n;
    std::memcpy(payload + 21, &n_bits, sizeof(double));
    return session.call(payload, 22, out, sba::DEPTH_OUT_COUNT);
  };

  // The step logic (trust region + Ceres' projected line search) is a host state machine, sba_depth_solver.hpp: it
  // names the next pass, is fed the pass's reductions, and says when the candidate planes become the current depths.
  sba::DepthStageSolver solver;
  solver.start(o);
  while (!solver.done()) {
    const int rc = session.active() ? resident_pass(solver.request()) : pass(solver.request());
    if (rc) return rc;
    solver.feed(out);
    if (solver.take_candidate()) { std::swap(cur1, c1); std::swap(cur2, c2); flip = !flip; }
  }
  {
    const int rc = session.end();
    if (rc) return rc;
  }
  *sum = solver.summary();
  sum->seconds_total = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_start).count();
  rc_final = solver.status();
  // the problem's depth planes must end up holding the result
  if (cur1 != p->dplane[0]) {
    SBA_TRY_HIP(hipMemcpyAsync(p->dplane[0], cur1, elems * sizeof(double), hipMemcpyDeviceToDevice, p->stream));
    SBA_TRY_HIP(hipMemcpyAsync(p->dplane[1], cur2, elems * sizeof(double), hipMemcpyDeviceToDevice, p->stream));
  }
  if (d12_out && n > 0) {
    sba::DeviceBuffer aos(&p->poisoned);
    SBA_TRY_HIP(aos.alloc(2 * n * sizeof(double)));
    SBA_TRY_HIP(sba::launch_planes_to_d12(p->dplane[0], p->dplane[1], n, aos.as<double>(), p->stream));
    SBA_TRY_HIP(hipMemcpyAsync(d12_out, aos.ptr, 2 * n * sizeof(double), hipMemcpyDeviceToHost, p->stream));
    { const int _rc = sba::stream_wait(p->stream, "stream synchronisation", &p->poisoned); if (_rc) return _rc; }
  }
  { const int _rc = sba::stream_wait(p->stream, "stream synchronisation", &p->poisoned); if (_rc) return _rc; }
  if (rc_final != SBA_OK) return sba::set_error(rc_final, "d-only stage failed: non-finite cost or 5 consecutive invalid steps");
  return SBA_OK;
}

// ---- 8-point initial guess (reference .cpp:47-181) -------------------------------------------------------
int sba_problem_epipolar_moments(sba_problem* p, double* groups) {
  if (!p || !groups) return sba::set_error(SBA_ERR_INVALID_ARG, "null argument");
  SBA_REFUSE_POISONED(p);
  if (!p->uploaded) return sba::set_error(SBA_ERR_NOT_UPLOADED, "no correspondences uploaded");
  SBA_TRY_HIP(hipSetDevice(p->device));
  const size_t nvec = (p->n + 1) / 2;
  // 256-thread blocks (4 waves x 64 vectors of 2 matches), two resident per CU
  int per_cu = 2;
  if (const char* e = std::getenv("SBA_EPI_BLOCKS_PER_CU")) { const int v = std::atoi(e); if (v >= 1 && v <= 8) per_cu = v; }   // tuning only
  const int grid = static_cast<int>(std::min<size_t>((nvec + 255) / 256, static_cast<size_t>(p->num_cus) * per_cu));
  const size_t gsz = static_cast<size_t>(sba::epi::kGroups) * sba::epi::kMom;
  // scratch (block partials + the groups) lives in the handle: allocating 23 MB per call cost more than the pass
  const size_t need = (static_cast<size_t>(std::max(grid, 1)) + 1) * gsz;
  if (p->epi_scratch_elems < need) {
    if (p->epi_scratch) SBA_TRY_HIP(hipFree(p->epi_scratch));
    p->epi_scratch = nullptr; p->epi_scratch_elems = 0;
    SBA_TRY_HIP(hipMalloc(reinterpret_cast<void**>(&p->epi_scratch), need * sizeof(double)));
    p->epi_scratch_elems = need;
  }
  double *groups_dev = p->epi_scratch, *partials = p->epi_scratch + gsz;
  sba::Planes pl;
  for (int k = 0; k < 3; ++k) { pl.x1[k] = p->coord[k]; pl.x2[k] = p->coord[3 + k]; }
  pl.d1 = p->dplane[0]; pl.d2 = p->dplane[1];
  SBA_TRY_HIP(sba::launch_epipolar_moments(p->store, pl, p->n, partials, grid, groups_dev, p->stream));
  // Sharded problem: group g of the whole problem is the union of every shard's group g, so the 64 x 45 sums are
  // all-reduced and every rank derives the same initial guess from the same numbers.
  const bool collective = p->comm != nullptr || p->hook != nullptr || p->peer_ready;
  if (collective) {
    const int rc = allreduce_buffer(p, groups_dev, gsz);
    if (rc) return rc;
  }
  SBA_TRY_HIP(hipMemcpyAsync(groups, groups_dev, gsz * sizeof(double), hipMemcpyDeviceToHost, p->stream));
  { const int _rc = sba::stream_wait(p->stream, "stream synchronisation", &p->poisoned); if (_rc) return _rc; }
  if (collective && p->peer_ready && reinterpret_cast<volatile unsigned long long*>(p->pack_host)[25] != 0)
    return sba::set_error(SBA_ERR_COMM, "peer exchange timed out waiting for another rank's group moments");
  return SBA_OK;
}

int sba_initial_guess_from_moments(const double* groups, int trials, double subset_fraction, unsigned long long seed,
                                   double rot_euler[3], double tran[3], int* num_candidates) {
  if (!groups || !rot_euler || !tran) return sba::set_error(SBA_ERR_INVALID_ARG, "null argument");
  if (trials < 1 || !(subset_fraction > 0.0) || subset_fraction > 1.0)
    return sba::set_error(SBA_ERR_INVALID_ARG, "bad trials / subset_fraction");
  const sba::epi::GuessResult r = sba::epi::initial_guess_from_groups(groups, trials, subset_fraction, seed,
                                                                       sba::host_threads());
  if (num_candidates) *num_candidates = r.num_candidates;
  if (r.picked < 0) return sba::set_error(SBA_ERR_NUMERIC, "no valid rotation candidate (all Euler angles >= 1.57)");
  for (int i = 0; i < 3; ++i) { rot_euler[i] = r.euler[i]; tran[i] = r.tran[i]; }
  return SBA_OK;
}

namespace {
// The stream the reference's random_array draws from (see sba_epipolar.hpp): by default the library's own copy of glibc's
// generator in its never-seeded state, running on across calls; SBA_GUESS_RAND=libc: the process's rand().
std::mutex g_ref_rand_mutex;
sba::epi::GlibcRand g_ref_rand(1);
bool use_libc_rand() {
  const char* env = std::getenv("SBA_GUESS_RAND");
  return env && std::strcmp(env, "libc") == 0;
}
void draw_reference_subsets(int n, int trials, double fraction, int* indices) {
  std::lock_guard<std::mutex> lock(g_ref_rand_mutex);
  if (use_libc_rand()) sba::epi::reference_trial_subsets(n, trials, fraction, indices, [] { return std::rand(); });
  else sba::epi::reference_trial_subsets(n, trials, fraction, indices, [] { return g_ref_rand.next(); });
}
}  // namespace

int sba_reference_rand_seed(unsigned int seed) {
  std::lock_guard<std::mutex> lock(g_ref_rand_mutex);
  if (use_libc_rand()) std::srand(seed); else g_ref_rand.reseed(seed);
  return SBA_OK;
}

int sba_reference_rand_next(void) {
  std::lock_guard<std::mutex> lock(g_ref_rand_mutex);
  return use_libc_rand() ? std::rand() : g_ref_rand.next();
}

int sba_reference_trial_subsets(int n, int trials, double subset_fraction, int* indices, int* sample_n_out) {
  if (n < 0 || trials < 0 || !(subset_fraction > 0.0) || subset_fraction > 1.0)
    return sba::set_error(SBA_ERR_INVALID_ARG, "bad n / trials / subset_fraction");
  if (sample_n_out) *sample_n_out = sba::epi::reference_sample_size(n, subset_fraction);
  if (indices) draw_reference_subsets(n, trials, subset_fraction, indices);
  return SBA_OK;
}

int sba_problem_epipolar_subset_moments(sba_problem* p, const int* indices, int trials, int sample_n, double* moments) {
  if (!p || !moments || (trials > 0 && sample_n > 0 && !indices)) return sba::set_error(SBA_ERR_INVALID_ARG, "null argument");
  SBA_REFUSE_POISONED(p);
  if (!p->uploaded) return sba::set_error(SBA_ERR_NOT_UPLOADED, "no correspondences uploaded");
  if (trials < 0 || sample_n < 0) return sba::set_error(SBA_ERR_INVALID_ARG, "bad trials / sample_n");
  if (sba::shim::is_collective(p))
    return sba::set_error(SBA_ERR_UNSUPPORTED, "index lists address one resident shard: detach the transport (sharded problems use "
                                               "the group sampling of sba_problem_initial_guess)");
  const size_t count = static_cast<size_t>(trials) * static_cast<size_t>(sample_n);
  for (size_t k = 0; k < count; ++k)
    if (indices[k] < 0 || static_cast<size_t>(indices[k]) >= p->n)
      return sba::set_error(SBA_ERR_INVALID_ARG, "index %d of list entry %zu is outside the %zu resident matches", indices[k], k, p->n);
  const size_t msz = static_cast<size_t>(trials) * sba::epi::kMom;
  if (trials == 0) return SBA_OK;
  if (sample_n == 0) { std::memset(moments, 0, msz * sizeof(double)); return SBA_OK; }
  SBA_TRY_HIP(hipSetDevice(p->device));
  // scratch lives in the handle (a hipMalloc / hipFree pair per call costs more than the pass): moments first, then the lists
  const size_t need = msz * sizeof(double) + count * sizeof(int);
  if (p->subset_scratch_bytes < need) {
    if (p->subset_scratch) SBA_TRY_HIP(hipFree(p->subset_scratch));
    p->subset_scratch = nullptr; p->subset_scratch_bytes = 0;
    SBA_TRY_HIP(hipMalloc(&p->subset_scratch, need));
    p->subset_scratch_bytes = need;
  }
  double* mom_dev_ptr = static_cast<double*>(p->subset_scratch);
  int* idx_dev_ptr = reinterpret_cast<int*>(mom_dev_ptr + msz);
  SBA_TRY_HIP(hipMemcpyAsync(idx_dev_ptr, indices, count * sizeof(int), hipMemcpyHostToDevice, p->stream));
  sba::Planes pl;
  for (int k = 0; k < 3; ++k) { pl.x1[k] = p->coord[k]; pl.x2[k] = p->coord[3 + k]; }
  pl.d1 = p->dplane[0]; pl.d2 = p->dplane[1];
  SBA_TRY_HIP(sba::launch_epipolar_subset_moments(p->store, pl, p->n, idx_dev_ptr, trials, sample_n, mom_dev_ptr, p->stream));
  SBA_TRY_HIP(hipMemcpyAsync(moments, mom_dev_ptr, msz * sizeof(double), hipMemcpyDeviceToHost, p->stream));
  { const int _rc = sba::stream_wait(p->stream, "subset moments", &p->poisoned); if (_rc) return _rc; }
  return SBA_OK;
}

int sba_problem_initial_guess_reference(sba_problem* p, int trials, double subset_fraction, double rot_euler[3],
                                        double tran[3], int* num_candidates) {
  if (!p || !rot_euler || !tran) return sba::set_error(SBA_ERR_INVALID_ARG, "null argument");
  SBA_REFUSE_POISONED(p);
  if (!p->uploaded) return sba::set_error(SBA_ERR_NOT_UPLOADED, "no correspondences uploaded");
  if (trials < 1 || !(subset_fraction > 0.0) || subset_fraction > 1.0)
    return sba::set_error(SBA_ERR_INVALID_ARG, "bad trials / subset_fraction");
  if (p->n > static_cast<size_t>(SBA_REFERENCE_SAMPLING_MAX_N))
    return sba::set_error(SBA_ERR_UNSUPPORTED, "reference sampling shuffles all %zu match indices per trial: meant for at most %d "
                                               "matches (use sba_problem_initial_guess)", p->n, SBA_REFERENCE_SAMPLING_MAX_N);
  const int n = static_cast<int>(p->n);
  const int sample_n = sba::epi::reference_sample_size(n, subset_fraction);
  if (sample_n < 1)
    return sba::set_error(SBA_ERR_INVALID_ARG, "%d matches give an empty subset (the reference's cv::SVDecomp of a 0 x 9 matrix fails too)", n);
  std::vector<int> indices(static_cast<size_t>(trials) * sample_n);
  draw_reference_subsets(n, trials, subset_fraction, indices.data());
  std::vector<double> moments(static_cast<size_t>(trials) * sba::epi::kMom);
  const int rc = sba_problem_epipolar_subset_moments(p, indices.data(), trials, sample_n, moments.data());
  if (rc) return rc;
  const sba::epi::GuessResult r = sba::epi::initial_guess_from_trial_moments(moments.data(), trials, sample_n);
  if (num_candidates) *num_candidates = r.num_candidates;
  if (r.picked < 0) return sba::set_error(SBA_ERR_NUMERIC, "no valid rotation candidate (all Euler angles >= 1.57)");
  for (int i = 0; i < 3; ++i) { rot_euler[i] = r.euler[i]; tran[i] = r.tran[i]; }
  return SBA_OK;
}

int sba_problem_initial_guess(sba_problem* p, int trials, double subset_fraction, unsigned long long seed,
                              double rot_euler[3], double tran[3], int* num_candidates) {
  if (const char* env = std::getenv("SBA_GUESS_SAMPLING"))
    if (std::strcmp(env, "reference") == 0 && p && !sba::shim::is_collective(p) && p->n >= 4 &&
        p->n <= static_cast<size_t>(SBA_REFERENCE_SAMPLING_MAX_N))
      return sba_problem_initial_guess_reference(p, trials, subset_fraction, rot_euler, tran, num_candidates);
  std::vector<double> groups(static_cast<size_t>(sba::epi::kGroups) * sba::epi::kMom);
  const int rc = sba_problem_epipolar_moments(p, groups.data());
  if (rc) return rc;
  return sba_initial_guess_from_moments(groups.data(), trials, subset_fraction, seed, rot_euler, tran, num_candidates);
}

}  // extern "C"
