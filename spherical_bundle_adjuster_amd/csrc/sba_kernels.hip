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
//
// This file: the single-problem sweep kernel, the final reduction / peer exchange / host hand-over kernels and their
// launchers.  Shared device core: sba_sweep_core.hpp; batched kernels: sba_batch_kernels.hip; upload / key-point / cubemap
// kernels: sba_side.hip.
#include "sba_publish.hpp"
#include "sba_sweep_core.hpp"

namespace sba {
namespace {

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
// ---- direct peer exchange over xGMI: the all-reduce of the 24-double pack without a collective library ----------
// Every rank owns an "inbox" in fine-grained device memory, mapped into all peers through HIP IPC:
//     inbox[parity][source rank][32]   (24 doubles of payload, word 31 = sequence number; 256 B per slot)
// One wave per rank: (1) store the local pack into slot [parity][my rank] of EVERY rank's inbox (peer stores travel
// over xGMI), (2) system-scope release, (3) store the sequence number into each of those slots, (4) poll the own
// inbox until all nranks slots carry this sequence number, (5) system-scope acquire, (6) sum the slots in RANK ORDER
// -- every rank adds the same numbers in the same order, so all ranks obtain the bit-identical result -- and
// (7) publish the pack to the host like publish_kernel.  Two parities: a rank can run at most one exchange ahead of
// the slowest (it needs everybody's slot of round s before it can finish round s), so round s+1 never overwrites a
// slot somebody still reads.  Waits are bounded in TIME (px.timeout_ticks of the constant-rate wall clock); a timeout
// sets the sticky device word px.sticky, and word 25 of the host pack carries that word with every publication, so the
// host reports SBA_ERR_COMM -- also for a timeout in an earlier exchange of a back-to-back burst -- instead of waiting
// forever or summing stale slots unnoticed.
// The exchange as executed by ONE wave (lanes 0..23 hold the local pack in `v`); returns the all-reduced value for the
// lane and whether every source rank arrived in time.
__device__ __forceinline__ double peer_exchange_wave(double v, const PeerInboxes& px, unsigned long long seq,
                                                     int lane, bool* all_ok) {
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
    const long long t0 = wall_clock64();
    while (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != seq) {
      if (static_cast<unsigned long long>(wall_clock64() - t0) > px.timeout_ticks) { ok = false; break; }
      __builtin_amdgcn_s_sleep(2);
    }
  }
  const bool timed_out = __ballot(!ok) != 0ull;
  if (timed_out && lane == 0)
    __hip_atomic_store(px.sticky, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  // every exchange reports the sticky word, not just its own wait: an earlier timeout stays visible
  *all_ok = !timed_out && __hip_atomic_load(px.sticky, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0ull;
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
  if (lane < 24) host_store(pack_host + lane, v);
  if (lane == 0) host_store(reinterpret_cast<unsigned long long*>(pack_host) + 25, ok ? 0ull : 1ull);
  host_release();
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
                                                           unsigned long long host_seq) {
  const int lane = threadIdx.x;
  bool all_ok = true;
  const double tot = peer_exchange_wave(lane < 24 ? pack_local[lane] : 0.0, px, seq, lane, &all_ok);
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
                                                        PeerInboxes px, unsigned long long xseq) {
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
  if (px.nranks > 0) tot = peer_exchange_wave(tot, px, xseq, lane, &ok);
  if (lane < 24) pack_out[lane] = tot;
  if (pack_host) publish_wave(tot, pack_host, seq, ok, lane);
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

}  // namespace

int points_per_lane(int store) { return store == 0 ? 2 : 4; }

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
                                hipStream_t stream) {
  hipLaunchKernelGGL(peer_exchange_kernel, dim3(1), dim3(64), 0, stream, pack_local, px, xseq, pack_out, pack_host_dev,
                     host_seq);
  return hipGetLastError();
}

hipError_t launch_finalize(const double* partials, int nblocks, double* pack_out, double* pack_host_dev,
                           unsigned long long seq, const PeerInboxes* px, unsigned long long xseq,
                           hipStream_t stream) {
  PeerInboxes none;
  for (auto& q : none.inbox) q = nullptr;
  none.nranks = 0;
  none.rank = 0;
  none.timeout_ticks = 0;
  none.sticky = nullptr;
  hipLaunchKernelGGL(finalize_kernel, dim3(1), dim3(1024), 0, stream, partials, nblocks, pack_out, pack_host_dev, seq,
                     px ? *px : none, xseq);
  return hipGetLastError();
}

}  // namespace sba
