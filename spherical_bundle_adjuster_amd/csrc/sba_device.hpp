// Internal interface between the C-ABI translation units (sba_shim / sba_transport / sba_stages / sba_batch .cpp) and the HIP kernels
// (sba_kernels.hip, sba_batch_kernels.hip, sba_side.hip, sba_depth.hip, sba_epipolar.hip).  Nothing here is exported.
#pragma once
#include <hip/hip_runtime.h>
#include <cstddef>
#include <cstdint>

#include "../../include/sba_hip.h"
#include "sba_depth_solver.hpp"
#include "sba_rotation.hpp"

namespace sba {

// Wave-uniform state of one sweep, passed BY VALUE as a kernel argument (kernarg segment ->
// scalar loads), staged in LDS by the kernel prologue.
//   uniform depths  : Rn = -d1 * R,  Gn_j = -d1 * dR/dw_j   (d1 folded in on the host)
//   per-match depths: Rn = -R,       Gn_j = -dR/dw_j        (d1_i applied per match)
// so that  e = t + d2 x2 + [d1_i] Rn x1   and   A = d(e)/d(rot) = [d1_i] [Gn_0 x1 | Gn_1 x1 | Gn_2 x1].
struct SweepParams {
  double Rn[9];
  double Gn[27];   // Gn[9*j + 3*r + c]
  double t[3];
  double d2;       // uniform depth of the right point (unused with per-match depths)
  double delta;    // Huber a; <= 0: no robustifier
  double delta2;   // a*a
  unsigned long long n;  // correspondences in this shard
};
static_assert(sizeof(SweepParams) == 43 * 8, "SweepParams layout");

// Per-sweep state from (rot, tran, depths, delta).  depth_mode: 0 = uniform (d1 folded into Rn / Gn), 1 = per match.
// with_derivatives = false (the factored kernel never reads Gn): Gn is zeroed, d R / d w is not formed.
SBA_HD inline void fill_sweep_params(unsigned long long n, int depth_mode, const double rot[3], const double tran[3],
                                     double d1, double d2, double huber_delta, SweepParams* prm,
                                     bool with_derivatives = true) {
  double R[9], G[27];
  rotation_and_derivatives(rot, R, with_derivatives ? G : nullptr);
  const double scale = depth_mode == 0 ? -d1 : -1.0;
  for (int i = 0; i < 9; ++i) prm->Rn[i] = scale * R[i];
  for (int i = 0; i < 27; ++i) prm->Gn[i] = with_derivatives ? scale * G[i] : 0.0;
  for (int i = 0; i < 3; ++i) prm->t[i] = tran[i];
  prm->d2 = d2;
  prm->delta = huber_delta > 0.0 ? huber_delta : 0.0;
  prm->delta2 = prm->delta * prm->delta;
  prm->n = n;
}

// Device-resident correspondences as planes (structure of arrays): every lane of a wave reads
// 16 consecutive bytes of one plane, so each wave load instruction covers 1 KiB contiguous.
struct Planes {
  const void* x1[3];  // left unit vectors  x, y, z   (double or float per `store`)
  const void* x2[3];  // right unit vectors x, y, z
  const double* d1;   // per-match depths (always f64), may be null
  const double* d2;
};

constexpr int kPackSize = 24;
constexpr int kRow = 32;   // doubles per block-partial row (256 B: two whole 128-byte lines, 24 used)
#ifndef SBA_BLOCK
#define SBA_BLOCK 256
#endif
constexpr int kBlock = SBA_BLOCK;   // threads per sweep block (multiple of 64)

// Sweep: residual + Jacobian + Huber + reduction of every block's partial pack into
// partials[grid][24]; then finalize() folds the partials in a fixed order into pack_out[24].
// Where a sweep leaves its result.  ticket == nullptr: only partials[grid][kRow] are written and
// launch_finalize() must follow.  Otherwise the last-arriving block folds the partials into pack_dev[24] and, if
// pack_host != nullptr (device-visible address of mapped pinned memory, >= 26 doubles), also stores the pack there
// followed by `seq` at pack_host[24] (as a 64-bit integer) for the host to poll.  ticket: 9 counters, one per
// 64-byte line (ticket[16*k]), zero before the first launch; the kernel re-zeroes them.
struct SweepOut {
  double* partials;
  double* pack_dev;
  double* pack_host;
  unsigned int* ticket;
  unsigned long long seq;
};

// kind: 0 = factored (moment pack, host applies J_l), 1 = explicit per-match Jacobian (SBA_PACK layout).
hipError_t launch_sweep(int mode, int depth, int store, int kind, const Planes& pl, const SweepParams& prm,
                        const SweepOut& out, int grid, hipStream_t stream);
hipError_t sweep_blocks_per_cu(int mode, int depth, int store, int kind, bool loss, int* blocks);
// Direct peer exchange (all-reduce of the pack over IPC-mapped inboxes, see peer_exchange_kernel).
constexpr int kMaxPeers = 8;
constexpr size_t kInboxDoubles = 2 * kMaxPeers * 32;   // [parity][source rank][32] = 4 KiB per rank
struct PeerInboxes {
  double* inbox[kMaxPeers];   // device-visible address of every rank's inbox (index = rank, own included)
  int nranks;
  int rank;
  unsigned long long timeout_ticks;   // longest wait for a peer's slot, in ticks of the constant-rate wall clock
                                      // (wall_clock64, hipDeviceAttributeWallClockRate) -- time, not loop iterations
  unsigned long long* sticky;         // device word, set to 1 by any exchange that timed out and never cleared: a
                                      // later exchange of the same burst publishes it, so no timeout goes unseen
};
hipError_t launch_peer_exchange(const double* pack_local, const PeerInboxes& px, unsigned long long xseq,
                                double* pack_out, double* pack_host_dev, unsigned long long host_seq,
                                hipStream_t stream);
// Folds partials[nblocks][kRow] into pack_out[24].  px != nullptr: the same launch then all-reduces the pack across
// ranks (peer exchange, sequence number xseq).  pack_host_dev != nullptr: ... and publishes it to mapped pinned host
// memory followed by `seq` at [24] for the host to poll.
hipError_t launch_finalize(const double* partials, int nblocks, double* pack_out, double* pack_host_dev,
                           unsigned long long seq, const PeerInboxes* px, unsigned long long xseq,
                           hipStream_t stream);

// pack_dev[24] -> mapped pinned host memory, then `seq` at pack_host[24] (64-bit) for the host to poll.
hipError_t launch_publish(const double* pack_dev, double* pack_host_dev, unsigned long long seq,
                          hipStream_t stream);

// Batched problems (config C5): pair g owns vectors [first_vec, first_vec + ceil(n/PPT)) of the shared planes and
// params[g] (its own R|t; params[g].n = 0 skips the pair).  bpp blocks per pair; partials [num_pairs*bpp][kRow];
// packs [num_pairs][24] in the selected kernel's layout.
// Vector p of a pair (p = 0, 1, ...: 16 bytes of every coordinate plane) sits at plane vector
//     first_vec + (p / 256) * tile_stride + (p % 256):
// tile_stride = 256: the pair's vectors are contiguous; tile_stride = 256 * num_pairs, first_vec = 256 * pair: the pairs are
// INTERLEAVED in 4 KiB tiles (tile t of every pair side by side), so that blocks that sweep their own pairs in step with one
// another read one contiguous window of every plane -- the access pattern of the single-problem grid-stride sweep -- instead of
// num_pairs x 8 separate sequential streams.
struct PairDesc {
  unsigned long long first_vec;
  unsigned long long n;
  unsigned long long tile_stride;
  unsigned long long pad_;
};
constexpr unsigned long long kPairTile = 256;   // vectors per tile = threads per sweep block
SBA_HD inline unsigned long long pair_vector(const PairDesc& d, unsigned long long p) {
  return d.first_vec + (p / kPairTile) * d.tile_stride + (p % kPairTile);
}
// What the host hands over per pair and step (mapped pinned host memory, 80 B per pair): the point to evaluate at and
// the number of matches that take part (0 = pair already converged).
struct BatchState {
  double rot[3], tran[3];
  double d1, d2;
  unsigned long long n;
  unsigned long long pad_;
};
static_assert(sizeof(BatchState) == 80, "BatchState is copied word by word");
// The same step in ONE launch, for batches with one block per pair (bpp == 1): every block prepares its own pair's
// state, sweeps, folds, converts and publishes its pack; the last one stores `seq` at packs_host[24 * num_pairs].
// ticket: one zeroed device word.
hipError_t launch_batch_step_fused(int mode, int depth, int store, int kind, double huber_delta, const Planes& pl,
                                   const BatchState* state, const PairDesc* desc, int num_pairs, double* packs,
                                   double* packs_host, unsigned int* ticket, unsigned long long seq, hipStream_t stream);

// In / out record of one pair for the one-launch per-pair solve (batch_lm_kernel): start point and uniform depths in,
// result, summary and status out.
struct BatchLmIo {
  double rot[3], tran[3];
  double d1, d2;
  sba_lm_summary summary;
  int status;
  int pad_;
};
// ticket: one zeroed device word; seq_host_dev (may be null): device-visible address of a host word that receives `seq`
// once every pair's record has been written.
// sweep_cap / cont_*: with cont_state != nullptr the kernel runs at most sweep_cap sweeps per pair; a pair that needs more
// leaves its solver (cont_state[pair], batch_lm_dyn_state_bytes() each), the state of its next sweep (cont_params / cont_frames)
// and cont_done[pair] = 0 behind, and its record says pad_ = 1; the launches with dynamic shares below take over.
hipError_t launch_batch_lm(int mode, int depth, int store, int kind, const Planes& pl, const PairDesc* desc,
                           BatchLmIo* io, const sba_lm_options& opt, int num_pairs, unsigned int* ticket,
                           unsigned long long* seq_host_dev, unsigned long long seq, hipStream_t stream, int sweep_cap = 0,
                           void* cont_state = nullptr, SweepParams* cont_params = nullptr, double* cont_frames = nullptr,
                           int* cont_done = nullptr);
hipError_t batch_blocks_per_cu(int mode, int depth, int store, int kind, bool loss, int* blocks);
// Device-resident per-pair solves with DYNAMIC shares (sba_batch_kernels.hip, sba_depth.hip): every iteration / pass is a
// sweep launch whose blocks are dealt out to the pairs that are still iterating, a per-pair step launch and a compaction
// launch -- all enqueued without waiting; pairs that have converged hand their CUs to the others.
struct BatchDynCtl { unsigned int nactive[2]; unsigned int pad_[2]; };
// Shares per active pair of a launch of `grid` blocks: grid / nactive (1 when there are at least as many pairs as blocks).
// Measured alternative: ceil(4 * grid / nactive) shares, several work items per block -- no block idle at 129 ... 255 active
// pairs, but every item pays its own fold and row and the per-pair step folds more rows: rot-only stage 2.50 -> 3.28 ms, d-only
// 7.0 -> 7.5 ms at C5 on the same box (profiles/r03c_dyn_shares_ab.log).  Rows of a launch: nactive * shares <= grid + nactive.
constexpr unsigned kDynOver = 1;
SBA_HD inline unsigned dyn_shares(unsigned nactive, unsigned grid) { return nactive == 0u || nactive >= grid ? 1u : grid / nactive; }
size_t batch_lm_dyn_state_bytes();
hipError_t launch_batch_dyn_first_list(BatchDynCtl* ctl, unsigned int* active, const int* done, int num_pairs, hipStream_t stream);
hipError_t launch_batch_lm_dyn_init(int mode, int depth, int kind, const PairDesc* desc, const BatchLmIo* io, const sba_lm_options& opt,
                                    int num_pairs, void* state, SweepParams* params, double* frames, BatchDynCtl* ctl,
                                    unsigned int* active, int* done, hipStream_t stream);
hipError_t launch_batch_lm_dyn_pass(int mode, int depth, int store, int kind, const Planes& pl, const PairDesc* desc,
                                    const sba_lm_options& opt, int num_pairs, int parity, int sweep_grid, void* state,
                                    SweepParams* params, double* frames, BatchDynCtl* ctl, unsigned int* active, int* done,
                                    double* partials, BatchLmIo* io, unsigned long long* host_words, unsigned long long seq,
                                    hipStream_t stream);

hipError_t launch_batch_sweep(int mode, int depth, int store, int kind, bool loss, const Planes& pl,
                              const SweepParams* params, const PairDesc* desc, int num_pairs, int bpp,
                              double* partials, double* packs, double* packs_host, unsigned long long seq,
                              hipStream_t stream);
// Same, with the per-pair preparation and conversion on the device: batch_prepare_kernel builds every pair's
// SweepParams (and, for the factored kernel, the frame (B, J) of its rotation) from `state`; batch_finalize_kernel folds
// the rows and maps the factored kernel's moments to the SBA_PACK_* layout before publishing.
hipError_t launch_batch_sweep_only(int mode, int depth, int store, int kind, bool loss, const Planes& pl,
                                   const SweepParams* params, const PairDesc* desc, int num_pairs, int bpp,
                                   double* partials, hipStream_t stream);
hipError_t launch_batch_step(int mode, int depth, int store, int kind, double huber_delta, const Planes& pl,
                             const BatchState* state, SweepParams* params, double* frames, const PairDesc* desc,
                             int num_pairs, int bpp, double* partials, double* packs, double* packs_host,
                             unsigned long long seq, hipStream_t stream);

// AoS (cv::Point3d layout, double[3n]) -> planes, element offset `first`, count `n`.  tile_stride_elems != 0 (batched,
// interleaved layout): element i goes to first + (i / tile_elems) * tile_stride_elems + i % tile_elems.
hipError_t launch_aos_to_planes(const double* aos, size_t n, size_t first, void* px, void* py,
                                void* pz, int store, hipStream_t stream, size_t tile_elems = 0, size_t tile_stride_elems = 0);
// d12 (double[2n]) -> two f64 planes.
hipError_t launch_d12_to_planes(const double* d12, size_t n, size_t first, double* d1, double* d2,
                                hipStream_t stream, size_t tile_elems = 0, size_t tile_stride_elems = 0);
// A chunk of the CONCATENATED arrays of a batch (rows first_row .. first_row + m, relative to the batch's first row) to its
// places in the pairs' tiles, one launch: offsets_dev[num_pairs + 1] = first row of every pair (relative likewise).
hipError_t launch_batch_aos_to_planes(const double* aos, size_t m, size_t first_row, const unsigned long long* offsets_dev,
                                      int num_pairs, const PairDesc* desc, size_t ppt, void* px, void* py, void* pz, int store,
                                      hipStream_t stream);
hipError_t launch_batch_d12_to_planes(const double* d12, size_t m, size_t first_row, const unsigned long long* offsets_dev,
                                      int num_pairs, const PairDesc* desc, size_t ppt, double* d1, double* d2,
                                      hipStream_t stream);
hipError_t launch_planes_to_d12(const double* d1, const double* d2, size_t n, double* d12,
                                hipStream_t stream);

// d-only stage (spherical_bundle_adjuster.cpp:1004-1063): one LM iteration of the global bounded problem.
struct DepthParams {
  double R[9];
  double t[3];
  double lambda, c;
  double radius, inv_radius;
  double min_diagonal, max_diagonal;
  double alpha;          // line-search step size: the candidate is P(d + alpha * delta); 1.0 for the trust-region step
  int first_iteration;   // compute and store the Jacobi scaling
  int reuse_diagonal;    // previous step was rejected (or a line-search pass): Ceres keeps its LM diagonal -- the pass
                         // recomputes it at the same point (identical bits), nothing is stored
  int jacobi_scaling;
  int stream_stores;     // candidate planes stored non-temporally (1) or with plain stores (0)
  unsigned long long n;
};
// Results of one pass: the DEPTH_OUT_* slots of out / host_out (sba_depth_solver.hpp): seven sums and two maxima.
// gather_slot >= 0 (sharded problem): `out` is a 24-double pack for a SUM all-reduce, the sums in [0..6], this rank's
// two maxima in [8 + gather_slot] and [16 + gather_slot] (gather_slot < 8), zeros elsewhere; nothing is published to
// the host.  Candidates go to (c1, c2); (sc*) hold the per-parameter Jacobi scaling.  partials: [grid][16].
hipError_t depth_blocks_per_cu(int store, int* blocks);   // resident 256-thread blocks per CU of depth_step_kernel
hipError_t launch_depth_step(int store, const Planes& pl, const double* d1, const double* d2, double* c1,
                             double* c2, double* sc1, double* sc2,
                             const DepthParams& prm, double* partials, int grid, double* out, double* host_out,
                             unsigned long long seq, int gather_slot, hipStream_t stream);

// Batched d-only stage (one 512-thread block per pair; sba_depth.hip).  BatchDepthConst: the pair's frozen rotation matrix
// and translation (device array, set once per solve).  BatchDepthPass: what the host hands over per pair and pass (mapped
// pinned host memory, 32 B): radius and step size of the pass, flags (bit 0 first pass, 1 keep diagonal, 2 jacobi scaling,
// 3 flip: current depths in the work planes), n = matches taking part (0: pair finished).  Results: out_host[pair][16]
// (DEPTH_OUT_* slots) in mapped host memory, then `seq` at out_host[num_pairs * 16].
struct BatchDepthConst { double R[9]; double t[3]; };
struct BatchDepthPass { double radius, alpha; unsigned long long n; unsigned int flags, pad_; };
static_assert(sizeof(BatchDepthPass) == 32, "BatchDepthPass layout");
hipError_t launch_batch_depth_step(int store, const Planes& pl, const PairDesc* desc, const BatchDepthConst* cst,
                                   const BatchDepthPass* pass_host_dev, int num_pairs, double lambda, double c, double min_diagonal,
                                   double max_diagonal, double* a1, double* a2, double* b1, double* b2, double* sc1, double* sc2,
                                   double* out_host_dev, unsigned int* ticket, unsigned long long seq, hipStream_t stream);
// pairs with flip_dev[pair] != 0: work planes -> the batch's depth planes; out_dev != nullptr: every pair's depths in init_d
// layout at offsets_dev[pair]
// The whole d-only stage of every pair in one launch (batch_depth_solve_kernel: one DepthStageSolver per pair on the device).
// io: the mapped per-pair records of the batch (summary, status; pad_ = passes run); offsets_dev / out_dev as for
// launch_batch_depth_finish (out_dev may be null); seq_host_dev receives `seq` once every pair has delivered.
hipError_t launch_batch_depth_solve(int store, const Planes& pl, const PairDesc* desc, const BatchDepthConst* cst, int num_pairs,
                                    double lambda, double c, const sba_lm_options& opt, double* a1, double* a2, double* b1, double* b2,
                                    double* sc1, double* sc2, const unsigned long long* offsets_dev, double* out_dev, BatchLmIo* io,
                                    unsigned int* ticket, unsigned long long* seq_host_dev, unsigned long long seq,
                                    hipStream_t stream, int pass_cap = 0, void* cont_state = nullptr, BatchDepthPass* cont_req = nullptr,
                                    int* cont_done = nullptr, unsigned char* cont_finish = nullptr);
// passes a pair can need: per iteration the trust-region pass, up to max_num_line_search_step_size_iterations contractions and
// one pass that restores the full step; a few more detect the iteration limit
inline int batch_depth_pass_bound(const sba_lm_options& opt) {
  const long long its = opt.max_num_iterations > 0 ? opt.max_num_iterations : 0;
  const long long lsn = opt.max_num_line_search_step_size_iterations > 0 ? opt.max_num_line_search_step_size_iterations : 0;
  long long bound = its * (lsn + 2) + 4;
  if (bound > (1ll << 24)) bound = 1ll << 24;
  return static_cast<int>(bound);
}
// With cont_state != nullptr (batch_depth_dyn_state_bytes() per pair) the kernel runs at most pass_cap passes per pair; a pair
// that needs more leaves its solver and next request behind (record.status == 1, cont_done[pair] = 0) for the passes with
// dynamic shares: launch_batch_depth_dyn_pass + launch_batch_dyn_compact per pass.  cont_finish[pair] tells
// launch_batch_depth_finish what is left to do for the pair (bit 0 copy back, bit 1 write out).
size_t batch_depth_dyn_state_bytes();
hipError_t launch_batch_depth_dyn_pass(int store, const Planes& pl, const PairDesc* desc, const BatchDepthConst* cst, int num_pairs,
                                       double lambda, double c, const sba_lm_options& opt, double* a1, double* a2, double* b1, double* b2,
                                       double* sc1, double* sc2, int parity, int sweep_grid, void* state, BatchDepthPass* req,
                                       const BatchDynCtl* ctl, const unsigned int* active, int* done, unsigned char* finish,
                                       double* partials, BatchLmIo* io, hipStream_t stream);
hipError_t launch_batch_dyn_compact(BatchDynCtl* ctl, unsigned int* active, const int* done, int num_pairs, int parity,
                                    unsigned long long* host_words, unsigned long long seq, hipStream_t stream);
hipError_t launch_batch_depth_finish(int store, const PairDesc* desc, const unsigned char* flip_dev, int num_pairs, double* a1,
                                     double* a2, const double* b1, const double* b2, const unsigned long long* offsets_dev,
                                     double* out_dev, hipStream_t stream);

// 8-point initial guess, device part (.cpp:53-68): A^T A of the kron(left, right) rows for 64 interleaved groups.
// groups_dev: [64][45]; partials: [grid][45][64] scratch.
hipError_t launch_epipolar_moments(int store, const Planes& pl, size_t n, double* partials, int grid,
                                   double* groups_dev, hipStream_t stream);

// The 64 x 45 group moments of every pair of a batch: groups_dev[num_pairs][64][45], one block per pair.
hipError_t launch_batch_epipolar_moments(int store, const Planes& pl, const PairDesc* desc, int num_pairs, double* groups_dev,
                                         hipStream_t stream);

// The trials and the consensus pick of every pair on the device (batch_guess_kernel): one block per pair reads the pair's
// group moments and writes its record.  trials <= kGuessMaxTrials.
constexpr int kGuessMaxTrials = 128;
struct BatchGuessOut { double euler[3]; double tran[3]; int num_candidates; int status; };
static_assert(sizeof(BatchGuessOut) == 56, "BatchGuessOut layout");
hipError_t launch_batch_guess(const double* groups_dev, int num_pairs, int trials, double fraction, unsigned long long seed,
                              BatchGuessOut* out_dev, hipStream_t stream);

// ... the same A^T A per TRIAL over explicit index lists (the reference's own random subsets, .cpp:130-141):
// indices_dev [trials][m] int32, moments_dev [trials][45].
hipError_t launch_epipolar_subset_moments(int store, const Planes& pl, size_t n, const int* indices_dev, int trials, int m,
                                          double* moments_dev, hipStream_t stream);

// pixel -> unit sphere (spherical_bundle_adjuster.cpp:271-298)
hipError_t launch_keypoints_to_sphere(const uint8_t* kp, size_t n, size_t stride_bytes, double im_w,
                                      double im_h, double* out_xyz, hipStream_t stream);
// ... fused with the upload: matched key-point records of both images -> the six coordinate planes (x1.xyz, x2.xyz)
hipError_t launch_keypoints_to_planes(const uint8_t* kp_left, const uint8_t* kp_right, size_t n, size_t stride_bytes,
                                      double im_w, double im_h, void* const planes[6], int store, hipStream_t stream);
int points_per_lane(int store);  // 2 for f64 planes, 4 for f32 planes

}  // namespace sba
