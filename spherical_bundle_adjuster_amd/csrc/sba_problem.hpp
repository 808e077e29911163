// The problem handle behind the C-ABI and the helpers shared by the translation units that implement it:
//   sba_shim.cpp       errors, handle life cycle, uploads, sweeps, eval / solve entry points, side entry points
//   sba_transport.cpp  multi-GPU transports: RCCL (bound at run time), direct peer exchange, user hook; pack all-reduce
//   sba_stages.cpp     d-only stage and 8-point initial guess entry points
// Internal: nothing here is exported from the library.
#pragma once
#include <cstdlib>
#include <hip/hip_runtime.h>

#include <string>

#include "../../include/sba_hip.h"
#include "sba_device.hpp"
#include "sba_internal.hpp"
#include "sba_resident.hpp"

// ---- the handle ---------------------------------------------------------------------------------
struct sba_problem {
  int device = 0;
  hipStream_t stream = nullptr;
  bool own_stream = false;
  int poisoned = 0;            // a device wait on `stream` timed out or the device faulted (sba_internal.hpp): every entry
                               // point refuses the handle from then on, destroy leaks the device resources
  int num_cus = 0;
  int blocks_per_cu_cap = 0;   // SBA_BLOCKS_PER_CU: resident blocks per CU used; 0 = the per-variant default of grid_for
                               // (1 or 2: the register double buffer supplies the memory-level parallelism, more
                               // waves only add rows to fold and finish-time spread, profiles/r01_tune_caps.log)
  int kind = SBA_KERNEL_FACTORED;
  int occ_cache[3][2][2][2][2];  // resident blocks/CU per [mode][depth][store][kind][loss], 0 = unknown
  int depth_occ[2] = {0, 0};     // same for depth_step_kernel per [store]
  double* epi_scratch = nullptr; // 8-point moments: [grid][45][64] block partials + [64][45] groups, kept across calls
  size_t epi_scratch_elems = 0;
  void* upload_pinned[2] = {nullptr, nullptr};   // large uploads: two pinned staging buffers (sba_shim.cpp: upload_common)
  void* depth_scratch = nullptr;   // d-only stage: candidate + scaling planes, block partials, results; kept across calls
  size_t depth_scratch_bytes = 0;
  void* subset_scratch = nullptr;  // reference sampling: [trials][45] moments, then the [trials][m] index lists; kept across calls
  size_t subset_scratch_bytes = 0;
  double frame_B[9], frame_J[9];  // factored kernel: host-side frame of the last enqueued sweep
  int last_mode = 0;

  size_t n = 0;
  int store = SBA_STORE_F64;
  bool has_d12 = false;
  bool uploaded = false;
  size_t plane_elems = 0;     // allocated elements per plane (n rounded up to a whole vector)
  void* coord[6] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
  double* dplane[2] = {nullptr, nullptr};
  void* plane_base[8] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};  // hipMalloc'ed blocks
  size_t plane_bytes[8] = {0, 0, 0, 0, 0, 0, 0, 0};  // their sizes: an upload that fits reuses them (a hipFree + hipMalloc pair per
                                                      // plane costs more than the whole upload of a 2 000-match problem)
  size_t plane_stagger = 4352; // SBA_PLANE_STAGGER: plane k starts k * 4352 B (17 x 256 B) into its allocation, so equal
                               // element indices of the 8 streams differ in their low address bits (measured 0-4 %
                               // faster with f64 planes, 3-5 % with f32 planes; never slower)

  double* partials = nullptr;  // [max_grid][24]
  int max_grid = 0;
  double* pack_dev = nullptr;  // 32 doubles
  double* pack_host = nullptr; // pinned + mapped, 32 doubles ([24] = sequence number published by the kernel)
  double* pack_host_dev = nullptr;  // device-visible address of pack_host
  unsigned int* ticket = nullptr;   // arrival counter of the fused final reduction
  unsigned long long seq = 0;       // sweeps launched with host publication
  int fused_mode = 0;               // SBA_FUSED: 0 (default) never, 1 always, 2 only for grids <= kFusedMaxGrid
  bool publish = true;              // SBA_PUBLISH=0: D2H copy + stream sync instead of kernel-side publication
  bool published = false;           // the last enqueued sweep publishes to pack_host itself
  hipEvent_t ev0 = nullptr, ev1 = nullptr;

  // direct peer exchange (IPC-mapped inboxes, sba_problem_peer_*)
  double* inbox = nullptr;
  sba::PeerInboxes peers{};
  void* peer_opened[sba::kMaxPeers] = {nullptr};
  bool peer_ready = false;
  unsigned long long xseq = 0;
  double peer_timeout_s = 20.0;           // longest wait for a peer's slot before SBA_ERR_COMM (SBA_PEER_TIMEOUT_S)
  int wall_clock_khz = 100000;            // rate of the device's constant wall clock (hipDeviceAttributeWallClockRate)
  unsigned long long* peer_sticky = nullptr;   // device word: some exchange timed out (never cleared while connected)

  // resident evaluator for small problems (sba_resident.hpp): command record in mapped pinned host memory
  sba::ResidentRecord* res_rec = nullptr;
  sba::ResidentRecord* res_rec_dev = nullptr;
  // One-launch stages of a small problem (a batch of one pair through batch_depth_solve_kernel / batch_lm_kernel): the pair
  // descriptor, the d-only stage's R | t, the in / out record and the completion word, in ONE mapped host block the kernel
  // reads / writes directly (sba_shim.cpp: SmallRecord).
  void* small_rec = nullptr;
  void* small_rec_dev = nullptr;
  unsigned long long small_seq = 0;
  unsigned long long res_cmd_seq = 0;     // sequence number of the last command written
  // Largest problem the LM stages / the d-only stage drive through a resident single-block kernel; above it every sweep /
  // pass is a launch.  Measured cross-over on MI355X (profiles/r03_c1_pipeline.md): one block sweeps 4 096 matches in less
  // time than two launches cost, but the d-only pass is arithmetic-heavy (4 exp + a reciprocal per match) and one CU falls
  // behind the 4-16 CUs a launch spreads it over from ~3 000 matches on.  SBA_RESIDENT_MAX_N overrides both (0 = never).
  size_t resident_max_n = 4096;
  size_t resident_max_n_depth = 2560;
  size_t one_launch_max_n_depth = 4096;     // d-only stage as ONE launch (batch of one pair): 331 us against 362 us with a launch per pass at
                                            // 4 096 matches, 405 against 344 at 6 144 (tools/small_depth_sizes.py); SBA_RESIDENT_MAX_N sets it too
  double resident_idle_s = 0.25;          // SBA_RESIDENT_IDLE_S: a resident kernel ends itself after this long without a command

  sba_allreduce_fn hook = nullptr;
  void* hook_user = nullptr;
  void* comm = nullptr;        // ncclComm_t
  int nranks = 1;
  int shard_rank = 0, shard_count = 1;   // which shard of the correspondences this problem holds (d-only stage)
};

namespace sba {
namespace shim {

// ---- RCCL, bound at run time -------------------------------------------------------------------
// The library is dlopen'ed instead of linked so that a process which already carries an RCCL
// (torch ships its own librccl.so.1) keeps exactly one copy.
struct Rccl {
  typedef struct { char internal[SBA_COMM_ID_BYTES]; } UniqueId;
  int (*GetUniqueId)(UniqueId*) = nullptr;
  int (*CommInitRank)(void**, int, UniqueId, int) = nullptr;
  int (*CommDestroy)(void*) = nullptr;
  int (*AllReduce)(const void*, void*, size_t, int, int, void*, hipStream_t) = nullptr;
  const char* (*GetErrorString)(int) = nullptr;
  void* handle = nullptr;
  bool ok = false;
  std::string why;
};
Rccl& rccl();
constexpr int kNcclFloat64 = 8;  // ncclDouble
constexpr int kNcclSum = 0;      // ncclSum

inline bool is_collective(const sba_problem* p) { return p->comm != nullptr || p->hook != nullptr || p->peer_ready; }

// sba_resident.cpp -- one resident kernel serving one solve stage of a small, unsharded problem.  start_*() launches it,
// call() sends a command and waits for its answer (restarting a kernel that ended itself: idle time-out, trip limit),
// end() sends QUIT and waits, bounded, for the stream to drain; the destructor ends a session that is still open.
bool resident_eligible(const sba_problem* p, bool depth_stage);
// The mapped host block of the one-launch stages (sba_problem::small_rec): what the batch kernels read per pair -- descriptor,
// frozen R | t of the d-only stage, the in / out record -- and the completion word they publish.
struct SmallRecord {
  sba::PairDesc desc;
  sba::BatchDepthConst cst;
  sba::BatchLmIo io;
  volatile unsigned long long seq;
  unsigned long long pad_[7];
};
// Which stages of a small problem run as ONE launch with the solver on the device.  Default: the d-only stage (its step
// logic is a few microseconds on a GPU lane: 302 us against 376 us through the resident evaluator at 2 048 matches), NOT the LM
// stages (LmSolver::feed costs 11-16 us on one lane, more than the host's step plus the command round trip: 127 / 89 us against
// 65 / 47 us for rot-only / tran-only).  SBA_SMALL_ONE_LAUNCH=0: no stage; =2: the LM stages too (A/B, tests).
inline bool small_one_launch(bool depth_stage) {
  const char* env = std::getenv("SBA_SMALL_ONE_LAUNCH");
  if (env && env[0] == '0') return false;
  if (env && env[0] == '2') return true;
  return depth_stage;
}
class ResidentSession {
 public:
  explicit ResidentSession(sba_problem* p) : p_(p) {}
  ResidentSession(const ResidentSession&) = delete;
  ResidentSession& operator=(const ResidentSession&) = delete;
  ~ResidentSession() { if (active_) (void)end(); }
  int start_sweep(int mode, int depth_mode, bool loss);
  int start_depth(double* a1, double* a2, double* b1, double* b2, double* sc1, double* sc2);
  int call(const double* payload, int count, double* out, int out_count);
  int end();
  bool active() const { return active_; }

 private:
  int launch();
  sba_problem* p_;
  bool active_ = false;
  bool depth_ = false;
  int mode_ = 0, depth_mode_ = 0;
  bool loss_ = true;
  double *a1_ = nullptr, *a2_ = nullptr, *b1_ = nullptr, *b2_ = nullptr, *sc1_ = nullptr, *sc2_ = nullptr;
  unsigned long long pending_cmd_ = 0, pending_pack_ = 0;   // what the next launched kernel waits for / answers with
};

// sba_transport.cpp
int allreduce_pack(sba_problem* p);                                  // p->pack_dev (24 doubles), then hand-over to the host
int allreduce_buffer(sba_problem* p, double* dev, size_t count);     // any device buffer, count a multiple of 24
// sba_shim.cpp
int fetch_pack_raw(sba_problem* p, double raw[SBA_PACK_SIZE]);       // wait for the pack handed over by the last sweep / pass

}  // namespace shim
}  // namespace sba
