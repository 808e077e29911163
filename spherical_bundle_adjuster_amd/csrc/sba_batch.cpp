// Batched per-pair solve (BASELINE config C5: "256 ERP pairs x 50k matches each, per-pair LM"): the reference
// would run main/main.cpp once per pair; here all pairs of a batch live in one set of planes, ONE launch
// evaluates every pair at its own R|t, and one host LmSolver per pair advances in lock-step off that launch.
// Pairs are independent: across GPUs they shard without any collective.
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <new>
#include <thread>
#include <vector>

#include "sba_depth_solver.hpp"
#include "sba_epipolar.hpp"
#include "sba_internal.hpp"
#include "sba_lm.hpp"
#include "sba_rotation.hpp"

struct sba_batch {
  int device = 0;
  hipStream_t stream = nullptr;
  bool own_stream = false;
  int poisoned = 0;            // a device wait timed out or the device faulted (sba_internal.hpp): the handle is refused
                               // from then on -- its ticket word and sequence numbers are out of step -- destroy leaks
  int num_cus = 0;
  int kind = SBA_KERNEL_FACTORED;

  int num_pairs = 0;
  int store = SBA_STORE_F64;
  bool has_d12 = false;
  bool uploaded = false;
  std::vector<size_t> n;            // matches per pair
  std::vector<size_t> first_vec;    // first 16-byte vector of the pair inside the planes
  size_t tile_stride = 256;         // vectors between consecutive 256-vector tiles of a pair (256 = contiguous pairs; sba_device.hpp: PairDesc)
  size_t total_vecs = 0;
  void* coord[6] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
  double* dplane[2] = {nullptr, nullptr};
  void* plane_base[8] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};   // hipMalloc'ed blocks
  size_t plane_stagger = 4352;   // plane k starts k * 4352 B into its allocation (as sba_problem: equal element indices of
                                 // the 8 streams then differ in their low address bits); SBA_PLANE_STAGGER overrides
  sba::PairDesc* desc_dev = nullptr;
  sba::SweepParams* params_dev = nullptr;    // built on the device by batch_prepare_kernel
  sba::BatchState* state_host = nullptr;     // pinned + mapped: what the host hands over per pair and step (80 B)
  sba::BatchState* state_host_dev = nullptr; // device-visible address of state_host
  double* frames_dev = nullptr;              // [pair][18]: (B, J) of the pair's rotation, for the moment conversion
  double* partials = nullptr;
  int bpp = 1;                                // blocks per pair
  double* packs_dev = nullptr;
  double* packs_host = nullptr;               // pinned + mapped: 24 doubles per pair, then the sequence word
  double* packs_host_dev = nullptr;           // device-visible address of packs_host
  unsigned long long seq = 0;                 // launches published so far
  bool publish = true;                        // SBA_PUBLISH=0: D2H copy + stream synchronise instead
  sba::BatchLmIo* lm_io_host = nullptr;       // pinned + mapped: per-pair start point in, result + summary out (batch_lm_kernel)
  sba::BatchLmIo* lm_io_host_dev = nullptr;   // device-visible address of lm_io_host
  unsigned int* lm_ticket = nullptr;          // device: blocks of batch_lm_kernel that have delivered their record
  std::vector<size_t> offsets;                // row offset of every pair in the caller's concatenated arrays (num_pairs + 1)
  size_t plane_elems = 0;                     // elements per plane
  // batched d-only stage (allocated on first use, kept while the batch lives)
  double* depth_work = nullptr;               // 4 planes: candidate depths (2), Jacobi scaling (2)
  sba::BatchDepthConst* depth_const_dev = nullptr;
  sba::BatchDepthPass* depth_pass_host = nullptr;     // pinned + mapped
  sba::BatchDepthPass* depth_pass_host_dev = nullptr;
  double* depth_out_host = nullptr;           // pinned + mapped: [num_pairs][16] results, then the sequence word
  double* depth_out_host_dev = nullptr;
  unsigned long long depth_seq = 0;
  // batched initial guess (allocated on first use): the 64 x 45 group moments of every pair
  double* epi_groups_dev = nullptr;
  double* epi_groups_host = nullptr;          // pinned
  sba::BatchGuessOut* guess_out_dev = nullptr;
  sba::BatchGuessOut* guess_out_host = nullptr;   // pinned
  // device-resident solves with dynamic shares (allocated on first use)
  sba::BatchDynCtl* dyn_ctl = nullptr;
  unsigned int* dyn_active = nullptr;        // [2][num_pairs]
  int* dyn_done = nullptr;                   // [num_pairs]
  void* dyn_state = nullptr;                 // per-pair solver state
  double* dyn_partials = nullptr;            // share rows of one launch
  size_t dyn_partial_rows = 0;
  unsigned long long* dyn_host = nullptr;    // pinned + mapped: [0] pairs still active, [1] sequence word
  unsigned long long* dyn_host_dev = nullptr;
  unsigned long long dyn_seq = 0;
  sba::BatchDepthPass* dyn_depth_req = nullptr;   // [num_pairs]: the next pass of every pair of the d-only stage
  unsigned char* dyn_finish = nullptr;            // [num_pairs]: what batch_depth_finish_kernel has left to do
  // upload: row offsets on the device (relative to the first row), two pinned staging buffers, their DMA-done events
  unsigned long long* offsets_dev = nullptr;
  void* upload_pinned[2] = {nullptr, nullptr};
  hipEvent_t upload_ev[2] = {nullptr, nullptr};
};

namespace {

int free_batch_data(sba_batch* b) {
  for (auto& pb : b->plane_base) { if (pb) SBA_TRY_HIP(hipFree(pb)); pb = nullptr; }
  for (auto& c : b->coord) c = nullptr;
  for (auto& d : b->dplane) d = nullptr;
  if (b->desc_dev) SBA_TRY_HIP(hipFree(b->desc_dev));
  if (b->params_dev) SBA_TRY_HIP(hipFree(b->params_dev));
  if (b->state_host) SBA_TRY_HIP(hipHostFree(b->state_host));
  if (b->frames_dev) SBA_TRY_HIP(hipFree(b->frames_dev));
  if (b->partials) SBA_TRY_HIP(hipFree(b->partials));
  if (b->packs_dev) SBA_TRY_HIP(hipFree(b->packs_dev));
  if (b->packs_host) SBA_TRY_HIP(hipHostFree(b->packs_host));
  if (b->lm_io_host) SBA_TRY_HIP(hipHostFree(b->lm_io_host));
  if (b->lm_ticket) SBA_TRY_HIP(hipFree(b->lm_ticket));
  if (b->depth_work) SBA_TRY_HIP(hipFree(b->depth_work));
  if (b->depth_const_dev) SBA_TRY_HIP(hipFree(b->depth_const_dev));
  if (b->depth_pass_host) SBA_TRY_HIP(hipHostFree(b->depth_pass_host));
  if (b->depth_out_host) SBA_TRY_HIP(hipHostFree(b->depth_out_host));
  if (b->epi_groups_dev) SBA_TRY_HIP(hipFree(b->epi_groups_dev));
  if (b->epi_groups_host) SBA_TRY_HIP(hipHostFree(b->epi_groups_host));
  if (b->offsets_dev) SBA_TRY_HIP(hipFree(b->offsets_dev));
  b->offsets_dev = nullptr;
  if (b->dyn_ctl) SBA_TRY_HIP(hipFree(b->dyn_ctl));       // one allocation: everything below is carved from it
  b->dyn_ctl = nullptr; b->dyn_active = nullptr; b->dyn_done = nullptr; b->dyn_state = nullptr; b->dyn_depth_req = nullptr; b->dyn_finish = nullptr;
  b->dyn_partials = nullptr; b->dyn_partial_rows = 0; b->dyn_host = nullptr; b->dyn_host_dev = nullptr; b->dyn_seq = 0;
  if (b->guess_out_dev) SBA_TRY_HIP(hipFree(b->guess_out_dev));
  if (b->guess_out_host) SBA_TRY_HIP(hipHostFree(b->guess_out_host));
  b->epi_groups_dev = nullptr; b->epi_groups_host = nullptr; b->guess_out_dev = nullptr; b->guess_out_host = nullptr;
  b->depth_work = nullptr; b->depth_const_dev = nullptr; b->depth_pass_host = nullptr; b->depth_pass_host_dev = nullptr;
  b->depth_out_host = nullptr; b->depth_out_host_dev = nullptr; b->depth_seq = 0; b->offsets.clear(); b->plane_elems = 0;
  b->lm_io_host = nullptr; b->lm_io_host_dev = nullptr; b->lm_ticket = nullptr;
  b->desc_dev = nullptr; b->params_dev = nullptr; b->state_host = nullptr; b->state_host_dev = nullptr;
  b->frames_dev = nullptr; b->partials = nullptr;
  b->packs_dev = nullptr; b->packs_host = nullptr; b->packs_host_dev = nullptr;
  b->uploaded = false; b->num_pairs = 0; b->n.clear(); b->first_vec.clear();
  return SBA_OK;
}

int check_batch_args(const sba_batch* b, int mode, int depth_mode, const double* rot, const double* tran) {
  if (!b) return sba::set_error(SBA_ERR_INVALID_ARG, "null batch handle");
  SBA_REFUSE_POISONED(b);
  if (!b->uploaded) return sba::set_error(SBA_ERR_NOT_UPLOADED, "no pairs uploaded");
  if (mode < SBA_MODE_ROT || mode > SBA_MODE_RT) return sba::set_error(SBA_ERR_INVALID_ARG, "bad mode %d", mode);
  if (depth_mode != SBA_DEPTH_UNIFORM && depth_mode != SBA_DEPTH_PER_MATCH)
    return sba::set_error(SBA_ERR_INVALID_ARG, "bad depth_mode %d", depth_mode);
  if (depth_mode == SBA_DEPTH_PER_MATCH && !b->has_d12)
    return sba::set_error(SBA_ERR_INVALID_ARG, "per-match depths requested but none were uploaded");
  if (b->num_pairs > 0 && (!rot || !tran)) return sba::set_error(SBA_ERR_INVALID_ARG, "rot/tran must not be null");
  return SBA_OK;
}

// One batched step: pair g is evaluated at (rot[g], tran[g]) unless active[g] == 0.  The host writes 80 bytes per pair
// into mapped pinned memory; batch_prepare_kernel builds every pair's sweep state from it ON THE DEVICE, the sweep runs,
// and the finalize kernel folds the rows, maps the factored kernel's moments to the SBA_PACK_* layout (also on the
// device) and publishes all packs into mapped host memory followed by a sequence word the host polls.  packs_host then
// holds the final packs.
bool step_is_fused(const sba_batch* b) {
  const char* e = std::getenv("SBA_BATCH_FUSED_STEP");      // read per call: tests compare both paths in one process
  return !(e && e[0] == '0') && b->bpp == 1 && b->publish;
}
int batch_launch(sba_batch* b, int mode, int depth_mode, const double* rot, const double* tran, const double* d1,
                 const double* d2, double huber_delta, const unsigned char* active, double* prepare_ms = nullptr,
                 bool allow_fused = true) {
  const int B = b->num_pairs;
  const auto t_prep = std::chrono::steady_clock::now();
  for (int g = 0; g < B; ++g) {
    sba::BatchState& st = b->state_host[g];
    for (int a = 0; a < 3; ++a) { st.rot[a] = rot[3 * g + a]; st.tran[a] = tran[3 * g + a]; }
    st.d1 = d1 ? d1[g] : 1.0;
    st.d2 = d2 ? d2[g] : 1.0;
    st.n = (!active || active[g]) ? b->n[g] : 0;
    st.pad_ = 0;
  }
  if (prepare_ms)
    *prepare_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_prep).count();
  sba::Planes pl;
  for (int k = 0; k < 3; ++k) { pl.x1[k] = b->coord[k]; pl.x2[k] = b->coord[3 + k]; }
  pl.d1 = b->dplane[0]; pl.d2 = b->dplane[1];
  // One block per pair (config C5): the whole step is ONE launch -- every block prepares its own pair's sweep state,
  // sweeps, folds, converts its moments and publishes its pack (batch_step_kernel).  SBA_BATCH_FUSED_STEP=0 keeps the
  // three-kernel chain, which also serves batches that spread a pair over several blocks.
  if (allow_fused && step_is_fused(b)) {
    const unsigned long long seq = ++b->seq;
    SBA_TRY_HIP(sba::launch_batch_step_fused(mode, depth_mode, b->store, b->kind, huber_delta, pl, b->state_host_dev,
                                             b->desc_dev, B, b->packs_dev, b->packs_host_dev, b->lm_ticket, seq, b->stream));
    return sba::wait_for_sequence(reinterpret_cast<volatile unsigned long long*>(b->packs_host + 24 * B), seq, b->stream,
                                  "batched step", &b->poisoned);
  }
  if (!b->publish) {
    SBA_TRY_HIP(sba::launch_batch_step(mode, depth_mode, b->store, b->kind, huber_delta, pl, b->state_host_dev,
                                       b->params_dev, b->frames_dev, b->desc_dev, B, b->bpp, b->partials, b->packs_dev,
                                       nullptr, 0, b->stream));
    SBA_TRY_HIP(hipMemcpyAsync(b->packs_host, b->packs_dev, sizeof(double) * 24 * B, hipMemcpyDeviceToHost, b->stream));
    { const int _rc = sba::stream_wait(b->stream, "stream synchronisation", &b->poisoned); if (_rc) return _rc; }
    return SBA_OK;
  }
  const unsigned long long seq = ++b->seq;
  SBA_TRY_HIP(sba::launch_batch_step(mode, depth_mode, b->store, b->kind, huber_delta, pl, b->state_host_dev,
                                     b->params_dev, b->frames_dev, b->desc_dev, B, b->bpp, b->partials, b->packs_dev,
                                     b->packs_host_dev, seq, b->stream));
  return sba::wait_for_sequence(reinterpret_cast<volatile unsigned long long*>(b->packs_host + 24 * B), seq, b->stream,
                                "batched sweep", &b->poisoned);
}

// The packs arrive in the SBA_PACK_* layout (the finalize kernel converts the factored kernel's moments on the device).
void convert_pack(const sba_batch*, int, const double*, const double* raw, double* pack) {
  std::memcpy(pack, raw, sizeof(double) * 24);
}

// Host threads for the per-pair LM steps of sba_batch_solve (sba_set_host_threads, the reference's set_omp): the pool
// lives for one solve; workers spin on a generation counter between steps (a step is ~0.2 ms, parking them would cost
// more than it saves), each owns a fixed contiguous range of pairs, so the result does not depend on the count.
class PairWorkers {
 public:
  template <typename Fn>
  PairWorkers(int threads, int pairs, Fn fn) : pairs_(pairs) {
    // a pair's LM step is ~0.5 us and starting a thread costs ~50 us per solve: at least 64 pairs per thread
    // (256 pairs x 50k: 1.15 ms with 1 thread, 1.07 ms with 4, 1.38 ms when forced to 8)
    nt_ = std::max(1, std::min(threads, pairs / 64));
    fn_ = fn;
    for (int t = 1; t < nt_; ++t) pool_.emplace_back([this, t] { loop(t); });
  }
  ~PairWorkers() {
    quit_.store(true, std::memory_order_release);
    gen_.fetch_add(1, std::memory_order_acq_rel);
    for (auto& th : pool_) th.join();
  }
  // runs fn(first, last) over all pairs, the calling thread taking the first range; returns when all ranges are done
  void run() {
    done_.store(0, std::memory_order_relaxed);
    gen_.fetch_add(1, std::memory_order_acq_rel);
    fn_(0, range_end(0));
    while (done_.load(std::memory_order_acquire) != nt_ - 1) __builtin_ia32_pause();
  }

 private:
  int range_end(int t) const { return static_cast<int>(static_cast<long long>(pairs_) * (t + 1) / nt_); }
  void loop(int t) {
    unsigned long long seen = 0;
    for (;;) {
      while (gen_.load(std::memory_order_acquire) == seen) __builtin_ia32_pause();
      ++seen;
      if (quit_.load(std::memory_order_acquire)) return;
      fn_(range_end(t - 1), range_end(t));
      done_.fetch_add(1, std::memory_order_acq_rel);
    }
  }
  int pairs_, nt_ = 1;
  std::function<void(int, int)> fn_;
  std::vector<std::thread> pool_;
  std::atomic<unsigned long long> gen_{0};
  std::atomic<int> done_{0};
  std::atomic<bool> quit_{false};
};

// Buffers of the device-resident solves with dynamic shares: ONE device allocation, carved (control words, the two active lists,
// done flags, finish flags, the d-only stage's requests, per-pair solver state sized for the larger of the two solvers, the share
// rows of one launch); the two host words the compaction kernel publishes to are the spare words behind the packs' sequence word
// (no further mapped allocation: the first solve of a fresh batch pays one hipMalloc).  max_rows: share rows of one launch.
int ensure_dyn(sba_batch* b, size_t max_rows) {
  const size_t B = static_cast<size_t>(b->num_pairs);
  if (b->dyn_ctl && b->dyn_partial_rows >= max_rows) return SBA_OK;
  if (b->dyn_ctl) SBA_TRY_HIP(hipFree(b->dyn_ctl));
  b->dyn_ctl = nullptr;
  auto up = [](size_t v) { return (v + 255) & ~size_t(255); };
  const size_t state_each = std::max(sba::batch_lm_dyn_state_bytes(), sba::batch_depth_dyn_state_bytes());
  const size_t o_active = up(sizeof(sba::BatchDynCtl)), o_done = o_active + up(sizeof(unsigned int) * 2 * B), o_finish = o_done + up(sizeof(int) * B),
               o_req = o_finish + up(B), o_state = o_req + up(sizeof(sba::BatchDepthPass) * B), o_rows = o_state + up(state_each * B),
               bytes = o_rows + up(max_rows * sba::kRow * sizeof(double));
  char* base = nullptr;
  SBA_TRY_HIP(hipMalloc(reinterpret_cast<void**>(&base), bytes));
  SBA_TRY_HIP(hipMemsetAsync(base, 0, o_state, b->stream));          // stream-ordered: never a blocking call here
  b->dyn_ctl = reinterpret_cast<sba::BatchDynCtl*>(base);
  b->dyn_active = reinterpret_cast<unsigned int*>(base + o_active);
  b->dyn_done = reinterpret_cast<int*>(base + o_done);
  b->dyn_finish = reinterpret_cast<unsigned char*>(base + o_finish);
  b->dyn_depth_req = reinterpret_cast<sba::BatchDepthPass*>(base + o_req);
  b->dyn_state = base + o_state;
  b->dyn_partials = reinterpret_cast<double*>(base + o_rows);
  b->dyn_partial_rows = max_rows;
  b->dyn_host = reinterpret_cast<unsigned long long*>(b->packs_host + 24 * B) + 2;       // words 2, 3 behind the packs (0: their sequence word)
  b->dyn_host_dev = reinterpret_cast<unsigned long long*>(b->packs_host_dev + 24 * B) + 2;
  b->dyn_host[0] = 0; b->dyn_host[1] = 0;
  b->dyn_seq = 0;
  return SBA_OK;
}

// Host arrays -> planes.  The caller's arrays are pageable (std::vector<cv::Point3d>::data() of many image pairs, or numpy):
// as for a large single problem (sba_shim.cpp: upload_common) a few host threads copy chunk k + 1 into one of two pinned
// staging buffers while the DMA engine moves chunk k and ONE re-layout launch (batch_aos_to_planes_kernel: every row finds
// its pair by bisection) turns chunk k - 1 into planes.  Chunks are 16 MiB of the CONCATENATED arrays, whatever the pair
// boundaries.  which: bit 0 left, bit 1 right, bit 2 d12.
constexpr size_t kBatchUploadChunk = 699050;       // rows per chunk: 16 MiB of xyz
int batch_transfer(sba_batch* b, const double* left_xyz, const double* right_xyz, const double* d12, unsigned which) {
  const int B = b->num_pairs;
  const size_t base = b->offsets.front(), total = b->offsets.back() - base;
  if (total == 0) return SBA_OK;
  const size_t ppt = static_cast<size_t>(sba::points_per_lane(b->store));
  const size_t chunk = std::min(kBatchUploadChunk, total), chunk_bytes = chunk * 3 * sizeof(double);
  for (int k = 0; k < 2; ++k) {
    if (!b->upload_pinned[k]) SBA_TRY_HIP(hipHostMalloc(&b->upload_pinned[k], kBatchUploadChunk * 3 * sizeof(double), hipHostMallocDefault));
    if (!b->upload_ev[k]) SBA_TRY_HIP(hipEventCreateWithFlags(&b->upload_ev[k], hipEventDisableTiming));
  }
  sba::DeviceBuffer stage_buf(&b->poisoned);
  SBA_TRY_HIP(stage_buf.alloc(2 * chunk_bytes));
  char* dev_stage[2] = {stage_buf.as<char>(), stage_buf.as<char>() + chunk_bytes};
  int threads = 6;
  if (const char* env = std::getenv("SBA_UPLOAD_THREADS")) { const int v = std::atoi(env); if (v >= 1 && v <= 64) threads = v; }
  threads = std::max(1, std::min<int>(threads, static_cast<int>(std::thread::hardware_concurrency())));
  if (total * 24 < (size_t(8) << 20)) threads = 1;            // a few MB: starting threads costs more than they save
  sba::CopyPool pool(threads);
  struct Job { const double* src; int width; int which; };
  const Job jobs[3] = {{left_xyz, 3, 0}, {right_xyz, 3, 1}, {d12, 2, 2}};
  size_t k = 0;
  for (const Job& job : jobs) {
    if (!job.src || !(which & (1u << job.which))) continue;
    for (size_t first = 0; first < total; first += chunk, ++k) {
      const size_t m = std::min(chunk, total - first), bytes = m * job.width * sizeof(double);
      const int s = static_cast<int>(k & 1);
      if (k >= 2) {                                          // the DMA out of pinned buffer s has finished
        const int rc = sba::event_wait(b->upload_ev[s], "batch upload staging", &b->poisoned);
        if (rc) return rc;
      }
      pool.copy(b->upload_pinned[s], job.src + job.width * (base + first), bytes);
      SBA_TRY_HIP(hipMemcpyAsync(dev_stage[s], b->upload_pinned[s], bytes, hipMemcpyHostToDevice, b->stream));
      SBA_TRY_HIP(hipEventRecord(b->upload_ev[s], b->stream));
      double* stage = reinterpret_cast<double*>(dev_stage[s]);      // stream order protects the device staging buffer
      if (job.which < 2)
        SBA_TRY_HIP(sba::launch_batch_aos_to_planes(stage, m, first, b->offsets_dev, B, b->desc_dev, ppt, b->coord[3 * job.which],
                                                    b->coord[3 * job.which + 1], b->coord[3 * job.which + 2], b->store, b->stream));
      else
        SBA_TRY_HIP(sba::launch_batch_d12_to_planes(stage, m, first, b->offsets_dev, B, b->desc_dev, ppt, b->dplane[0], b->dplane[1],
                                                    b->stream));
    }
  }
  return sba::stream_wait(b->stream, "batch upload", &b->poisoned);      // before the staging buffer goes out of scope
}

}  // namespace

extern "C" {

int sba_batch_create(sba_batch** out, int device, void* stream) {
  if (!out) return sba::set_error(SBA_ERR_INVALID_ARG, "out is null");
  *out = nullptr;
  int count = 0;
  const hipError_t e = hipGetDeviceCount(&count);
  if (e != hipSuccess || count <= 0)
    return sba::set_error(SBA_ERR_NO_DEVICE, "no HIP device available (%s); this library has no CPU path",
                          e != hipSuccess ? hipGetErrorString(e) : "device count is 0");
  if (device < 0 || device >= count) return sba::set_error(SBA_ERR_INVALID_ARG, "device %d out of range [0,%d)", device, count);
  SBA_TRY_HIP(hipSetDevice(device));
  struct Guard {      // any failing HIP call below releases what was created so far
    sba_batch* b;
    ~Guard() { if (b) (void)sba_batch_destroy(b); }
  } guard{new (std::nothrow) sba_batch()};
  sba_batch* b = guard.b;
  if (!b) return sba::set_error(SBA_ERR_HIP, "out of host memory");
  b->device = device;
  hipDeviceProp_t prop;
  SBA_TRY_HIP(hipGetDeviceProperties(&prop, device));
  b->num_cus = prop.multiProcessorCount;
  if (const char* env = std::getenv("SBA_KERNEL"))
    if (std::strcmp(env, "explicit") == 0) b->kind = SBA_KERNEL_EXPLICIT;
  if (const char* env = std::getenv("SBA_PUBLISH")) b->publish = std::strcmp(env, "0") != 0;
  if (const char* env = std::getenv("SBA_PLANE_STAGGER")) {
    const long v = std::atol(env);
    if (v >= 0 && v <= (1 << 20) && v % 16 == 0) b->plane_stagger = static_cast<size_t>(v);
  }
  if (stream) {
    b->stream = static_cast<hipStream_t>(stream);
  } else {
    SBA_TRY_HIP(hipStreamCreateWithFlags(&b->stream, hipStreamNonBlocking));
    b->own_stream = true;
  }
  guard.b = nullptr;     // hand over
  *out = b;
  return SBA_OK;
}

int sba_batch_destroy(sba_batch* b) {
  if (!b) return SBA_OK;
  (void)hipSetDevice(b->device);
  // Bounded drain; a poisoned handle keeps its device resources (hipFree / hipHostFree / hipStreamDestroy wait for the
  // wedged device -- gpurun_out/r2_gputest10.log shows a close() blocked here for good).
  if (!b->poisoned && b->stream) (void)sba::stream_wait(b->stream, "destroy", &b->poisoned);
  if (b->poisoned) {
    delete b;
    return sba::set_error(SBA_ERR_HIP, "batch destroyed while poisoned: its device memory and stream were leaked (a device "
                                       "wait timed out or the device faulted); the process should exit non-zero");
  }
  free_batch_data(b);
  for (void*& q : b->upload_pinned) { if (q) (void)hipHostFree(q); q = nullptr; }
  for (hipEvent_t& e : b->upload_ev) { if (e) (void)hipEventDestroy(e); e = nullptr; }
  if (b->own_stream && b->stream) (void)hipStreamDestroy(b->stream);
  delete b;
  return SBA_OK;
}

int sba_batch_set_kernel(sba_batch* b, int kind) {
  if (!b) return sba::set_error(SBA_ERR_INVALID_ARG, "null batch handle");
  if (kind != SBA_KERNEL_FACTORED && kind != SBA_KERNEL_EXPLICIT) return sba::set_error(SBA_ERR_INVALID_ARG, "bad kernel kind %d", kind);
  b->kind = kind;
  return SBA_OK;
}

int sba_batch_upload(sba_batch* b, const double* left_xyz, const double* right_xyz, const double* d12,
                     const size_t* offsets, int num_pairs, int store) {
  if (!b) return sba::set_error(SBA_ERR_INVALID_ARG, "null batch handle");
  SBA_REFUSE_POISONED(b);
  if (num_pairs < 0 || (num_pairs > 0 && !offsets)) return sba::set_error(SBA_ERR_INVALID_ARG, "bad offsets / num_pairs");
  if (store != SBA_STORE_F64 && store != SBA_STORE_F32) return sba::set_error(SBA_ERR_INVALID_ARG, "bad store %d", store);
  for (int g = 0; g < num_pairs; ++g)
    if (offsets[g + 1] < offsets[g]) return sba::set_error(SBA_ERR_INVALID_ARG, "offsets must be non-decreasing");
  const size_t total = num_pairs > 0 ? offsets[num_pairs] - offsets[0] : 0;
  if (total > 0 && (!left_xyz || !right_xyz)) return sba::set_error(SBA_ERR_INVALID_ARG, "null coordinate array");
  SBA_TRY_HIP(hipSetDevice(b->device));
  // The same pairs again (same offsets, store, depths or not -- e.g. the next images of the same rig, or a re-run): the
  // planes, the descriptors and every mapped buffer stay; only the data moves.  The padding of the planes is still zero:
  // nothing but real elements is ever written.
  if (b->uploaded && b->num_pairs == num_pairs && num_pairs > 0 && b->store == store && b->has_d12 == (d12 != nullptr) &&
      std::equal(b->offsets.begin(), b->offsets.end(), offsets))
    return batch_transfer(b, left_xyz, right_xyz, d12, 7u);
  int rc = free_batch_data(b);
  if (rc) return rc;
  b->num_pairs = num_pairs;
  b->store = store;
  b->has_d12 = d12 != nullptr;
  // every pair starts on a whole vector; vectors are sized for the widest lane group (4 elements) so that the
  // same offsets serve f64 (2 per vector) and f32 (4 per vector) planes, plus one spare vector for the tail load
  const size_t ppt = static_cast<size_t>(sba::points_per_lane(store));
  b->n.resize(num_pairs);
  b->first_vec.resize(num_pairs);
  size_t vec = 0, max_n = 0;
  std::vector<sba::PairDesc> desc(num_pairs);
  for (int g = 0; g < num_pairs; ++g) {
    b->n[g] = offsets[g + 1] - offsets[g];
    b->first_vec[g] = vec;
    desc[g] = sba::PairDesc{vec, b->n[g], sba::kPairTile, 0ull};
    vec += (b->n[g] + ppt - 1) / ppt + 1;
    max_n = std::max(max_n, b->n[g]);
  }
  // Layout.  Contiguous pairs (above) make every block of a one-block-per-pair sweep its own sequential stream per plane:
  // num_pairs x 8 streams.  INTERLEAVED (tile t of every pair side by side, 4 KiB tiles) the blocks, which advance in step,
  // read one contiguous window of every plane like the single-problem grid-stride sweep does (8 streams): 3-4 % more of the
  // HBM bandwidth (tools/stream_probe.hip: stride vs chunk; measured on the C5 step: DESIGN.md section 3.5).  It pads
  // every pair to the longest one's tile count, so it is used when that wastes at most a quarter (SBA_BATCH_INTERLEAVE = 0 /
  // 1 forces either).
  const size_t tiles = ((max_n + ppt - 1) / ppt + 1 + sba::kPairTile - 1) / sba::kPairTile;      // of the longest pair, spare vector included
  const size_t inter_vecs = tiles * static_cast<size_t>(num_pairs) * sba::kPairTile;
  bool interleave = num_pairs >= 2 && inter_vecs <= vec + vec / 4;
  if (const char* env = std::getenv("SBA_BATCH_INTERLEAVE")) interleave = num_pairs >= 1 && env[0] != '0';
  b->tile_stride = sba::kPairTile;
  if (interleave && max_n > 0) {
    b->tile_stride = sba::kPairTile * static_cast<size_t>(num_pairs);
    for (int g = 0; g < num_pairs; ++g) {
      b->first_vec[g] = static_cast<size_t>(g) * sba::kPairTile;
      desc[g] = sba::PairDesc{b->first_vec[g], b->n[g], b->tile_stride, 0ull};
    }
    vec = inter_vecs;
  }
  b->total_vecs = vec + 1;
  const size_t elems = b->total_vecs * ppt, esz = store == SBA_STORE_F64 ? 8 : 4;
  b->plane_elems = elems;
  b->offsets.assign(offsets, offsets + num_pairs + 1);
  for (int k = 0; k < 6; ++k) {
    const size_t lead = b->plane_stagger * static_cast<size_t>(k);
    SBA_TRY_HIP(hipMalloc(&b->plane_base[k], lead + elems * esz));
    SBA_TRY_HIP(hipMemsetAsync(b->plane_base[k], 0, lead + elems * esz, b->stream));
    b->coord[k] = static_cast<char*>(b->plane_base[k]) + lead;
  }
  if (d12)
    for (int k = 0; k < 2; ++k) {
      const size_t lead = b->plane_stagger * static_cast<size_t>(6 + k);
      SBA_TRY_HIP(hipMalloc(&b->plane_base[6 + k], lead + elems * 8));
      SBA_TRY_HIP(hipMemsetAsync(b->plane_base[6 + k], 0, lead + elems * 8, b->stream));
      b->dplane[k] = reinterpret_cast<double*>(static_cast<char*>(b->plane_base[6 + k]) + lead);
    }
  if (num_pairs == 0) { b->uploaded = true; return SBA_OK; }

  // Blocks per pair: with at least one pair per CU, one block per pair (measured best at 256 pairs x 50k matches:
  // 198 / 206 / 205 / 213 us per step for 1 / 2 / 3 / 4 blocks per pair); with fewer pairs, spread each pair over
  // enough blocks to put two blocks on every CU like the single-problem sweep -- never more than a pair can use.
  const size_t need = (((max_n + ppt - 1) / ppt) + sba::kBlock - 1) / sba::kBlock;
  const size_t share = num_pairs >= b->num_cus ? 1 : static_cast<size_t>(2 * b->num_cus) / num_pairs;
  b->bpp = static_cast<int>(std::max<size_t>(1, std::min(std::max<size_t>(need, 1), share)));
  if (const char* env = std::getenv("SBA_BATCH_BPP")) { const int v = std::atoi(env); if (v >= 1 && v <= 1024) b->bpp = v; }

  SBA_TRY_HIP(hipMalloc(reinterpret_cast<void**>(&b->desc_dev), sizeof(sba::PairDesc) * num_pairs));
  SBA_TRY_HIP(hipMemcpy(b->desc_dev, desc.data(), sizeof(sba::PairDesc) * num_pairs, hipMemcpyHostToDevice));
  SBA_TRY_HIP(hipMalloc(reinterpret_cast<void**>(&b->params_dev), sizeof(sba::SweepParams) * num_pairs));
  SBA_TRY_HIP(hipMemset(b->params_dev, 0, sizeof(sba::SweepParams) * num_pairs));   // n = 0 until a prepare kernel has run
  SBA_TRY_HIP(hipHostMalloc(reinterpret_cast<void**>(&b->state_host), sizeof(sba::BatchState) * num_pairs,
                            hipHostMallocMapped | hipHostMallocCoherent));
  std::memset(b->state_host, 0, sizeof(sba::BatchState) * num_pairs);
  SBA_TRY_HIP(hipHostGetDevicePointer(reinterpret_cast<void**>(&b->state_host_dev), b->state_host, 0));
  SBA_TRY_HIP(hipMalloc(reinterpret_cast<void**>(&b->frames_dev), sizeof(double) * 18 * num_pairs));
  SBA_TRY_HIP(hipMalloc(reinterpret_cast<void**>(&b->partials), sizeof(double) * sba::kRow * num_pairs * b->bpp));
  SBA_TRY_HIP(hipMalloc(reinterpret_cast<void**>(&b->packs_dev), sizeof(double) * 24 * num_pairs));
  SBA_TRY_HIP(hipHostMalloc(reinterpret_cast<void**>(&b->packs_host), sizeof(double) * (24 * num_pairs + 8),
                            hipHostMallocMapped | hipHostMallocCoherent));
  std::memset(b->packs_host, 0, sizeof(double) * (24 * num_pairs + 8));
  SBA_TRY_HIP(hipHostGetDevicePointer(reinterpret_cast<void**>(&b->packs_host_dev), b->packs_host, 0));
  SBA_TRY_HIP(hipHostMalloc(reinterpret_cast<void**>(&b->lm_io_host), sizeof(sba::BatchLmIo) * num_pairs,
                            hipHostMallocMapped | hipHostMallocCoherent));
  std::memset(b->lm_io_host, 0, sizeof(sba::BatchLmIo) * num_pairs);
  SBA_TRY_HIP(hipHostGetDevicePointer(reinterpret_cast<void**>(&b->lm_io_host_dev), b->lm_io_host, 0));
  SBA_TRY_HIP(hipMalloc(reinterpret_cast<void**>(&b->lm_ticket), 64));
  SBA_TRY_HIP(hipMemset(b->lm_ticket, 0, 64));
  b->seq = 0;

  {
    std::vector<unsigned long long> off64(num_pairs + 1);
    for (int g = 0; g <= num_pairs; ++g) off64[g] = offsets[g] - offsets[0];
    SBA_TRY_HIP(hipMalloc(reinterpret_cast<void**>(&b->offsets_dev), sizeof(unsigned long long) * off64.size()));
    SBA_TRY_HIP(hipMemcpy(b->offsets_dev, off64.data(), sizeof(unsigned long long) * off64.size(), hipMemcpyHostToDevice));
  }
  rc = batch_transfer(b, left_xyz, right_xyz, d12, 7u);
  if (rc) return rc;
  b->uploaded = true;
  return SBA_OK;
}

// Only the per-match depths again (init_d): the pipeline refines them in place, the coordinates stay resident.
int sba_batch_set_depths(sba_batch* b, const double* d12) {
  if (!b) return sba::set_error(SBA_ERR_INVALID_ARG, "null batch handle");
  SBA_REFUSE_POISONED(b);
  if (!b->uploaded) return sba::set_error(SBA_ERR_NOT_UPLOADED, "no pairs uploaded");
  if (!b->has_d12) return sba::set_error(SBA_ERR_INVALID_ARG, "the batch was uploaded without per-match depths");
  if (b->num_pairs == 0) return SBA_OK;
  if (!d12) return sba::set_error(SBA_ERR_INVALID_ARG, "d12 is null");
  SBA_TRY_HIP(hipSetDevice(b->device));
  return batch_transfer(b, nullptr, nullptr, d12, 4u);
}

int sba_batch_size(const sba_batch* b, int* num_pairs, int* blocks_per_pair) {
  if (!b) return sba::set_error(SBA_ERR_INVALID_ARG, "null batch handle");
  if (num_pairs) *num_pairs = b->num_pairs;
  if (blocks_per_pair) *blocks_per_pair = b->bpp;
  return SBA_OK;
}

int sba_batch_eval(sba_batch* b, int mode, int depth_mode, const double* rot, const double* tran,
                   const double* d1, const double* d2, double huber_delta, double* packs) {
  int rc = check_batch_args(b, mode, depth_mode, rot, tran);
  if (rc) return rc;
  if (b->num_pairs == 0) return SBA_OK;
  if (!packs) return sba::set_error(SBA_ERR_INVALID_ARG, "packs is null");
  SBA_TRY_HIP(hipSetDevice(b->device));
  rc = batch_launch(b, mode, depth_mode, rot, tran, d1, d2, huber_delta, nullptr);
  if (rc) return rc;
  for (int g = 0; g < b->num_pairs; ++g) convert_pack(b, mode, rot + 3 * g, b->packs_host + 24 * g, packs + 24 * g);
  return SBA_OK;
}

int sba_batch_eval_timed(sba_batch* b, int mode, int depth_mode, const double* rot, const double* tran,
                         const double* d1, const double* d2, double huber_delta, int steps, double* packs,
                         double* mean_step_ms, double* mean_prepare_ms, double* mean_device_ms,
                         double* mean_convert_ms) {
  int rc = check_batch_args(b, mode, depth_mode, rot, tran);
  if (rc) return rc;
  if (steps <= 0) return sba::set_error(SBA_ERR_INVALID_ARG, "steps must be positive");
  if (b->num_pairs > 0 && !packs) return sba::set_error(SBA_ERR_INVALID_ARG, "packs is null");
  SBA_TRY_HIP(hipSetDevice(b->device));
  using clk = std::chrono::steady_clock;
  double t_prep = 0, t_dev = 0, t_conv = 0;
  const auto t_begin = clk::now();
  for (int s = 0; s < steps && b->num_pairs > 0; ++s) {
    double split[2] = {0, 0};
    const auto t0 = clk::now();
    rc = batch_launch(b, mode, depth_mode, rot, tran, d1, d2, huber_delta, nullptr, split);
    if (rc) return rc;
    const auto t1 = clk::now();
    for (int g = 0; g < b->num_pairs; ++g) convert_pack(b, mode, rot + 3 * g, b->packs_host + 24 * g, packs + 24 * g);
    const auto t2 = clk::now();
    t_prep += split[0];
    t_dev += std::chrono::duration<double, std::milli>(t1 - t0).count() - split[0];
    t_conv += std::chrono::duration<double, std::milli>(t2 - t1).count();
  }
  const double total = std::chrono::duration<double, std::milli>(clk::now() - t_begin).count();
  if (mean_step_ms) *mean_step_ms = total / steps;
  if (mean_prepare_ms) *mean_prepare_ms = t_prep / steps;
  if (mean_device_ms) *mean_device_ms = t_dev / steps;
  if (mean_convert_ms) *mean_convert_ms = t_conv / steps;
  return SBA_OK;
}

namespace {
// `repeat` launches of one kernel with a HIP event between every two.  step_kernel = false: the bare batch_sweep_kernel;
// true: the dominant kernel of the step this batch really runs -- batch_step_kernel EXACTLY as a step launches it (state
// read from mapped host memory, packs and sequence word published to the host: the kernel rocprofv3 sees in a step) when
// the step is fused, batch_sweep_kernel otherwise.  The only difference from a step: the host does not wait in between.
int kernel_launch_times(sba_batch* b, int mode, int depth_mode, const double* rot, const double* tran, const double* d1,
                        const double* d2, double huber_delta, int repeat, float* launch_ms, bool step_kernel) {
  int rc = check_batch_args(b, mode, depth_mode, rot, tran);
  if (rc) return rc;
  if (!launch_ms || repeat < 1 || repeat > 4096) return sba::set_error(SBA_ERR_INVALID_ARG, "bad launch_ms/repeat (1..4096)");
  if (b->num_pairs == 0) { for (int i = 0; i < repeat; ++i) launch_ms[i] = 0.f; return SBA_OK; }
  SBA_TRY_HIP(hipSetDevice(b->device));
  // the three-kernel chain: its prepare kernel leaves params_dev filled for the bare sweep launches below (the fused
  // one-launch step keeps every pair's sweep state in LDS and never writes params_dev)
  rc = batch_launch(b, mode, depth_mode, rot, tran, d1, d2, huber_delta, nullptr, nullptr, false);
  if (rc) return rc;
  sba::Planes pl;
  for (int k = 0; k < 3; ++k) { pl.x1[k] = b->coord[k]; pl.x2[k] = b->coord[3 + k]; }
  pl.d1 = b->dplane[0]; pl.d2 = b->dplane[1];
  struct Events {
    std::vector<hipEvent_t> ev;
    ~Events() { for (hipEvent_t e : ev) if (e) (void)hipEventDestroy(e); }
  } events;
  events.ev.assign(static_cast<size_t>(repeat) + 1, nullptr);
  for (hipEvent_t& e : events.ev) SBA_TRY_HIP(hipEventCreate(&e));
  SBA_TRY_HIP(hipEventRecord(events.ev[0], b->stream));
  const bool fused = step_kernel && step_is_fused(b);
  for (int i = 0; i < repeat; ++i) {
    if (fused)
      SBA_TRY_HIP(sba::launch_batch_step_fused(mode, depth_mode, b->store, b->kind, huber_delta, pl, b->state_host_dev,
                                               b->desc_dev, b->num_pairs, b->packs_dev, b->packs_host_dev, b->lm_ticket,
                                               ++b->seq, b->stream));
    else
      SBA_TRY_HIP(sba::launch_batch_sweep_only(mode, depth_mode, b->store, b->kind, huber_delta > 0.0, pl, b->params_dev,
                                               b->desc_dev, b->num_pairs, b->bpp, b->partials, b->stream));
    SBA_TRY_HIP(hipEventRecord(events.ev[static_cast<size_t>(i) + 1], b->stream));
  }
  SBA_TRY_HIP(hipEventSynchronize(events.ev[static_cast<size_t>(repeat)]));
  for (int i = 0; i < repeat; ++i)
    SBA_TRY_HIP(hipEventElapsedTime(&launch_ms[i], events.ev[static_cast<size_t>(i)], events.ev[static_cast<size_t>(i) + 1]));
  return SBA_OK;
}
}  // namespace

int sba_batch_sweep_launch_times(sba_batch* b, int mode, int depth_mode, const double* rot, const double* tran,
                                 const double* d1, const double* d2, double huber_delta, int repeat, float* launch_ms) {
  return kernel_launch_times(b, mode, depth_mode, rot, tran, d1, d2, huber_delta, repeat, launch_ms, false);
}
int sba_batch_step_launch_times(sba_batch* b, int mode, int depth_mode, const double* rot, const double* tran,
                                const double* d1, const double* d2, double huber_delta, int repeat, float* launch_ms) {
  return kernel_launch_times(b, mode, depth_mode, rot, tran, d1, d2, huber_delta, repeat, launch_ms, true);
}
int sba_batch_step_is_fused(const sba_batch* b) {
  if (!b) return sba::set_error(SBA_ERR_INVALID_ARG, "null batch");
  if (!b->uploaded) return sba::set_error(SBA_ERR_NOT_UPLOADED, "no pairs uploaded");
  return step_is_fused(b) ? 1 : 0;
}

// The d-only stage for every pair of the batch (reference spherical_bundle_adjuster.cpp:196-197 once per pair): B independent
// bounded trust-region problems -- own radius, own projected line search, own convergence -- advanced in LOCK-STEP: one
// launch of batch_depth_step_kernel runs the next pass of every unfinished pair, the host feeds every pair's nine
// reductions to that pair's DepthStageSolver (the very state machine of sba_problem_solve_depths).
namespace {
// d_uniform (may be null): double[num_pairs][2] <- the refined init_d[0][0], init_d[1][0] of every pair (the uniform depths of
// the reference's rot / tran stages, .cpp:941-942); a pair with one match gets its only depth twice, an empty pair zeros.
int batch_solve_depths_impl(sba_batch* b, const double* rot, const double* tran, double lambda, double c, const sba_lm_options* opt,
                            double* d12_out, sba_lm_summary* summaries, int* status, double* d_uniform);
}  // namespace

int sba_batch_solve_depths(sba_batch* b, const double* rot, const double* tran, double lambda, double c,
                           const sba_lm_options* opt, double* d12_out, sba_lm_summary* summaries, int* status) {
  return batch_solve_depths_impl(b, rot, tran, lambda, c, opt, d12_out, summaries, status, nullptr);
}

}  // extern "C"

namespace {
int batch_solve_depths_impl(sba_batch* b, const double* rot, const double* tran, double lambda, double c, const sba_lm_options* opt,
                            double* d12_out, sba_lm_summary* summaries, int* status, double* d_uniform) {
  if (!b) return sba::set_error(SBA_ERR_INVALID_ARG, "null batch handle");
  SBA_REFUSE_POISONED(b);
  if (!b->uploaded) return sba::set_error(SBA_ERR_NOT_UPLOADED, "no pairs uploaded");
  const int B = b->num_pairs;
  if (B == 0) return SBA_OK;
  if (!rot || !tran) return sba::set_error(SBA_ERR_INVALID_ARG, "rot/tran must not be null");
  if (!b->has_d12) return sba::set_error(SBA_ERR_INVALID_ARG, "the d-only stage needs per-match depths uploaded");
  if (!b->publish) return sba::set_error(SBA_ERR_UNSUPPORTED, "the batched d-only stage needs host-mapped publication (SBA_PUBLISH=0 is set)");
  SBA_TRY_HIP(hipSetDevice(b->device));
  sba_lm_options o;
  if (opt) o = *opt; else sba::lm_default_options(&o);
  const auto t_start = std::chrono::steady_clock::now();
  const size_t elems = b->plane_elems;
  if (!b->depth_work) {
    SBA_TRY_HIP(hipMalloc(reinterpret_cast<void**>(&b->depth_work), 4 * elems * sizeof(double)));
    SBA_TRY_HIP(hipMalloc(reinterpret_cast<void**>(&b->depth_const_dev), sizeof(sba::BatchDepthConst) * B));
    SBA_TRY_HIP(hipHostMalloc(reinterpret_cast<void**>(&b->depth_pass_host), sizeof(sba::BatchDepthPass) * B,
                              hipHostMallocMapped | hipHostMallocCoherent));
    SBA_TRY_HIP(hipHostGetDevicePointer(reinterpret_cast<void**>(&b->depth_pass_host_dev), b->depth_pass_host, 0));
    SBA_TRY_HIP(hipHostMalloc(reinterpret_cast<void**>(&b->depth_out_host), sizeof(double) * (static_cast<size_t>(B) * sba::DEPTH_ROW + 8),
                              hipHostMallocMapped | hipHostMallocCoherent));
    std::memset(b->depth_out_host, 0, sizeof(double) * (static_cast<size_t>(B) * sba::DEPTH_ROW + 8));
    SBA_TRY_HIP(hipHostGetDevicePointer(reinterpret_cast<void**>(&b->depth_out_host_dev), b->depth_out_host, 0));
    b->depth_seq = 0;
    // candidate + scaling planes start zeroed, once: a candidate plane becomes a pair's depth plane when a step is accepted,
    // and its padding must be zeros like the uploaded planes' -- the passes only ever write real elements (and zeros into the
    // padding element of an odd-sized pair), so the padding stays zero from stage to stage
    SBA_TRY_HIP(hipMemsetAsync(b->depth_work, 0, 4 * elems * sizeof(double), b->stream));
  }
  double *w1 = b->depth_work, *w2 = w1 + elems, *sc1 = w1 + 2 * elems, *sc2 = w1 + 3 * elems;
  std::vector<sba::BatchDepthConst> cst(B);
  for (int g = 0; g < B; ++g) {
    for (int a = 0; a < 3; ++a)
      if (!std::isfinite(rot[3 * g + a]) || !std::isfinite(tran[3 * g + a])) return sba::set_error(SBA_ERR_INVALID_ARG, "non-finite rot/tran of pair %d", g);
    sba::rotation_and_derivatives(rot + 3 * g, cst[g].R, nullptr);
    for (int a = 0; a < 3; ++a) cst[g].t[a] = tran[3 * g + a];
  }
  SBA_TRY_HIP(hipMemcpyAsync(b->depth_const_dev, cst.data(), sizeof(sba::BatchDepthConst) * B, hipMemcpyHostToDevice, b->stream));
  { const int _rc = sba::stream_wait(b->stream, "batched d-only set-up", &b->poisoned); if (_rc) return _rc; }   // cst goes out of use

  sba::Planes pl;
  for (int k = 0; k < 3; ++k) { pl.x1[k] = b->coord[k]; pl.x2[k] = b->coord[3 + k]; }
  pl.d1 = b->dplane[0]; pl.d2 = b->dplane[1];
  // d12_out is indexed like the uploaded arrays (pair g at offsets[g]); the device buffer holds rows offsets[0] .. offsets[B]
  const size_t base = b->offsets.empty() ? 0 : b->offsets.front();
  std::vector<unsigned long long> off64;
  for (size_t v : b->offsets) off64.push_back(v - base);
  const size_t total = b->offsets.empty() ? 0 : b->offsets.back() - base;
  sba::DeviceBuffer flip_dev(&b->poisoned), off_dev(&b->poisoned), out_dev(&b->poisoned);
  std::vector<double> d12_tmp;                  // the lock-step path takes the uniform depths from the full read-back
  bool device_solve = true;
  if (const char* env = std::getenv("SBA_BATCH_DEVICE_DEPTH")) device_solve = std::strcmp(env, "0") != 0;
  if (!device_solve && d_uniform && !d12_out && total > 0) { d12_tmp.resize(2 * (base + total)); d12_out = d12_tmp.data(); }
  const bool want_out = d12_out && total > 0;
  if (want_out) {
    SBA_TRY_HIP(off_dev.alloc(sizeof(unsigned long long) * off64.size()));
    SBA_TRY_HIP(hipMemcpyAsync(off_dev.ptr, off64.data(), sizeof(unsigned long long) * off64.size(), hipMemcpyHostToDevice, b->stream));
    SBA_TRY_HIP(out_dev.alloc(2 * total * sizeof(double)));
  }
  volatile unsigned long long* flag = reinterpret_cast<volatile unsigned long long*>(b->depth_out_host + static_cast<size_t>(B) * sba::DEPTH_ROW);

  // One launch for the whole stage: every pair's solver runs on the device next to its passes (batch_depth_solve_kernel).
  // SBA_BATCH_DEVICE_DEPTH=0 keeps the host lock-step loop below (the A/B and the cross-check of tests/test_gpu_batch.py).
  if (device_solve) {
    sba::BatchLmIo* io = b->lm_io_host;
    for (int g = 0; g < B; ++g) { io[g] = sba::BatchLmIo{}; io[g].status = SBA_ERR_NUMERIC; }
    const unsigned long long seq = ++b->depth_seq;
    unsigned long long* flag_dev = reinterpret_cast<unsigned long long*>(b->depth_out_host_dev + static_cast<size_t>(B) * sba::DEPTH_ROW);
    // Hybrid (default): the one-launch kernel runs the first passes of every pair and hands the pairs that need more over to
    // per-pass launches whose blocks are dealt out to the pairs still iterating -- the CUs of the pairs that are done join in.
    // SBA_BATCH_DEPTH_FIRST_PASSES: the cap of the first launch (default 16; 0: the one-launch kernel runs to the end).
    int first_passes = 2 * B <= b->num_cus ? 1 : 16;      // few pairs: one block per pair would leave most CUs idle -- hand over at once
    if (const char* env = std::getenv("SBA_BATCH_DEPTH_FIRST_PASSES")) { const int v = std::atoi(env); if (v >= 0 && v <= 100000) first_passes = v; }
    const bool hybrid = first_passes > 0;
    const int sweep_grid = std::max(1, b->num_cus);
    if (hybrid) {
      const int rc2 = ensure_dyn(b, static_cast<size_t>(sba::kDynOver) * static_cast<size_t>(sweep_grid) + 2 * static_cast<size_t>(B));
      if (rc2) return rc2;
      SBA_TRY_HIP(hipMemsetAsync(b->dyn_finish, 0, static_cast<size_t>(B), b->stream));
    }
    SBA_TRY_HIP(sba::launch_batch_depth_solve(b->store, pl, b->desc_dev, b->depth_const_dev, B, lambda, c, o, b->dplane[0], b->dplane[1],
                                              w1, w2, sc1, sc2, off_dev.as<unsigned long long>(), want_out ? out_dev.as<double>() : nullptr,
                                              b->lm_io_host_dev, b->lm_ticket, flag_dev, seq, b->stream, hybrid ? first_passes : 0,
                                              hybrid ? b->dyn_state : nullptr, b->dyn_depth_req, hybrid ? b->dyn_done : nullptr, b->dyn_finish));
    int wrc = sba::wait_for_sequence(flag, seq, b->stream, "batched d-only stage", &b->poisoned);
    if (wrc) return wrc;
    unsigned long long left = 0;
    for (int g = 0; g < B; ++g) if (io[g].status == 1) { ++left; io[g].status = SBA_ERR_NUMERIC; }
    if (left > 0) {
      SBA_TRY_HIP(sba::launch_batch_dyn_first_list(b->dyn_ctl, b->dyn_active, b->dyn_done, B, b->stream));
      const int max_passes = sba::batch_depth_pass_bound(o);
      volatile unsigned long long* words = b->dyn_host;
      int pass = 0;
      while (left > 0 && pass < max_passes) {
        const int group = std::min(4, max_passes - pass);
        unsigned long long dseq = 0;
        for (int k = 0; k < group; ++k, ++pass) {
          const int parity = (1 + pass) & 1;
          const bool last = k == group - 1;
          if (last) dseq = ++b->dyn_seq;
          SBA_TRY_HIP(sba::launch_batch_depth_dyn_pass(b->store, pl, b->desc_dev, b->depth_const_dev, B, lambda, c, o, b->dplane[0], b->dplane[1],
                                                       w1, w2, sc1, sc2, parity, sweep_grid, b->dyn_state, b->dyn_depth_req, b->dyn_ctl,
                                                       b->dyn_active, b->dyn_done, b->dyn_finish, b->dyn_partials, b->lm_io_host_dev, b->stream));
          SBA_TRY_HIP(sba::launch_batch_dyn_compact(b->dyn_ctl, b->dyn_active, b->dyn_done, B, parity, last ? b->dyn_host_dev : nullptr, dseq,
                                                    b->stream));
        }
        wrc = sba::wait_for_sequence(words + 1, dseq, b->stream, "batched d-only stage (dynamic shares)", &b->poisoned);
        if (wrc) return wrc;
        left = words[0];
      }
      // the pairs that went through the dynamic passes: copy-back and output (bits of dyn_finish)
      SBA_TRY_HIP(sba::launch_batch_depth_finish(b->store, b->desc_dev, b->dyn_finish, B, b->dplane[0], b->dplane[1], w1, w2,
                                                 off_dev.as<unsigned long long>(), want_out ? out_dev.as<double>() : nullptr, b->stream));
    }
    if (want_out)
      SBA_TRY_HIP(hipMemcpyAsync(d12_out + 2 * base, out_dev.ptr, 2 * total * sizeof(double), hipMemcpyDeviceToHost, b->stream));
    wrc = sba::stream_wait(b->stream, "batched d-only stage", &b->poisoned);
    if (wrc) return wrc;
    const double seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_start).count();
    int failures = 0;
    for (int g = 0; g < B; ++g) {
      if (summaries) { summaries[g] = io[g].summary; summaries[g].seconds_total = seconds; }   // wall clock of the whole batch
      if (status) status[g] = io[g].status;
      if (d_uniform) { d_uniform[2 * g] = io[g].d1; d_uniform[2 * g + 1] = io[g].d2; }
      if (io[g].status != SBA_OK) ++failures;
    }
    if (failures) return sba::set_error(SBA_ERR_NUMERIC, "%d of %d pairs failed in the d-only stage (see per-pair status)", failures, B);
    return SBA_OK;
  }

  std::vector<sba::DepthStageSolver> solver(B);
  std::vector<unsigned char> flip(B, 0), active(B, 1);
  for (int g = 0; g < B; ++g) solver[g].start(o);
  int remaining = B;
  while (remaining > 0) {
    for (int g = 0; g < B; ++g) {
      sba::BatchDepthPass& ps = b->depth_pass_host[g];
      if (!active[g]) { ps = sba::BatchDepthPass{1.0, 1.0, 0ull, 0u, 0u}; continue; }
      const sba::DepthPassRequest& rq = solver[g].request();
      ps.radius = rq.radius; ps.alpha = rq.alpha; ps.n = b->n[g]; ps.pad_ = 0;
      ps.flags = (rq.first ? 1u : 0u) | (rq.keep_diagonal ? 2u : 0u) | (o.jacobi_scaling ? 4u : 0u) | (flip[g] ? 8u : 0u);
    }
    const unsigned long long seq = ++b->depth_seq;
    SBA_TRY_HIP(sba::launch_batch_depth_step(b->store, pl, b->desc_dev, b->depth_const_dev, b->depth_pass_host_dev, B, lambda, c,
                                             o.min_lm_diagonal, o.max_lm_diagonal, b->dplane[0], b->dplane[1], w1, w2, sc1, sc2,
                                             b->depth_out_host_dev, b->lm_ticket, seq, b->stream));
    const int wrc = sba::wait_for_sequence(flag, seq, b->stream, "batched d-only pass", &b->poisoned);
    if (wrc) return wrc;
    for (int g = 0; g < B; ++g) {
      if (!active[g]) continue;
      solver[g].feed(b->depth_out_host + static_cast<size_t>(g) * sba::DEPTH_ROW);
      if (solver[g].take_candidate()) flip[g] ^= 1;
      if (solver[g].done()) { active[g] = 0; --remaining; }
    }
  }
  // results back into the batch's own depth planes (pairs that ended on an odd number of accepted steps), and out to the host
  for (unsigned char& f : flip) f = static_cast<unsigned char>((f & 1) | 2);      // bit 0: copy back; bit 1: write `out`
  SBA_TRY_HIP(flip_dev.alloc(static_cast<size_t>(B)));
  SBA_TRY_HIP(hipMemcpyAsync(flip_dev.ptr, flip.data(), static_cast<size_t>(B), hipMemcpyHostToDevice, b->stream));
  SBA_TRY_HIP(sba::launch_batch_depth_finish(b->store, b->desc_dev, flip_dev.as<unsigned char>(), B, b->dplane[0], b->dplane[1], w1, w2,
                                             off_dev.as<unsigned long long>(), want_out ? out_dev.as<double>() : nullptr, b->stream));
  if (want_out)
    SBA_TRY_HIP(hipMemcpyAsync(d12_out + 2 * base, out_dev.ptr, 2 * total * sizeof(double), hipMemcpyDeviceToHost, b->stream));
  { const int _rc = sba::stream_wait(b->stream, "batched d-only stage", &b->poisoned); if (_rc) return _rc; }
  const double seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_start).count();
  int failures = 0;
  for (int g = 0; g < B; ++g) {
    if (summaries) { summaries[g] = solver[g].summary(); summaries[g].seconds_total = seconds; }   // wall clock of the whole batch
    if (status) status[g] = solver[g].status();
    if (d_uniform) {
      const size_t lo = b->offsets[g], n_g = b->n[g];
      d_uniform[2 * g] = n_g > 0 ? d12_out[2 * lo] : 0.0;
      d_uniform[2 * g + 1] = n_g > 1 ? d12_out[2 * (lo + 1)] : d_uniform[2 * g];
    }
    if (solver[g].status() != SBA_OK) ++failures;
  }
  if (failures) return sba::set_error(SBA_ERR_NUMERIC, "%d of %d pairs failed in the d-only stage (see per-pair status)", failures, B);
  return SBA_OK;
}
}  // namespace

extern "C" {

// ---- the 8-point initial guess of every pair (reference initial_guess, .cpp:47-181, once per pair) -----------------------------
namespace {
// the moments of every pair -> b->epi_groups_dev; to_host: also -> b->epi_groups_host, synchronised
int batch_group_moments(sba_batch* b, bool to_host) {
  const int B = b->num_pairs;
  const size_t gsz = static_cast<size_t>(sba::epi::kGroups) * sba::epi::kMom;
  if (!b->epi_groups_dev) {
    SBA_TRY_HIP(hipMalloc(reinterpret_cast<void**>(&b->epi_groups_dev), gsz * B * sizeof(double)));
    SBA_TRY_HIP(hipHostMalloc(reinterpret_cast<void**>(&b->epi_groups_host), gsz * B * sizeof(double), hipHostMallocDefault));
    SBA_TRY_HIP(hipMalloc(reinterpret_cast<void**>(&b->guess_out_dev), sizeof(sba::BatchGuessOut) * B));
    SBA_TRY_HIP(hipHostMalloc(reinterpret_cast<void**>(&b->guess_out_host), sizeof(sba::BatchGuessOut) * B, hipHostMallocDefault));
  }
  sba::Planes pl;
  for (int k = 0; k < 3; ++k) { pl.x1[k] = b->coord[k]; pl.x2[k] = b->coord[3 + k]; }
  pl.d1 = b->dplane[0]; pl.d2 = b->dplane[1];
  SBA_TRY_HIP(sba::launch_batch_epipolar_moments(b->store, pl, b->desc_dev, B, b->epi_groups_dev, b->stream));
  if (!to_host) return SBA_OK;
  SBA_TRY_HIP(hipMemcpyAsync(b->epi_groups_host, b->epi_groups_dev, gsz * B * sizeof(double), hipMemcpyDeviceToHost, b->stream));
  return sba::stream_wait(b->stream, "batched group moments", &b->poisoned);
}
}  // namespace

int sba_batch_epipolar_moments(sba_batch* b, double* groups) {
  if (!b || !groups) return sba::set_error(SBA_ERR_INVALID_ARG, "null argument");
  SBA_REFUSE_POISONED(b);
  if (!b->uploaded) return sba::set_error(SBA_ERR_NOT_UPLOADED, "no pairs uploaded");
  if (b->num_pairs == 0) return SBA_OK;
  SBA_TRY_HIP(hipSetDevice(b->device));
  const int rc = batch_group_moments(b, true);
  if (rc) return rc;
  std::memcpy(groups, b->epi_groups_host, static_cast<size_t>(b->num_pairs) * sba::epi::kGroups * sba::epi::kMom * sizeof(double));
  return SBA_OK;
}

int sba_batch_initial_guess(sba_batch* b, int trials, double subset_fraction, unsigned long long seed, double* rot_euler,
                            double* tran, int* num_candidates, int* status) {
  if (!b) return sba::set_error(SBA_ERR_INVALID_ARG, "null batch handle");
  SBA_REFUSE_POISONED(b);
  if (!b->uploaded) return sba::set_error(SBA_ERR_NOT_UPLOADED, "no pairs uploaded");
  if (trials < 1 || !(subset_fraction > 0.0) || subset_fraction > 1.0)
    return sba::set_error(SBA_ERR_INVALID_ARG, "bad trials / subset_fraction");
  const int B = b->num_pairs;
  if (B == 0) return SBA_OK;
  if (!rot_euler || !tran) return sba::set_error(SBA_ERR_INVALID_ARG, "rot_euler/tran must not be null");
  SBA_TRY_HIP(hipSetDevice(b->device));
  // The trials and the consensus pick run on the device too, one block per pair (batch_guess_kernel: the host's own source
  // for a trial).  SBA_BATCH_DEVICE_GUESS=0, or more trials than the kernel holds: the host loop below.
  bool device_trials = trials <= sba::kGuessMaxTrials;
  if (const char* env = std::getenv("SBA_BATCH_DEVICE_GUESS")) device_trials = device_trials && std::strcmp(env, "0") != 0;
  if (device_trials) {
    int rc = batch_group_moments(b, false);
    if (rc) return rc;
    SBA_TRY_HIP(sba::launch_batch_guess(b->epi_groups_dev, B, trials, subset_fraction, seed, b->guess_out_dev, b->stream));
    SBA_TRY_HIP(hipMemcpyAsync(b->guess_out_host, b->guess_out_dev, sizeof(sba::BatchGuessOut) * B, hipMemcpyDeviceToHost, b->stream));
    rc = sba::stream_wait(b->stream, "batched initial guess", &b->poisoned);
    if (rc) return rc;
    int failures = 0;
    for (int g = 0; g < B; ++g) {
      const sba::BatchGuessOut& r = b->guess_out_host[g];
      for (int i = 0; i < 3; ++i) { rot_euler[3 * g + i] = r.euler[i]; tran[3 * g + i] = r.tran[i]; }
      if (num_candidates) num_candidates[g] = r.num_candidates;
      if (status) status[g] = r.status;
      if (r.status != SBA_OK) ++failures;
    }
    if (failures) return sba::set_error(SBA_ERR_NUMERIC, "%d of %d pairs have no valid rotation candidate (see per-pair status)", failures, B);
    return SBA_OK;
  }
  const int rc = batch_group_moments(b, true);
  if (rc) return rc;
  // the trials of a pair are ~0.5 ms of small-matrix host work: pairs are spread over the host threads (sba_set_host_threads,
  // the reference's set_omp), each pair's trials run serially in trial order -- the result does not depend on the thread count
  const size_t gsz = static_cast<size_t>(sba::epi::kGroups) * sba::epi::kMom;
  std::vector<int> st(B, SBA_OK);
  auto run = [&](int first, int last) {
    for (int g = first; g < last; ++g) {
      const sba::epi::GuessResult r = sba::epi::initial_guess_from_groups(b->epi_groups_host + gsz * g, trials, subset_fraction, seed, 1);
      if (num_candidates) num_candidates[g] = r.num_candidates;
      for (int i = 0; i < 3; ++i) { rot_euler[3 * g + i] = r.picked >= 0 ? r.euler[i] : 0.0; tran[3 * g + i] = r.picked >= 0 ? r.tran[i] : 0.0; }
      if (r.picked < 0) st[g] = SBA_ERR_NUMERIC;
    }
  };
  const int nt = std::max(1, std::min(sba::host_threads(), B));
  if (nt == 1) {
    run(0, B);
  } else {
    std::vector<std::thread> pool;
    for (int t = 1; t < nt; ++t)
      pool.emplace_back(run, static_cast<int>(static_cast<long long>(B) * t / nt), static_cast<int>(static_cast<long long>(B) * (t + 1) / nt));
    run(0, static_cast<int>(static_cast<long long>(B) / nt));
    for (auto& th : pool) th.join();
  }
  int failures = 0;
  for (int g = 0; g < B; ++g) { if (status) status[g] = st[g]; if (st[g] != SBA_OK) ++failures; }
  if (failures) return sba::set_error(SBA_ERR_NUMERIC, "%d of %d pairs have no valid rotation candidate (see per-pair status)", failures, B);
  return SBA_OK;
}

// The reference's whole per-pair pipeline for every pair of the batch (do_bundle_adjustment -> initial_guess -> solve_problem,
// spherical_bundle_adjuster.cpp:302-331, :183-217): 8-point consensus guess, init_rot = -R_vec_out, init_tran = T_vec_out,
// d-only stage, rot-only and tran-only stages with the first two refined depths as the depths of every match.  Four
// launches for the whole batch (moments, trials + consensus, d-only stage, and one per-pair LM launch per remaining stage),
// no per-pair host work beyond copying records.
int sba_batch_solve_problem(sba_batch* b, int use_initial_guess, int trials, double subset_fraction, unsigned long long seed,
                            double* rot, double* tran, const sba_lm_options* opt, double* d12_out, double* d_uniform,
                            int* guess_candidates, sba_lm_summary* depth_summaries, sba_lm_summary* rot_summaries,
                            sba_lm_summary* tran_summaries, int* status) {
  if (!b) return sba::set_error(SBA_ERR_INVALID_ARG, "null batch handle");
  SBA_REFUSE_POISONED(b);
  if (!b->uploaded) return sba::set_error(SBA_ERR_NOT_UPLOADED, "no pairs uploaded");
  const int B = b->num_pairs;
  if (B == 0) return SBA_OK;
  if (!rot || !tran) return sba::set_error(SBA_ERR_INVALID_ARG, "rot/tran must not be null");
  if (!b->has_d12) return sba::set_error(SBA_ERR_INVALID_ARG, "the pipeline needs per-match depths uploaded (init_d)");
  std::vector<int> st(B, SBA_OK), stage_st(B, SBA_OK);
  auto merge = [&](int rc) -> int {       // per-pair failures are collected; anything else ends the call
    if (rc != SBA_OK && rc != SBA_ERR_NUMERIC) return rc;
    for (int g = 0; g < B; ++g) if (st[g] == SBA_OK && stage_st[g] != SBA_OK) st[g] = stage_st[g];
    return SBA_OK;
  };
  if (use_initial_guess) {
    std::vector<double> euler(3 * B), tvec(3 * B);
    std::fill(stage_st.begin(), stage_st.end(), SBA_OK);
    const int rc = merge(sba_batch_initial_guess(b, trials, subset_fraction, seed, euler.data(), tvec.data(), guess_candidates, stage_st.data()));
    if (rc) return rc;
    for (int g = 0; g < B; ++g) {
      if (stage_st[g] != SBA_OK) continue;           // no candidate: the pair keeps the caller's start values
      for (int a = 0; a < 3; ++a) { rot[3 * g + a] = -euler[3 * g + a]; tran[3 * g + a] = tvec[3 * g + a]; }   // .cpp:330-331
    }
  } else if (guess_candidates) {
    std::fill(guess_candidates, guess_candidates + B, 0);
  }
  std::vector<double> du(2 * B);
  std::fill(stage_st.begin(), stage_st.end(), SBA_OK);
  int rc = merge(batch_solve_depths_impl(b, rot, tran, 1.0, 1.0, opt, d12_out, depth_summaries, stage_st.data(), du.data()));   // .cpp:196-197, :1057-1058
  if (rc) return rc;
  std::vector<double> d1(B), d2(B);
  for (int g = 0; g < B; ++g) { d1[g] = du[2 * g]; d2[g] = du[2 * g + 1]; }
  if (d_uniform) std::memcpy(d_uniform, du.data(), sizeof(double) * 2 * B);
  std::fill(stage_st.begin(), stage_st.end(), SBA_OK);
  rc = merge(sba_batch_solve(b, SBA_MODE_ROT, SBA_DEPTH_UNIFORM, rot, tran, d1.data(), d2.data(), opt, rot_summaries, stage_st.data()));    // .cpp:202-203
  if (rc) return rc;
  std::fill(stage_st.begin(), stage_st.end(), SBA_OK);
  rc = merge(sba_batch_solve(b, SBA_MODE_TRAN, SBA_DEPTH_UNIFORM, rot, tran, d1.data(), d2.data(), opt, tran_summaries, stage_st.data()));  // .cpp:208-209
  if (rc) return rc;
  int failures = 0;
  for (int g = 0; g < B; ++g) { if (status) status[g] = st[g]; if (st[g] != SBA_OK) ++failures; }
  if (failures) return sba::set_error(SBA_ERR_NUMERIC, "%d of %d pairs failed in a stage of the pipeline (see per-pair status)", failures, B);
  return SBA_OK;
}

int sba_batch_solve(sba_batch* b, int mode, int depth_mode, double* rot, double* tran, const double* d1,
                    const double* d2, const sba_lm_options* opt, sba_lm_summary* summaries, int* status) {
  int rc = check_batch_args(b, mode, depth_mode, rot, tran);
  if (rc) return rc;
  const int B = b->num_pairs;
  if (B == 0) return SBA_OK;
  SBA_TRY_HIP(hipSetDevice(b->device));
  sba_lm_options o;
  if (opt) o = *opt; else sba::lm_default_options(&o);
  o.verbose = 0;   // 256 interleaved progress tables help nobody
  // One block per pair (at least one pair per CU: config C5): a pair's sweep is reduced entirely inside its block, so its
  // whole Levenberg-Marquardt solve runs on the device in ONE launch (batch_lm_kernel: the same LmSolver source, one
  // solver per pair, no host round trip, no lock-step).  SBA_BATCH_DEVICE_LM=0 keeps the host lock-step loop below,
  // which also serves batches that spread a pair over several blocks.
  bool device_lm = b->bpp == 1;
  if (const char* env = std::getenv("SBA_BATCH_DEVICE_LM")) device_lm = device_lm && std::strcmp(env, "0") != 0;
  // Device-resident LM with DYNAMIC shares (one launch triple per iteration, the blocks dealt out to the pairs that are still
  // iterating): serves any number of blocks per pair.  SBA_BATCH_DYNAMIC=1 selects it (see DESIGN.md section 3.5 for when it wins).
  // Default with one block per pair (hybrid): the one-launch kernel runs the first sweeps of every pair -- enough for most --
  // and hands the pairs that need more over to the dynamic launches, where the CUs of the finished pairs join in.
  // SBA_BATCH_DYNAMIC=1: dynamic launches from the first sweep whatever the blocks per pair; =0: the one-launch kernel to the end
  // (one block per pair) or the host lock-step loop (several).  SBA_BATCH_LM_FIRST_SWEEPS: the cap of the first launch (default 6).
  // Several blocks per pair (fewer pairs than CUs): dynamic launches from the first sweep -- device-resident, no host step per
  // iteration (0.36 / 0.47 ms against 0.42 / 0.54 ms for the host lock-step loop at 4 x 10^6 / 128 x 50 000 matches, all pairs
  // needing the same 3 iterations; tools/few_pairs_lm_ab.py).
  bool dynamic = b->bpp > 1 && b->publish, hybrid = device_lm && b->publish;
  if (const char* env = std::getenv("SBA_BATCH_DYNAMIC")) { dynamic = b->publish && std::strcmp(env, "0") != 0; hybrid = false; }
  int first_sweeps = 6;
  if (const char* env = std::getenv("SBA_BATCH_LM_FIRST_SWEEPS")) { const int v = std::atoi(env); if (v >= 1 && v <= 1000) first_sweeps = v; }
  if (device_lm || dynamic) {
    const auto t0 = std::chrono::steady_clock::now();
    // start points go to the device through mapped pinned memory (thread 0 of every block reads its own record once,
    // and writes the result back there at the end): no allocation, no copies -- one launch and one synchronise
    sba::BatchLmIo* io = b->lm_io_host;
    for (int g = 0; g < B; ++g) {
      for (int a = 0; a < 3; ++a) { io[g].rot[a] = rot[3 * g + a]; io[g].tran[a] = tran[3 * g + a]; }
      io[g].d1 = d1 ? d1[g] : 1.0;
      io[g].d2 = d2 ? d2[g] : 1.0;
      io[g].summary = sba_lm_summary{};
      io[g].status = SBA_ERR_NUMERIC;
      io[g].pad_ = 0;
    }
    sba::Planes pl;
    for (int k = 0; k < 3; ++k) { pl.x1[k] = b->coord[k]; pl.x2[k] = b->coord[3 + k]; }
    pl.d1 = b->dplane[0]; pl.d2 = b->dplane[1];
    // completion word: the slot after the packs in the mapped host buffer (the same one the batched step publishes to)
    volatile unsigned long long* flag = reinterpret_cast<volatile unsigned long long*>(b->packs_host + 24 * B);
    unsigned long long* flag_dev = reinterpret_cast<unsigned long long*>(b->packs_host_dev + 24 * B);
    // The remaining iterations as launches with dynamic shares, enqueued in groups of four without waiting; after each group
    // the host learns from the compaction kernel's publication how many pairs are still iterating.  A launch without active
    // pairs is three empty kernels.  first_parity: which active list the first pass reads.
    int per_cu = 1;
    if (const char* env = std::getenv("SBA_BATCH_DYN_BLOCKS_PER_CU")) { const int v = std::atoi(env); if (v >= 1 && v <= 8) per_cu = v; }
    const int sweep_grid = std::max(1, b->num_cus * per_cu);
    auto run_dynamic = [&](int first_parity, unsigned long long remaining) -> int {
      const int max_passes = std::max(0, o.max_num_iterations) + 8;
      volatile unsigned long long* words = b->dyn_host;
      int pass = 0;
      while (remaining > 0 && pass < max_passes) {
        const int group = std::min(4, max_passes - pass);
        unsigned long long seq = 0;
        for (int k = 0; k < group; ++k, ++pass) {
          const bool last = k == group - 1;
          if (last) seq = ++b->dyn_seq;
          SBA_TRY_HIP(sba::launch_batch_lm_dyn_pass(mode, depth_mode, b->store, b->kind, pl, b->desc_dev, o, B, (first_parity + pass) & 1,
                                                    sweep_grid, b->dyn_state, b->params_dev, b->frames_dev, b->dyn_ctl, b->dyn_active,
                                                    b->dyn_done, b->dyn_partials, b->lm_io_host_dev, last ? b->dyn_host_dev : nullptr, seq,
                                                    b->stream));
        }
        const int wrc = sba::wait_for_sequence(words + 1, seq, b->stream, "batched per-pair solve (dynamic shares)", &b->poisoned);
        if (wrc) return wrc;
        remaining = words[0];
      }
      return SBA_OK;      // pairs that ran out of launches keep SBA_ERR_NUMERIC (cannot happen: the solver's limit ends it first)
    };
    if (dynamic || hybrid) {
      const int rc2 = ensure_dyn(b, static_cast<size_t>(sba::kDynOver) * static_cast<size_t>(sweep_grid) + 2 * static_cast<size_t>(B));
      if (rc2) return rc2;
    }
    if (dynamic) {
      SBA_TRY_HIP(sba::launch_batch_lm_dyn_init(mode, depth_mode, b->kind, b->desc_dev, b->lm_io_host_dev, o, B, b->dyn_state,
                                                b->params_dev, b->frames_dev, b->dyn_ctl, b->dyn_active, b->dyn_done, b->stream));
      const int rc2 = run_dynamic(0, static_cast<unsigned long long>(B));
      if (rc2) return rc2;
    } else if (b->publish) {
      // One launch for the first sweeps of every pair (hybrid: at most first_sweeps of them; all of them otherwise) ...
      const unsigned long long seq = ++b->seq;
      SBA_TRY_HIP(sba::launch_batch_lm(mode, depth_mode, b->store, b->kind, pl, b->desc_dev, b->lm_io_host_dev, o, B,
                                       b->lm_ticket, flag_dev, seq, b->stream, hybrid ? first_sweeps : 0, hybrid ? b->dyn_state : nullptr,
                                       hybrid ? b->params_dev : nullptr, hybrid ? b->frames_dev : nullptr, hybrid ? b->dyn_done : nullptr));
      const int wrc = sba::wait_for_sequence(flag, seq, b->stream, "batched per-pair solve", &b->poisoned);
      if (wrc) return wrc;
      // ... and the pairs it handed over (record.pad_ == 1) go on with dynamic shares: their CUs are joined by those of the
      // pairs that are done.
      unsigned long long left = 0;
      for (int g = 0; g < B; ++g) if (io[g].pad_ == 1) { ++left; io[g].status = SBA_ERR_NUMERIC; }
      if (left > 0) {
        SBA_TRY_HIP(sba::launch_batch_dyn_first_list(b->dyn_ctl, b->dyn_active, b->dyn_done, B, b->stream));
        const int rc2 = run_dynamic(1, left);
        if (rc2) return rc2;
      }
    } else {
      SBA_TRY_HIP(sba::launch_batch_lm(mode, depth_mode, b->store, b->kind, pl, b->desc_dev, b->lm_io_host_dev, o, B,
                                       b->lm_ticket, nullptr, 0, b->stream));
      { const int _rc = sba::stream_wait(b->stream, "stream synchronisation", &b->poisoned); if (_rc) return _rc; }
    }
    const double seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    int failures = 0;
    for (int g = 0; g < B; ++g) {
      for (int a = 0; a < 3; ++a) { rot[3 * g + a] = io[g].rot[a]; tran[3 * g + a] = io[g].tran[a]; }
      if (summaries) { summaries[g] = io[g].summary; summaries[g].seconds_total = seconds; }   // wall clock of the whole batch
      if (status) status[g] = io[g].status;
      if (io[g].status != SBA_OK) ++failures;
    }
    if (failures) return sba::set_error(SBA_ERR_NUMERIC, "%d of %d pairs failed (see per-pair status)", failures, B);
    return SBA_OK;
  }
  std::vector<sba::LmSolver> solver(B);
  std::vector<unsigned char> active(B, 1);
  std::vector<double> qrot(3 * B), qtran(3 * B);
  for (int g = 0; g < B; ++g) solver[g].start(mode, rot + 3 * g, tran + 3 * g, o);
  // per-pair host work of one lock-step iteration: feed the pair's normal equations to its solver, fetch its next query
  std::vector<int> newly_done(B, 0);
  auto feed_range = [&](int first, int last) {
    for (int g = first; g < last; ++g) {
      if (!active[g]) continue;
      sba_normal_eq ne;
      sba::expand_pack(mode, b->packs_host + 24 * g, &ne);
      solver[g].feed(ne);
      if (solver[g].done()) newly_done[g] = 1;
      for (int a = 0; a < 3; ++a) { qrot[3 * g + a] = solver[g].query_rot()[a]; qtran[3 * g + a] = solver[g].query_tran()[a]; }
    }
  };
  PairWorkers workers(sba::host_threads(), B, feed_range);
  for (int g = 0; g < B; ++g)
    for (int a = 0; a < 3; ++a) { qrot[3 * g + a] = solver[g].query_rot()[a]; qtran[3 * g + a] = solver[g].query_tran()[a]; }
  int remaining = B;
  while (remaining > 0) {
    rc = batch_launch(b, mode, depth_mode, qrot.data(), qtran.data(), d1, d2, o.huber_delta, active.data());
    if (rc) return rc;
    workers.run();
    for (int g = 0; g < B; ++g)
      if (newly_done[g]) { newly_done[g] = 0; active[g] = 0; --remaining; }
  }
  int failures = 0;
  for (int g = 0; g < B; ++g) {
    for (int a = 0; a < 3; ++a) { rot[3 * g + a] = solver[g].rot()[a]; tran[3 * g + a] = solver[g].tran()[a]; }
    if (summaries) summaries[g] = solver[g].summary();
    if (status) status[g] = solver[g].status();
    if (solver[g].status() != SBA_OK) ++failures;
  }
  if (failures) return sba::set_error(SBA_ERR_NUMERIC, "%d of %d pairs failed (see per-pair status)", failures, B);
  return SBA_OK;
}

}  // extern "C"
