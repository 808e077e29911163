// C-ABI shim (include/sba_hip.h) over the HIP kernels: errors, handle life cycle, uploads, sweeps, eval / solve entry
// points and the side entry points.  Transports: sba_transport.cpp; d-only stage and initial guess: sba_stages.cpp.
// Host code only; compiled with hipcc for the HIP runtime API.  No CPU fallback exists here:
// every compute entry point needs a HIP device and fails with SBA_ERR_NO_DEVICE otherwise.
#include "../../include/sba_hip.h"

#include <dlfcn.h>
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <thread>
#include <chrono>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>
#include <vector>

#include "sba_device.hpp"
#include "sba_epipolar.hpp"
#include "sba_internal.hpp"
#include "sba_problem.hpp"
#include "sba_lm.hpp"
#include "sba_rotation.hpp"

namespace {

// Host threads for the host-side loops that have any (the 80 trials of the initial guess): sba_set_host_threads,
// initialised from SBA_HOST_THREADS; 1 = serial.  Results do not depend on it.
std::atomic<int> g_host_threads{[] {
  const char* env = std::getenv("SBA_HOST_THREADS");
  const int v = env ? std::atoi(env) : 1;
  return v >= 1 && v <= 256 ? v : 1;
}()};

thread_local std::string g_last_error;

int fail(int code, const char* fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof(buf), fmt, ap);
  va_end(ap);
  g_last_error = buf;
  return code;
}
}  // namespace

namespace sba {
int host_threads() { return g_host_threads.load(); }
namespace {
double wait_limit_seconds() {
  static const double limit_s = [] {
    const char* env = std::getenv("SBA_WAIT_TIMEOUT_S");
    const double v = env ? std::atof(env) : 60.0;
    return v > 0.0 ? v : 60.0;
  }();
  return limit_s;
}
}  // namespace
int wait_for_sequence(const volatile unsigned long long* flag, unsigned long long seq, hipStream_t stream,
                      const char* what, int* poisoned) {
  // Wall-clock bound (SBA_WAIT_TIMEOUT_S, default 60 s): a device that never publishes -- a wedged kernel keeps the
  // stream at hipErrorNotReady for ever -- must come back as an error, not as a spinning host thread.  Giving up (or a
  // device error) poisons the handle: nothing may wait on that stream again, not even its destroy.
  const double limit_s = wait_limit_seconds();
  std::chrono::steady_clock::time_point t0;
  bool timing = false;
  for (unsigned long spins = 0; *flag != seq; ++spins) {
    if ((spins & 0xfff) == 0xfff) {
      const hipError_t q = hipStreamQuery(stream);
      if (q != hipSuccess && q != hipErrorNotReady) {
        if (poisoned) *poisoned = 1;
        return set_error(SBA_ERR_HIP, "%s failed on the device: %s", what, hipGetErrorString(q));
      }
      if (q == hipSuccess && *flag != seq) {
        if (poisoned) *poisoned = 1;    // the protocol state (sequence numbers, tickets) can no longer be trusted
        return set_error(SBA_ERR_HIP, "%s finished without publishing its result", what);
      }
      const auto now = std::chrono::steady_clock::now();
      if (!timing) { t0 = now; timing = true; }
      else if (std::chrono::duration<double>(now - t0).count() > limit_s) {
        if (poisoned) *poisoned = 1;
        return set_error(SBA_ERR_HIP, "%s: no result from the device after %.0f s (SBA_WAIT_TIMEOUT_S)", what, limit_s);
      }
    }
    __builtin_ia32_pause();
  }
  __atomic_thread_fence(__ATOMIC_ACQUIRE);
  return SBA_OK;
}
int stream_wait(hipStream_t stream, const char* what, int* poisoned) {
  const double limit_s = wait_limit_seconds();
  std::chrono::steady_clock::time_point t0;
  bool timing = false;
  for (unsigned long spins = 0;; ++spins) {
    const hipError_t q = hipStreamQuery(stream);
    if (q == hipSuccess) return SBA_OK;
    if (q != hipErrorNotReady) {
      if (poisoned) *poisoned = 1;
      return set_error(SBA_ERR_HIP, "%s failed on the device: %s", what, hipGetErrorString(q));
    }
    if (spins >= 256) {     // ~0.3 ms of polling, then yield the core between queries and watch the clock
      const auto now = std::chrono::steady_clock::now();
      if (!timing) { t0 = now; timing = true; }
      else if (std::chrono::duration<double>(now - t0).count() > limit_s) {
        if (poisoned) *poisoned = 1;
        return set_error(SBA_ERR_HIP, "%s: the stream did not drain within %.0f s (SBA_WAIT_TIMEOUT_S)", what, limit_s);
      }
      std::this_thread::sleep_for(std::chrono::microseconds(spins < 4096 ? 20 : 500));
    } else {
      __builtin_ia32_pause();
    }
  }
}
int event_wait(hipEvent_t ev, const char* what, int* poisoned) {
  const double limit_s = wait_limit_seconds();
  std::chrono::steady_clock::time_point t0;
  bool timing = false;
  for (unsigned long spins = 0;; ++spins) {
    const hipError_t q = hipEventQuery(ev);
    if (q == hipSuccess) return SBA_OK;
    if (q != hipErrorNotReady) {
      if (poisoned) *poisoned = 1;
      return set_error(SBA_ERR_HIP, "%s failed on the device: %s", what, hipGetErrorString(q));
    }
    if (spins >= 256) {
      const auto now = std::chrono::steady_clock::now();
      if (!timing) { t0 = now; timing = true; }
      else if (std::chrono::duration<double>(now - t0).count() > limit_s) {
        if (poisoned) *poisoned = 1;
        return set_error(SBA_ERR_HIP, "%s: not reached within %.0f s (SBA_WAIT_TIMEOUT_S)", what, limit_s);
      }
      std::this_thread::sleep_for(std::chrono::microseconds(spins < 4096 ? 20 : 500));
    } else {
      __builtin_ia32_pause();
    }
  }
}

struct CopyPool::Impl {
  std::vector<std::thread> pool;
  std::atomic<unsigned long long> gen{0};
  std::atomic<int> done{0};
  std::atomic<bool> quit{false};
  char* dst = nullptr;
  const char* src = nullptr;
  size_t bytes = 0;
  int nt = 1;
  void share(int t) const {
    const size_t per = ((bytes + nt - 1) / nt + 63) & ~size_t(63);
    const size_t lo = std::min(bytes, per * t), hi = std::min(bytes, per * (t + 1));
    if (hi > lo) std::memcpy(dst + lo, src + lo, hi - lo);
  }
  void loop(int t) {
    unsigned long long seen = 0;
    for (;;) {
      unsigned long spins = 0;
      while (gen.load(std::memory_order_acquire) == seen) {
        if (++spins > 20000) std::this_thread::yield(); else __builtin_ia32_pause();
      }
      ++seen;
      if (quit.load(std::memory_order_acquire)) return;
      share(t);
      done.fetch_add(1, std::memory_order_acq_rel);
    }
  }
};
CopyPool::CopyPool(int threads) : impl_(new Impl()), nt_(std::max(1, threads)) {
  impl_->nt = nt_;
  for (int t = 1; t < nt_; ++t) impl_->pool.emplace_back([this, t] { impl_->loop(t); });
}
CopyPool::~CopyPool() {
  impl_->quit.store(true, std::memory_order_release);
  impl_->gen.fetch_add(1, std::memory_order_acq_rel);
  for (auto& th : impl_->pool) th.join();
  delete impl_;
}
void CopyPool::copy(void* dst, const void* src, size_t bytes) {
  impl_->dst = static_cast<char*>(dst); impl_->src = static_cast<const char*>(src); impl_->bytes = bytes;
  impl_->done.store(0, std::memory_order_relaxed);
  impl_->gen.fetch_add(1, std::memory_order_acq_rel);
  impl_->share(0);
  while (impl_->done.load(std::memory_order_acquire) != nt_ - 1) __builtin_ia32_pause();
}

int set_error(int code, const char* fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof(buf), fmt, ap);
  va_end(ap);
  g_last_error = buf;
  return code;
}
void make_sweep_params(size_t n, int depth_mode, const double rot[3], const double tran[3], double d1, double d2,
                       double huber_delta, SweepParams* prm) {
  static_assert(SBA_DEPTH_UNIFORM == 0, "fill_sweep_params takes 0 for uniform depths");
  fill_sweep_params(n, depth_mode, rot, tran, d1, d2, huber_delta, prm);   // same source as the batched path's device side
}
}  // namespace sba

namespace {
using sba::shim::allreduce_pack;
using sba::shim::kNcclFloat64;
using sba::shim::kNcclSum;
using sba::shim::Rccl;
using sba::shim::rccl;

#define SBA_HIP_TRY(expr)                                                                   \
  do {                                                                                      \
    hipError_t _e = (expr);                                                                 \
    if (_e != hipSuccess)                                                                   \
      return fail(SBA_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, \
                  __LINE__);                                                                \
  } while (0)

#define SBA_SYNC(p, what)                                                   \
  do {                                                                      \
    const int _rc = sba::stream_wait((p)->stream, what, &(p)->poisoned);    \
    if (_rc) return _rc;                                                    \
  } while (0)

}  // namespace


namespace {

int free_planes(sba_problem* p) {
  for (int k = 0; k < 8; ++k) {
    if (p->plane_base[k]) SBA_HIP_TRY(hipFree(p->plane_base[k]));
    p->plane_base[k] = nullptr;
    p->plane_bytes[k] = 0;
  }
  for (auto& c : p->coord) c = nullptr;
  for (auto& d : p->dplane) d = nullptr;
  p->uploaded = false;
  p->n = 0;
  p->plane_elems = 0;
  p->has_d12 = false;
  return SBA_OK;
}

// Plane k: at least `bytes` bytes, zeroed.  An existing allocation that is large enough -- and not more than four times too
// large -- is kept (the usual case: one handle fed image pair after image pair of similar size).
int ensure_plane(sba_problem* p, int k, size_t bytes) {
  if (!p->plane_base[k] || p->plane_bytes[k] < bytes || p->plane_bytes[k] > 4 * bytes + (size_t(1) << 20)) {
    if (p->plane_base[k]) SBA_HIP_TRY(hipFree(p->plane_base[k]));
    p->plane_base[k] = nullptr;
    p->plane_bytes[k] = 0;
    SBA_HIP_TRY(hipMalloc(&p->plane_base[k], bytes));
    p->plane_bytes[k] = bytes;
  }
  SBA_HIP_TRY(hipMemsetAsync(p->plane_base[k], 0, bytes, p->stream));
  return SBA_OK;
}

int alloc_planes(sba_problem* p, size_t n, bool with_d12, int store) {
  p->uploaded = false;
  const size_t esz = store == SBA_STORE_F64 ? 8 : 4;
  // whole 16-byte vectors, plus one spare vector so that the ragged tail load stays in bounds
  const size_t ppt = static_cast<size_t>(sba::points_per_lane(store));
  const size_t elems = ((n + ppt - 1) / ppt + 1) * ppt;
  // Each plane sits at offset k * plane_stagger inside its own allocation (16-byte aligned), so that equal element
  // indices of different planes do not share the low address bits.
  const size_t pad = 8 * p->plane_stagger;
  for (int k = 0; k < 6; ++k) {
    const int rc = ensure_plane(p, k, elems * esz + pad);
    if (rc) return rc;
    p->coord[k] = static_cast<char*>(p->plane_base[k]) + k * p->plane_stagger;
  }
  for (int k = 0; k < 2; ++k) {
    if (with_d12) {
      const int rc = ensure_plane(p, 6 + k, elems * 8 + pad);
      if (rc) return rc;
      p->dplane[k] = reinterpret_cast<double*>(static_cast<char*>(p->plane_base[6 + k]) + (6 + k) * p->plane_stagger);
    } else {
      if (p->plane_base[6 + k]) SBA_HIP_TRY(hipFree(p->plane_base[6 + k]));
      p->plane_base[6 + k] = nullptr; p->plane_bytes[6 + k] = 0; p->dplane[k] = nullptr;
    }
  }
  p->n = n;
  p->store = store;
  p->has_d12 = with_d12;
  p->plane_elems = elems;
  return SBA_OK;
}

// One resident wave of blocks: min(blocks needed, CUs x resident blocks per CU of this kernel).
int grid_for(sba_problem* p, int mode, int depth_mode, bool loss, int* grid) {
  const size_t ppt = static_cast<size_t>(sba::points_per_lane(p->store));
  const size_t nvec = (p->n + ppt - 1) / ppt;
  const size_t want = (nvec + sba::kBlock - 1) / sba::kBlock;
  int& occ = p->occ_cache[mode][depth_mode][p->store][p->kind][loss ? 1 : 0];
  if (occ == 0) {
    int b = 0;
    SBA_HIP_TRY(sba::sweep_blocks_per_cu(mode, depth_mode, p->store, p->kind, loss, &b));
    // Resident blocks per CU actually used.  Measured per variant at 10^7 matches (profiles/r01_tune_caps.log, re-checked
    // A/B on one box): the f64 factored kernel streaming 8 planes (R|t with per-match depths: 96.7 / 96.7 / 95.3 us at
    // one block per CU against 99.6 / 99.0 / 97.4 us at two) and its tran-only form finish tighter with ONE block per
    // CU -- fewer resident blocks, less finish-time spread between XCDs (profiles/r01_stream_probe.md) -- every other
    // variant (6 planes, f32 planes, explicit Jacobian) needs the second block to cover its arithmetic.
    int cap = p->blocks_per_cu_cap;
    if (cap <= 0)
      cap = (p->store == SBA_STORE_F64 && p->kind == SBA_KERNEL_FACTORED &&
             ((mode == SBA_MODE_RT && depth_mode == SBA_DEPTH_PER_MATCH) || mode == SBA_MODE_TRAN)) ? 1 : 2;
    occ = std::max(1, std::min(b, cap));
  }
  *grid = static_cast<int>(std::min<size_t>(want, static_cast<size_t>(std::min(p->max_grid, p->num_cus * occ))));
  return SBA_OK;
}

int check_args(const sba_problem* p, int mode, int depth_mode, const double* rot, const double* tran) {
  if (!p) return fail(SBA_ERR_INVALID_ARG, "null problem handle");
  SBA_REFUSE_POISONED(p);
  if (!rot || !tran) return fail(SBA_ERR_INVALID_ARG, "rot/tran must not be null");
  if (mode < SBA_MODE_ROT || mode > SBA_MODE_RT) return fail(SBA_ERR_INVALID_ARG, "bad mode %d", mode);
  if (depth_mode != SBA_DEPTH_UNIFORM && depth_mode != SBA_DEPTH_PER_MATCH)
    return fail(SBA_ERR_INVALID_ARG, "bad depth_mode %d", depth_mode);
  if (!p->uploaded) return fail(SBA_ERR_NOT_UPLOADED, "no correspondences uploaded");
  if (depth_mode == SBA_DEPTH_PER_MATCH && !p->has_d12 && p->n > 0)
    return fail(SBA_ERR_INVALID_ARG, "per-match depths requested but none were uploaded");
  for (int i = 0; i < 3; ++i)
    if (!std::isfinite(rot[i]) || !std::isfinite(tran[i]))
      return fail(SBA_ERR_INVALID_ARG, "non-finite rot/tran");
  return SBA_OK;
}

void make_params(const sba_problem* p, int depth_mode, const double rot[3], const double tran[3],
                 double d1, double d2, double huber_delta, sba::SweepParams* prm) {
  sba::make_sweep_params(p->n, depth_mode, rot, tran, d1, d2, huber_delta, prm);
}

void make_frame(sba_problem* p, int mode, const double rot[3]) {
  sba::factored_frame(rot, p->frame_B, p->frame_J);
  p->last_mode = mode;
}

// Enqueue one sweep + finalize (+ all-reduce) on the problem's stream; pack_dev holds the result.
int enqueue_sweep(sba_problem* p, int mode, int depth_mode, const sba::SweepParams& prm) {
  sba::Planes pl;
  for (int k = 0; k < 3; ++k) {
    pl.x1[k] = p->coord[k];
    pl.x2[k] = p->coord[3 + k];
  }
  pl.d1 = p->dplane[0];
  pl.d2 = p->dplane[1];
  int grid = 0;
  int rc0 = grid_for(p, mode, depth_mode, prm.delta > 0.0, &grid);
  if (rc0) return rc0;
  // Final reduction + hand-over.  Default: the sweep stays a pure streaming kernel and a one-block finalize kernel
  // folds the block rows and publishes the pack into mapped pinned host memory, followed by a sequence number the
  // host polls -- no blit kernel for a 192-byte D2H, no stream synchronisation.  Optional (SBA_FUSED=1/2): the
  // sweep's last-arriving block does the fold and the publication itself; measured equal in step time at 10^7 and at
  // 2 048 matches, but it lengthens the dominant kernel by ~5 us, so it is not the default.  With an all-reduce to
  // follow, the pack stays on the device and publish_kernel hands it over afterwards.
  constexpr int kFusedMaxGrid = 128;
  const bool collective = p->comm != nullptr || p->hook != nullptr || p->peer_ready;
  const bool fused = grid > 0 && (p->fused_mode == 1 || (p->fused_mode == 2 && grid <= kFusedMaxGrid));
  const bool to_host = p->publish && !collective;
  sba::SweepOut out;
  out.partials = p->partials;
  out.pack_dev = p->pack_dev;
  out.ticket = fused ? p->ticket : nullptr;
  p->published = to_host;
  if (to_host) ++p->seq;
  out.pack_host = (fused && to_host) ? p->pack_host_dev : nullptr;
  out.seq = p->seq;
  SBA_HIP_TRY(sba::launch_sweep(mode, depth_mode, p->store, p->kind, pl, prm, out, grid, p->stream));
  if (p->peer_ready) {
    // all-reduce by direct peer stores over xGMI + rank-ordered local sum; the exchanging wave publishes to the host.
    // Two-kernel mode: the finalize kernel's wave 0 does fold, exchange and publication in ONE launch.
    if (p->publish) { ++p->seq; p->published = true; }
    double* host = p->publish ? p->pack_host_dev : nullptr;
    if (!fused)
      SBA_HIP_TRY(sba::launch_finalize(p->partials, grid, p->pack_dev, host, p->seq, &p->peers, ++p->xseq,
                                       p->stream));
    else
      SBA_HIP_TRY(sba::launch_peer_exchange(p->pack_dev, p->peers, ++p->xseq, p->pack_dev, host, p->seq,
                                            p->stream));
    return SBA_OK;
  }
  if (!fused)
    SBA_HIP_TRY(sba::launch_finalize(p->partials, grid, p->pack_dev, to_host ? p->pack_host_dev : nullptr, p->seq,
                                     nullptr, 0, p->stream));
  if (collective) return allreduce_pack(p);
  return SBA_OK;
}

}  // namespace

namespace sba {
namespace shim {
// Wait for the reduced pack on the host; the factored kernel's moments are mapped to the SBA_PACK_* layout.
int fetch_pack_raw(sba_problem* p, double raw[SBA_PACK_SIZE]) {
  if (p->published) {
    const int rc = sba::wait_for_sequence(reinterpret_cast<volatile unsigned long long*>(p->pack_host + 24), p->seq,
                                          p->stream, "sweep", &p->poisoned);
    if (rc) return rc;
    if (p->peer_ready && reinterpret_cast<volatile unsigned long long*>(p->pack_host)[25] != 0)
      return fail(SBA_ERR_COMM, "peer exchange timed out waiting for another rank's pack");
  } else {
    SBA_HIP_TRY(hipMemcpyAsync(p->pack_host, p->pack_dev, SBA_PACK_SIZE * sizeof(double),
                               hipMemcpyDeviceToHost, p->stream));
    SBA_SYNC(p, "stream synchronisation");
  }
  std::memcpy(raw, p->pack_host, SBA_PACK_SIZE * sizeof(double));
  return SBA_OK;
}

}  // namespace shim
}  // namespace sba

namespace {
using sba::shim::fetch_pack_raw;

// ... and map the factored kernel's moments to the SBA_PACK_* layout.
int fetch_pack(sba_problem* p, double pack[SBA_PACK_SIZE]) {
  double raw[SBA_PACK_SIZE];
  const int rc = fetch_pack_raw(p, raw);
  if (rc) return rc;
  if (p->kind == SBA_KERNEL_FACTORED && p->last_mode != SBA_MODE_TRAN)
    sba::moments_to_normal_pack(true, p->last_mode == SBA_MODE_RT, p->frame_B, p->frame_J, raw, pack);
  else
    std::memcpy(pack, raw, SBA_PACK_SIZE * sizeof(double));
  return SBA_OK;
}

}  // namespace

// =================================================================================================
extern "C" {

int sba_abi_version(void) { return SBA_ABI_VERSION; }

int sba_set_host_threads(int n) {
  if (n < 0) return fail(SBA_ERR_INVALID_ARG, "thread count must be >= 0 (0 = one per hardware thread, at most 16)");
  if (n == 0) n = static_cast<int>(std::min(16u, std::max(1u, std::thread::hardware_concurrency())));
  g_host_threads.store(n);
  return SBA_OK;
}
const char* sba_last_error(void) { return g_last_error.c_str(); }

int sba_device_count(int* count) {
  if (!count) return fail(SBA_ERR_INVALID_ARG, "count is null");
  int c = 0;
  const hipError_t e = hipGetDeviceCount(&c);
  if (e != hipSuccess) {
    *count = 0;
    return fail(SBA_ERR_NO_DEVICE, "hipGetDeviceCount: %s", hipGetErrorString(e));
  }
  *count = c;
  return SBA_OK;
}

void sba_lm_options_default(sba_lm_options* opt) {
  if (opt) sba::lm_default_options(opt);
}

int sba_problem_create(sba_problem** out, int device, void* stream) {
  if (!out) return fail(SBA_ERR_INVALID_ARG, "out is null");
  *out = nullptr;
  int count = 0;
  const hipError_t e = hipGetDeviceCount(&count);
  if (e != hipSuccess || count <= 0)
    return fail(SBA_ERR_NO_DEVICE, "no HIP device available (%s); this library has no CPU path",
                e != hipSuccess ? hipGetErrorString(e) : "device count is 0");
  if (device < 0 || device >= count)
    return fail(SBA_ERR_INVALID_ARG, "device %d out of range [0,%d)", device, count);
  SBA_HIP_TRY(hipSetDevice(device));
  // Owned until the very end: any failing HIP call below releases whatever was created so far (sba_problem_destroy
  // copes with a partially built handle -- every member starts out null).
  struct Guard {
    sba_problem* p;
    ~Guard() { if (p) (void)sba_problem_destroy(p); }
  } guard{new (std::nothrow) sba_problem()};
  sba_problem* p = guard.p;
  if (!p) return fail(SBA_ERR_HIP, "out of host memory");
  p->device = device;
  hipDeviceProp_t prop;
  SBA_HIP_TRY(hipGetDeviceProperties(&prop, device));
  p->num_cus = prop.multiProcessorCount;
  int wall_khz = 0;
  if (hipDeviceGetAttribute(&wall_khz, hipDeviceAttributeWallClockRate, device) == hipSuccess && wall_khz > 0)
    p->wall_clock_khz = wall_khz;
  else
    (void)hipGetLastError();
  if (const char* env = std::getenv("SBA_BLOCKS_PER_CU")) {
    const int v = std::atoi(env);
    if (v >= 1 && v <= 8) p->blocks_per_cu_cap = v;
  }
  if (const char* env = std::getenv("SBA_KERNEL")) {
    if (std::strcmp(env, "explicit") == 0) p->kind = SBA_KERNEL_EXPLICIT;
  }
  std::memset(p->occ_cache, 0, sizeof(p->occ_cache));
  if (const char* env = std::getenv("SBA_PLANE_STAGGER")) {
    const long v = std::atol(env);
    if (v >= 0 && v <= (1 << 20) && v % 16 == 0) p->plane_stagger = static_cast<size_t>(v);
  }
  if (stream) {
    p->stream = static_cast<hipStream_t>(stream);
  } else {
    SBA_HIP_TRY(hipStreamCreateWithFlags(&p->stream, hipStreamNonBlocking));
    p->own_stream = true;
  }
  p->max_grid = std::max(1, p->num_cus * 8);
  SBA_HIP_TRY(hipMalloc(reinterpret_cast<void**>(&p->partials),
                        static_cast<size_t>(p->max_grid) * sba::kRow * sizeof(double)));
  SBA_HIP_TRY(hipMalloc(reinterpret_cast<void**>(&p->pack_dev), 32 * sizeof(double)));
  SBA_HIP_TRY(hipMemset(p->pack_dev, 0, 32 * sizeof(double)));
  SBA_HIP_TRY(hipHostMalloc(reinterpret_cast<void**>(&p->pack_host), 32 * sizeof(double),
                            hipHostMallocMapped | hipHostMallocCoherent));
  std::memset(p->pack_host, 0, 32 * sizeof(double));
  SBA_HIP_TRY(hipHostGetDevicePointer(reinterpret_cast<void**>(&p->pack_host_dev), p->pack_host, 0));
  SBA_HIP_TRY(hipMalloc(reinterpret_cast<void**>(&p->ticket), 9 * 64));
  SBA_HIP_TRY(hipMemset(p->ticket, 0, 9 * 64));
  if (const char* env = std::getenv("SBA_FUSED")) p->fused_mode = std::atoi(env) == 0 ? 0 : (std::atoi(env) == 1 ? 1 : 2);
  if (const char* env = std::getenv("SBA_PUBLISH")) p->publish = std::strcmp(env, "0") != 0;
  SBA_HIP_TRY(hipEventCreate(&p->ev0));
  SBA_HIP_TRY(hipEventCreate(&p->ev1));
  // resident evaluator for small problems: the command record the host writes and a resident kernel polls
  SBA_HIP_TRY(hipHostMalloc(reinterpret_cast<void**>(&p->res_rec), sizeof(sba::ResidentRecord),
                            hipHostMallocMapped | hipHostMallocCoherent));
  std::memset(p->res_rec, 0, sizeof(sba::ResidentRecord));
  SBA_HIP_TRY(hipHostGetDevicePointer(reinterpret_cast<void**>(&p->res_rec_dev), p->res_rec, 0));
  SBA_HIP_TRY(hipHostMalloc(&p->small_rec, sizeof(sba::shim::SmallRecord), hipHostMallocMapped | hipHostMallocCoherent));
  std::memset(p->small_rec, 0, sizeof(sba::shim::SmallRecord));
  SBA_HIP_TRY(hipHostGetDevicePointer(&p->small_rec_dev, p->small_rec, 0));
  if (const char* env = std::getenv("SBA_RESIDENT_MAX_N")) {
    const long v = std::atol(env);
    if (v >= 0) p->resident_max_n = p->resident_max_n_depth = p->one_launch_max_n_depth = static_cast<size_t>(v);
  }
  if (const char* env = std::getenv("SBA_RESIDENT_IDLE_S")) { const double v = std::atof(env); if (v > 0.0 && v <= 10.0) p->resident_idle_s = v; }
  guard.p = nullptr;     // hand over
  *out = p;
  return SBA_OK;
}

int sba_problem_destroy(sba_problem* p) {
  if (!p) return SBA_OK;
  (void)hipSetDevice(p->device);
  // Drain the stream with the bounded wait; a handle that is (or hereby becomes) poisoned keeps its device resources:
  // hipFree / hipHostFree / hipStreamDestroy / ncclCommDestroy / hipIpcCloseMemHandle all wait for the wedged device.
  if (!p->poisoned && p->stream) (void)sba::stream_wait(p->stream, "destroy", &p->poisoned);
  if (p->poisoned) {
    delete p;   // host object only; the mapped pinned pack stays allocated (a wedged kernel may still store into it)
    return fail(SBA_ERR_HIP, "problem destroyed while poisoned: its device memory, stream and communicator were leaked "
                             "(a device wait timed out or the device faulted); the process should exit non-zero");
  }
  if (p->comm) {
    Rccl& r = rccl();
    if (r.ok) r.CommDestroy(p->comm);
  }
  (void)sba_problem_peer_disable(p);
  free_planes(p);
  if (p->partials) (void)hipFree(p->partials);
  if (p->pack_dev) (void)hipFree(p->pack_dev);
  if (p->ticket) (void)hipFree(p->ticket);
  if (p->peer_sticky) (void)hipFree(p->peer_sticky);
  if (p->epi_scratch) (void)hipFree(p->epi_scratch);
  if (p->subset_scratch) (void)hipFree(p->subset_scratch);
  if (p->depth_scratch) (void)hipFree(p->depth_scratch);
  if (p->pack_host) (void)hipHostFree(p->pack_host);
  if (p->res_rec) (void)hipHostFree(p->res_rec);
  if (p->small_rec) (void)hipHostFree(p->small_rec);
  for (void* q : p->upload_pinned) if (q) (void)hipHostFree(q);
  if (p->ev0) (void)hipEventDestroy(p->ev0);
  if (p->ev1) (void)hipEventDestroy(p->ev1);
  if (p->own_stream && p->stream) (void)hipStreamDestroy(p->stream);
  delete p;
  return SBA_OK;
}

constexpr size_t kPipelinedUploadMin = size_t(1) << 21;     // correspondences: below this one hipMemcpyAsync per array is as fast
constexpr size_t kUploadChunk = 699050;                     // correspondences per pipelined chunk (16 MiB of xyz)

static bool upload_pipeline_off() {      // SBA_UPLOAD_PIPELINE=0: the single-buffer path at every size (A/B measurements)
  const char* env = std::getenv("SBA_UPLOAD_PIPELINE");
  return env && env[0] == '0';
}

static int upload_common(sba_problem* p, const void* left, const void* right, const void* d12,
                         size_t n, int store, bool from_device) {
  if (!p) return fail(SBA_ERR_INVALID_ARG, "null problem handle");
  SBA_REFUSE_POISONED(p);
  if (store != SBA_STORE_F64 && store != SBA_STORE_F32)
    return fail(SBA_ERR_INVALID_ARG, "bad store %d", store);
  if (n > 0 && (!left || !right)) return fail(SBA_ERR_INVALID_ARG, "null coordinate array");
  SBA_HIP_TRY(hipSetDevice(p->device));
  int rc = alloc_planes(p, n, d12 != nullptr, store);
  if (rc) return rc;
  if (n > 0) {
    if (from_device) {
      SBA_HIP_TRY(sba::launch_aos_to_planes(static_cast<const double*>(left), n, 0, p->coord[0],
                                            p->coord[1], p->coord[2], store, p->stream));
      SBA_HIP_TRY(sba::launch_aos_to_planes(static_cast<const double*>(right), n, 0, p->coord[3],
                                            p->coord[4], p->coord[5], store, p->stream));
      if (d12)
        SBA_HIP_TRY(sba::launch_d12_to_planes(static_cast<const double*>(d12), n, 0, p->dplane[0],
                                              p->dplane[1], p->stream));
    } else if (n < kPipelinedUploadMin || upload_pipeline_off()) {
      // Small and mid-size problems: one staging buffer, chunks of <= 4M correspondences (96 MB) go H2D then are re-laid
      // out as planes on the device.
      const size_t chunk = std::min<size_t>(n, size_t(4) << 20);
      sba::DeviceBuffer stage_buf(&p->poisoned);
      SBA_HIP_TRY(stage_buf.alloc(chunk * 3 * sizeof(double)));
      double* stage = stage_buf.as<double>();
      const double* src[2] = {static_cast<const double*>(left), static_cast<const double*>(right)};
      for (int side = 0; side < 2; ++side)
        for (size_t first = 0; first < n; first += chunk) {
          const size_t m = std::min(chunk, n - first);
          SBA_HIP_TRY(hipMemcpyAsync(stage, src[side] + 3 * first, m * 3 * sizeof(double),
                                     hipMemcpyHostToDevice, p->stream));
          SBA_HIP_TRY(sba::launch_aos_to_planes(stage, m, first, p->coord[3 * side + 0],
                                                p->coord[3 * side + 1], p->coord[3 * side + 2],
                                                store, p->stream));
          SBA_SYNC(p, "stream synchronisation");
        }
      if (d12)
        for (size_t first = 0; first < n; first += chunk) {
          const size_t m = std::min(chunk, n - first);
          SBA_HIP_TRY(hipMemcpyAsync(stage, static_cast<const double*>(d12) + 2 * first,
                                     m * 2 * sizeof(double), hipMemcpyHostToDevice, p->stream));
          SBA_HIP_TRY(sba::launch_d12_to_planes(stage, m, first, p->dplane[0], p->dplane[1], p->stream));
          SBA_SYNC(p, "stream synchronisation");
        }
    } else {
      // Large problems: the caller's arrays are pageable (std::vector<cv::Point3d>::data()), and a fresh pageable array
      // reaches the device at 16-30 GB/s through hipMemcpy, against 57 GB/s from pinned memory (tools/h2d_probe.cpp,
      // profiles/r03_h2d_probe.log).  So: a few host threads copy chunk k + 1 into one of two pinned staging buffers while
      // the DMA engine moves chunk k to the device and the re-layout kernel turns chunk k - 1 into planes -- three stages
      // in flight, 51-52 GB/s end to end with 4-8 copy threads.  The pinned buffers live in the handle.
      const size_t chunk = kUploadChunk;                        // correspondences per chunk: 16 MiB of xyz
      const size_t chunk_bytes = chunk * 3 * sizeof(double);
      for (int k = 0; k < 2; ++k)
        if (!p->upload_pinned[k]) SBA_HIP_TRY(hipHostMalloc(&p->upload_pinned[k], chunk_bytes, hipHostMallocDefault));
      sba::DeviceBuffer stage_buf(&p->poisoned);
      SBA_HIP_TRY(stage_buf.alloc(2 * chunk_bytes));
      char* dev_stage[2] = {stage_buf.as<char>(), stage_buf.as<char>() + chunk_bytes};
      int threads = 6;
      if (const char* env = std::getenv("SBA_UPLOAD_THREADS")) { const int v = std::atoi(env); if (v >= 1 && v <= 64) threads = v; }
      threads = std::max(1, std::min<int>(threads, static_cast<int>(std::thread::hardware_concurrency())));
      sba::CopyPool pool(threads);
      hipEvent_t h2d_done[2] = {p->ev0, p->ev1};
      struct Job { const double* src; int width; int which; };
      const Job jobs[3] = {{static_cast<const double*>(left), 3, 0}, {static_cast<const double*>(right), 3, 1},
                           {static_cast<const double*>(d12), 2, 2}};
      size_t k = 0;
      for (const Job& job : jobs) {
        if (!job.src) continue;
        for (size_t first = 0; first < n; first += chunk, ++k) {
          const size_t m = std::min(chunk, n - first), bytes = m * job.width * sizeof(double);
          const int b = static_cast<int>(k & 1);
          if (k >= 2) {                                        // the DMA out of pinned buffer b has finished
            const int rc2 = sba::event_wait(h2d_done[b], "upload staging", &p->poisoned);
            if (rc2) return rc2;
          }
          pool.copy(p->upload_pinned[b], job.src + job.width * first, bytes);
          SBA_HIP_TRY(hipMemcpyAsync(dev_stage[b], p->upload_pinned[b], bytes, hipMemcpyHostToDevice, p->stream));
          SBA_HIP_TRY(hipEventRecord(h2d_done[b], p->stream));
          // stream order: this kernel reads dev_stage[b] after the copy above, and the copy of chunk k + 2 into the same
          // device buffer waits for it
          double* stage = reinterpret_cast<double*>(dev_stage[b]);
          if (job.which < 2)
            SBA_HIP_TRY(sba::launch_aos_to_planes(stage, m, first, p->coord[3 * job.which + 0], p->coord[3 * job.which + 1],
                                                  p->coord[3 * job.which + 2], store, p->stream));
          else
            SBA_HIP_TRY(sba::launch_d12_to_planes(stage, m, first, p->dplane[0], p->dplane[1], p->stream));
        }
      }
      SBA_SYNC(p, "upload");      // before the staging buffer goes out of scope
    }
  }
  SBA_SYNC(p, "stream synchronisation");
  p->uploaded = true;
  return SBA_OK;
}

int sba_problem_upload(sba_problem* p, const double* left_xyz, const double* right_xyz,
                       const double* d12, size_t n, int store) {
  return upload_common(p, left_xyz, right_xyz, d12, n, store, false);
}

int sba_problem_upload_keypoints(sba_problem* p, const void* left_keypoints, const void* right_keypoints, size_t n,
                                 size_t stride_bytes, int im_width, int im_height, const double* d12, int store) {
  if (!p) return fail(SBA_ERR_INVALID_ARG, "null problem handle");
  SBA_REFUSE_POISONED(p);
  if (store != SBA_STORE_F64 && store != SBA_STORE_F32) return fail(SBA_ERR_INVALID_ARG, "bad store %d", store);
  if (n > 0 && (!left_keypoints || !right_keypoints)) return fail(SBA_ERR_INVALID_ARG, "null key-point array");
  if (stride_bytes < 8 || stride_bytes % 4 != 0) return fail(SBA_ERR_INVALID_ARG, "stride_bytes must be a multiple of 4 and >= 8");
  if (im_width <= 0 || im_height <= 0) return fail(SBA_ERR_INVALID_ARG, "bad image size");
  SBA_HIP_TRY(hipSetDevice(p->device));
  int rc = alloc_planes(p, n, d12 != nullptr, store);
  if (rc) return rc;
  if (n > 0) {
    sba::DeviceBuffer kl(&p->poisoned), kr(&p->poisoned);
    SBA_HIP_TRY(kl.alloc(n * stride_bytes));
    SBA_HIP_TRY(kr.alloc(n * stride_bytes));
    SBA_HIP_TRY(hipMemcpyAsync(kl.ptr, left_keypoints, n * stride_bytes, hipMemcpyHostToDevice, p->stream));
    SBA_HIP_TRY(hipMemcpyAsync(kr.ptr, right_keypoints, n * stride_bytes, hipMemcpyHostToDevice, p->stream));
    SBA_HIP_TRY(sba::launch_keypoints_to_planes(kl.as<uint8_t>(), kr.as<uint8_t>(), n, stride_bytes, im_width, im_height,
                                                p->coord, store, p->stream));
    SBA_SYNC(p, "stream synchronisation");
  }
  p->uploaded = true;
  if (d12) return sba_problem_set_depths(p, d12);
  return SBA_OK;
}

int sba_problem_upload_device(sba_problem* p, const void* left_xyz_dev, const void* right_xyz_dev,
                              const void* d12_dev, size_t n, int store) {
  return upload_common(p, left_xyz_dev, right_xyz_dev, d12_dev, n, store, true);
}

int sba_problem_set_depths(sba_problem* p, const double* d12) {
  if (!p || !d12) return fail(SBA_ERR_INVALID_ARG, "null argument");
  SBA_REFUSE_POISONED(p);
  if (!p->uploaded) return fail(SBA_ERR_NOT_UPLOADED, "no correspondences uploaded");
  SBA_HIP_TRY(hipSetDevice(p->device));
  if (!p->has_d12) {
    const size_t pad = 8 * p->plane_stagger, bytes = std::max<size_t>(p->plane_elems, 1) * 8 + pad;
    for (int k = 0; k < 2; ++k) {
      const int rc = ensure_plane(p, 6 + k, bytes);
      if (rc) return rc;
      p->dplane[k] = reinterpret_cast<double*>(static_cast<char*>(p->plane_base[6 + k]) + (6 + k) * p->plane_stagger);
    }
    p->has_d12 = true;
  }
  if (p->n > 0) {
    sba::DeviceBuffer stage(&p->poisoned);
    SBA_HIP_TRY(stage.alloc(p->n * 2 * sizeof(double)));
    SBA_HIP_TRY(hipMemcpyAsync(stage.ptr, d12, p->n * 2 * sizeof(double), hipMemcpyHostToDevice, p->stream));
    SBA_HIP_TRY(sba::launch_d12_to_planes(stage.as<double>(), p->n, 0, p->dplane[0], p->dplane[1], p->stream));
    SBA_SYNC(p, "stream synchronisation");
  }
  return SBA_OK;
}

int sba_problem_set_kernel(sba_problem* p, int kind) {
  if (!p) return fail(SBA_ERR_INVALID_ARG, "null problem handle");
  if (kind != SBA_KERNEL_FACTORED && kind != SBA_KERNEL_EXPLICIT) return fail(SBA_ERR_INVALID_ARG, "bad kernel kind %d", kind);
  p->kind = kind;
  return SBA_OK;
}

int sba_problem_size(const sba_problem* p, size_t* n) {
  if (!p || !n) return fail(SBA_ERR_INVALID_ARG, "null argument");
  *n = p->n;
  return SBA_OK;
}

int sba_problem_eval_pack(sba_problem* p, int mode, int depth_mode, const double rot[3],
                          const double tran[3], double d1, double d2, double huber_delta,
                          double pack[SBA_PACK_SIZE]) {
  int rc = check_args(p, mode, depth_mode, rot, tran);
  if (rc) return rc;
  if (!pack) return fail(SBA_ERR_INVALID_ARG, "pack is null");
  SBA_HIP_TRY(hipSetDevice(p->device));
  sba::SweepParams prm;
  make_params(p, depth_mode, rot, tran, d1, d2, huber_delta, &prm);
  make_frame(p, mode, rot);
  rc = enqueue_sweep(p, mode, depth_mode, prm);
  if (rc) return rc;
  return fetch_pack(p, pack);
}

int sba_problem_eval_steps(sba_problem* p, int mode, int depth_mode, const double rot[3],
                           const double tran[3], double d1, double d2, double huber_delta, int steps,
                           double pack[SBA_PACK_SIZE], double* seconds) {
  int rc = check_args(p, mode, depth_mode, rot, tran);
  if (rc) return rc;
  if (!pack || steps < 1) return fail(SBA_ERR_INVALID_ARG, "bad pack/steps");
  SBA_HIP_TRY(hipSetDevice(p->device));
  const auto t0 = std::chrono::steady_clock::now();
  for (int i = 0; i < steps; ++i) {
    sba::SweepParams prm;
    make_params(p, depth_mode, rot, tran, d1, d2, huber_delta, &prm);   // the host-side R, dR/dw of every iteration
    make_frame(p, mode, rot);
    rc = enqueue_sweep(p, mode, depth_mode, prm);
    if (rc) return rc;
    rc = fetch_pack(p, pack);
    if (rc) return rc;
  }
  if (seconds) *seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  return SBA_OK;
}

int sba_expand_pack(int mode, const double pack[SBA_PACK_SIZE], sba_normal_eq* out) {
  if (!pack || !out) return fail(SBA_ERR_INVALID_ARG, "null argument");
  if (mode < SBA_MODE_ROT || mode > SBA_MODE_RT) return fail(SBA_ERR_INVALID_ARG, "bad mode %d", mode);
  sba::expand_pack(mode, pack, out);
  return SBA_OK;
}

int sba_problem_eval(sba_problem* p, int mode, int depth_mode, const double rot[3],
                     const double tran[3], double d1, double d2, double huber_delta,
                     sba_normal_eq* out) {
  if (!out) return fail(SBA_ERR_INVALID_ARG, "out is null");
  double pack[SBA_PACK_SIZE];
  const int rc = sba_problem_eval_pack(p, mode, depth_mode, rot, tran, d1, d2, huber_delta, pack);
  if (rc) return rc;
  sba::expand_pack(mode, pack, out);
  return SBA_OK;
}

int sba_problem_eval_timed(sba_problem* p, int mode, int depth_mode, const double rot[3],
                           const double tran[3], double d1, double d2, double huber_delta,
                           int repeat, double pack[SBA_PACK_SIZE], double* mean_step_ms,
                           double* mean_sweep_ms) {
  int rc = check_args(p, mode, depth_mode, rot, tran);
  if (rc) return rc;
  if (!pack || repeat < 1 || repeat > 100000) return fail(SBA_ERR_INVALID_ARG, "bad pack/repeat");
  SBA_HIP_TRY(hipSetDevice(p->device));
  sba::SweepParams prm;
  make_params(p, depth_mode, rot, tran, d1, d2, huber_delta, &prm);
  make_frame(p, mode, rot);
  float ms = 0.f;
  if (mean_sweep_ms) {
    // the sweep kernel alone: `repeat` launches back to back under ONE event pair (kernel + the ~1.5 us boundary
    // between dependent launches; per-launch event brackets would add another 2-4 us each)
    int grid = 0;
    rc = grid_for(p, mode, depth_mode, prm.delta > 0.0, &grid);
    if (rc) return rc;
    sba::Planes pl;
    for (int k = 0; k < 3; ++k) { pl.x1[k] = p->coord[k]; pl.x2[k] = p->coord[3 + k]; }
    pl.d1 = p->dplane[0]; pl.d2 = p->dplane[1];
    sba::SweepOut out;
    out.partials = p->partials; out.pack_dev = p->pack_dev; out.pack_host = nullptr; out.ticket = nullptr; out.seq = 0;
    SBA_HIP_TRY(hipEventRecord(p->ev0, p->stream));
    for (int i = 0; i < repeat; ++i)
      SBA_HIP_TRY(sba::launch_sweep(mode, depth_mode, p->store, p->kind, pl, prm, out, grid, p->stream));
    SBA_HIP_TRY(hipEventRecord(p->ev1, p->stream));
    SBA_HIP_TRY(hipEventSynchronize(p->ev1));
    SBA_HIP_TRY(hipEventElapsedTime(&ms, p->ev0, p->ev1));
    *mean_sweep_ms = static_cast<double>(ms) / repeat;
  }
  // complete steps (sweep + final reduction [+ all-reduce] + publication), still without host synchronisation between them
  SBA_HIP_TRY(hipEventRecord(p->ev0, p->stream));
  for (int i = 0; i < repeat; ++i) {
    rc = enqueue_sweep(p, mode, depth_mode, prm);
    if (rc) return rc;
  }
  SBA_HIP_TRY(hipEventRecord(p->ev1, p->stream));
  rc = fetch_pack(p, pack);
  if (rc) return rc;
  SBA_HIP_TRY(hipEventSynchronize(p->ev1));
  SBA_HIP_TRY(hipEventElapsedTime(&ms, p->ev0, p->ev1));
  if (mean_step_ms) *mean_step_ms = static_cast<double>(ms) / repeat;
  return SBA_OK;
}

int sba_problem_eval_launch_times(sba_problem* p, int mode, int depth_mode, const double rot[3],
                                  const double tran[3], double d1, double d2, double huber_delta,
                                  int repeat, float* launch_ms) {
  int rc = check_args(p, mode, depth_mode, rot, tran);
  if (rc) return rc;
  if (!launch_ms || repeat < 1 || repeat > 4096) return fail(SBA_ERR_INVALID_ARG, "bad launch_ms/repeat (1..4096)");
  SBA_HIP_TRY(hipSetDevice(p->device));
  sba::SweepParams prm;
  make_params(p, depth_mode, rot, tran, d1, d2, huber_delta, &prm);
  make_frame(p, mode, rot);
  int grid = 0;
  rc = grid_for(p, mode, depth_mode, prm.delta > 0.0, &grid);
  if (rc) return rc;
  sba::Planes pl;
  for (int k = 0; k < 3; ++k) { pl.x1[k] = p->coord[k]; pl.x2[k] = p->coord[3 + k]; }
  pl.d1 = p->dplane[0]; pl.d2 = p->dplane[1];
  sba::SweepOut out;
  out.partials = p->partials; out.pack_dev = p->pack_dev; out.pack_host = nullptr; out.ticket = nullptr; out.seq = 0;
  // one event between every two launches: launch i runs between events i and i + 1
  struct Events {
    std::vector<hipEvent_t> ev;
    ~Events() { for (hipEvent_t e : ev) if (e) (void)hipEventDestroy(e); }
  } events;
  events.ev.assign(static_cast<size_t>(repeat) + 1, nullptr);
  for (hipEvent_t& e : events.ev) SBA_HIP_TRY(hipEventCreate(&e));
  SBA_HIP_TRY(hipEventRecord(events.ev[0], p->stream));
  for (int i = 0; i < repeat; ++i) {
    SBA_HIP_TRY(sba::launch_sweep(mode, depth_mode, p->store, p->kind, pl, prm, out, grid, p->stream));
    SBA_HIP_TRY(hipEventRecord(events.ev[static_cast<size_t>(i) + 1], p->stream));
  }
  SBA_HIP_TRY(hipEventSynchronize(events.ev[static_cast<size_t>(repeat)]));
  for (int i = 0; i < repeat; ++i)
    SBA_HIP_TRY(hipEventElapsedTime(&launch_ms[i], events.ev[static_cast<size_t>(i)], events.ev[static_cast<size_t>(i) + 1]));
  return SBA_OK;
}

int sba_problem_solve(sba_problem* p, int mode, int depth_mode, double rot[3], double tran[3],
                      double d1, double d2, const sba_lm_options* opt, sba_lm_summary* summary) {
  int rc = check_args(p, mode, depth_mode, rot, tran);
  if (rc) return rc;
  sba_lm_options o;
  if (opt) o = *opt; else sba::lm_default_options(&o);
  double eval_seconds = 0.0;
  int eval_rc = SBA_OK;
  // Small, unsharded problems (the reference's real sizes): ONE resident single-block kernel serves every sweep of this
  // stage -- the host LM below is unchanged, only the evaluator talks to a kernel that is already running instead of
  // launching two per iteration (sba_resident.hpp).  SBA_RESIDENT_MAX_N=0 keeps the launch-per-sweep path.
  // SBA_SMALL_ONE_LAUNCH=2 (not the default: measured slower, sba_problem.hpp) and nobody watching the iterations: the whole
  // stage as ONE launch -- the problem as a batch of one pair through batch_lm_kernel, the solver on the device.
  if (sba::shim::resident_eligible(p, false) && !o.verbose && sba::shim::small_one_launch(false)) {
    SBA_HIP_TRY(hipSetDevice(p->device));
    const auto t0 = std::chrono::steady_clock::now();
    sba::shim::SmallRecord* rec = static_cast<sba::shim::SmallRecord*>(p->small_rec);
    sba::shim::SmallRecord* rec_dev = static_cast<sba::shim::SmallRecord*>(p->small_rec_dev);
    rec->desc = sba::PairDesc{0ull, p->n, sba::kPairTile, 0ull};
    for (int a = 0; a < 3; ++a) { rec->io.rot[a] = rot[a]; rec->io.tran[a] = tran[a]; }
    rec->io.d1 = d1; rec->io.d2 = d2;
    rec->io.summary = sba_lm_summary{};
    rec->io.status = SBA_ERR_NUMERIC; rec->io.pad_ = 0;
    sba::Planes pl;
    for (int k = 0; k < 3; ++k) { pl.x1[k] = p->coord[k]; pl.x2[k] = p->coord[3 + k]; }
    pl.d1 = p->dplane[0]; pl.d2 = p->dplane[1];
    const unsigned long long seq = ++p->small_seq;
    SBA_HIP_TRY(sba::launch_batch_lm(mode, depth_mode, p->store, p->kind, pl, &rec_dev->desc, &rec_dev->io, o, 1, p->ticket,
                                     const_cast<unsigned long long*>(&rec_dev->seq), seq, p->stream));
    rc = sba::wait_for_sequence(&rec->seq, seq, p->stream, "one-launch solve stage", &p->poisoned);
    if (rc) return rc;
    for (int a = 0; a < 3; ++a) { rot[a] = rec->io.rot[a]; tran[a] = rec->io.tran[a]; }
    sba_lm_summary local;
    sba_lm_summary* s = summary ? summary : &local;
    *s = rec->io.summary;
    s->seconds_total = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    s->seconds_eval = s->seconds_total;
    if (rec->io.status != SBA_OK) return fail(rec->io.status, "LM failed: non-finite or singular normal equations");
    return SBA_OK;
  }
  sba::shim::ResidentSession session(p);
  if (sba::shim::resident_eligible(p, false)) {
    SBA_HIP_TRY(hipSetDevice(p->device));
    rc = session.start_sweep(mode, depth_mode, o.huber_delta > 0.0);
    if (rc) return rc;
  }
  auto evaluator = [&](const double r[3], const double t[3], sba_normal_eq* ne) -> bool {
    const auto t0 = std::chrono::steady_clock::now();
    if (session.active()) {
      sba::SweepParams prm;
      make_params(p, depth_mode, r, t, d1, d2, o.huber_delta, &prm);
      make_frame(p, mode, r);
      double payload[44], raw[SBA_PACK_SIZE], pack[SBA_PACK_SIZE];
      payload[0] = static_cast<double>(sba::RESIDENT_OP_SWEEP);
      std::memcpy(payload + 1, &prm, 42 * sizeof(double));
      const unsigned long long n_bits = prm.n;
      std::memcpy(payload + 43, &n_bits, sizeof(double));
      eval_rc = session.call(payload, 44, raw, SBA_PACK_SIZE);
      if (eval_rc == SBA_OK) {
        if (p->kind == SBA_KERNEL_FACTORED && mode != SBA_MODE_TRAN)
          sba::moments_to_normal_pack(true, mode == SBA_MODE_RT, p->frame_B, p->frame_J, raw, pack);
        else
          std::memcpy(pack, raw, sizeof(pack));
        sba::expand_pack(mode, pack, ne);
      }
    } else {
      eval_rc = sba_problem_eval(p, mode, depth_mode, r, t, d1, d2, o.huber_delta, ne);
    }
    eval_seconds += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    return eval_rc == SBA_OK;
  };
  sba_lm_summary local;
  sba_lm_summary* s = summary ? summary : &local;
  const int lm_rc = sba::lm_solve(mode, rot, tran, o, evaluator, s);
  s->seconds_eval = eval_seconds;
  const int end_rc = session.end();
  if (eval_rc != SBA_OK) return eval_rc;  // message already set by the failing eval
  if (lm_rc != SBA_OK) return fail(lm_rc, "LM failed: non-finite or singular normal equations");
  return end_rc;
}

static int require_device(int device) {
  int count = 0;
  const hipError_t e = hipGetDeviceCount(&count);
  if (e != hipSuccess || count <= 0)
    return fail(SBA_ERR_NO_DEVICE, "no HIP device available (%s); this library has no CPU path",
                e != hipSuccess ? hipGetErrorString(e) : "device count is 0");
  if (device < 0 || device >= count)
    return fail(SBA_ERR_INVALID_ARG, "device %d out of range [0,%d)", device, count);
  SBA_HIP_TRY(hipSetDevice(device));
  return SBA_OK;
}

int sba_keypoints_to_sphere(int device, const void* keypoints, size_t n, size_t stride_bytes,
                            int im_width, int im_height, double* out_xyz) {
  if (n > 0 && (!keypoints || !out_xyz)) return fail(SBA_ERR_INVALID_ARG, "null array");
  if (stride_bytes < 8 || stride_bytes % 4 != 0)
    return fail(SBA_ERR_INVALID_ARG, "stride_bytes must be a multiple of 4 and >= 8");
  if (im_width <= 0 || im_height <= 0) return fail(SBA_ERR_INVALID_ARG, "bad image size");
  int rc = require_device(device);
  if (rc) return rc;
  if (n == 0) return SBA_OK;
  sba::DeviceBuffer kp_dev, out_dev;
  SBA_HIP_TRY(kp_dev.alloc(n * stride_bytes));
  SBA_HIP_TRY(out_dev.alloc(n * 3 * sizeof(double)));
  SBA_HIP_TRY(hipMemcpy(kp_dev.ptr, keypoints, n * stride_bytes, hipMemcpyHostToDevice));
  SBA_HIP_TRY(sba::launch_keypoints_to_sphere(kp_dev.as<uint8_t>(), n, stride_bytes, im_width, im_height,
                                              out_dev.as<double>(), nullptr));
  SBA_HIP_TRY(hipMemcpy(out_xyz, out_dev.ptr, n * 3 * sizeof(double), hipMemcpyDeviceToHost));
  return SBA_OK;
}

}  // extern "C"
