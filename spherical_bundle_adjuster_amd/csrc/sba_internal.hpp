// Internal helpers shared by the host-side sources of libsba_hip.so (not exported through the C-ABI header).
#pragma once
#include <hip/hip_runtime.h>

#include "../../include/sba_hip.h"
#include "sba_device.hpp"

namespace sba {

// Records the message returned by sba_last_error() on this thread and returns `code`.
int set_error(int code, const char* fmt, ...) __attribute__((format(printf, 2, 3)));
// sba_set_host_threads / SBA_HOST_THREADS (>= 1).
int host_threads();
// Host side of the publish protocol: a kernel stores its results into mapped pinned host memory, fences at system
// scope, then stores `seq` into `*flag`; the host polls that word.  A stream query every 4096 spins turns a device
// fault into an error instead of an endless spin.  `what` names the work in the error text.
// `poisoned` (may be null) is the owning handle's poison word: it is set to 1 when the wait gives up (SBA_WAIT_TIMEOUT_S)
// or the stream reports a device error -- the work enqueued on that stream can then never be waited for again.
int wait_for_sequence(const volatile unsigned long long* flag, unsigned long long seq, hipStream_t stream,
                      const char* what, int* poisoned = nullptr);
// Bounded replacement of hipStreamSynchronize for a handle's stream: polls hipStreamQuery, gives up after
// SBA_WAIT_TIMEOUT_S like wait_for_sequence and poisons the handle then (or on a device error).
int stream_wait(hipStream_t stream, const char* what, int* poisoned);

// A handle whose device work timed out or faulted is POISONED: every later entry point on it fails with SBA_ERR_HIP, and
// its destroy must not touch the wedged stream again -- hipStreamSynchronize / hipFree / hipHostFree / hipStreamDestroy
// all wait for the device and would block for ever.  Destroy then leaks the device resources, frees the host object and
// returns SBA_ERR_HIP; the process is expected to exit (non-zero), never to re-exec.
#define SBA_REFUSE_POISONED(h)                                                                                        \
  do {                                                                                                                \
    if ((h)->poisoned)                                                                                                \
      return sba::set_error(SBA_ERR_HIP, "handle is poisoned: an earlier device wait timed out or the device faulted; " \
                                         "destroy it (its device memory is leaked) and exit the process");              \
  } while (0)

// Host-side per-sweep state from (rot, tran, depths, delta): SweepParams for the device, and the frame
// (B, J) that maps the factored kernel's moments to normal equations.
void make_sweep_params(size_t n, int depth_mode, const double rot[3], const double tran[3], double d1, double d2,
                       double huber_delta, SweepParams* prm);

// Bounded wait for a HIP event (same limit and poisoning as stream_wait).
int event_wait(hipEvent_t ev, const char* what, int* poisoned);

// A handful of host threads that copy one byte range in parallel, again and again (the staging copies of a large upload):
// the workers spin on a generation counter between copies -- a copy is a few milliseconds apart at most -- and the
// calling thread takes the first share.
class CopyPool {
 public:
  explicit CopyPool(int threads);
  ~CopyPool();
  CopyPool(const CopyPool&) = delete;
  CopyPool& operator=(const CopyPool&) = delete;
  void copy(void* dst, const void* src, size_t bytes);
  int threads() const { return nt_; }

 private:
  struct Impl;
  Impl* impl_;
  int nt_;
};

// Scratch device memory that is released on every exit path.
// `poison` (may be null): the owning handle's poison word -- hipFree waits for the device, so a buffer whose handle got
// poisoned meanwhile is leaked instead of freed.
struct DeviceBuffer {
  void* ptr = nullptr;
  const int* poison = nullptr;
  DeviceBuffer() = default;
  explicit DeviceBuffer(const int* poison_word) : poison(poison_word) {}
  DeviceBuffer(const DeviceBuffer&) = delete;
  DeviceBuffer& operator=(const DeviceBuffer&) = delete;
  ~DeviceBuffer() { if (ptr && !(poison && *poison)) (void)hipFree(ptr); }
  hipError_t alloc(size_t bytes) { return hipMalloc(&ptr, bytes > 0 ? bytes : 1); }
  template <typename T> T* as() const { return static_cast<T*>(ptr); }
};

}  // namespace sba

#define SBA_TRY_HIP(expr)                                                                            \
  do {                                                                                               \
    hipError_t _e = (expr);                                                                          \
    if (_e != hipSuccess)                                                                            \
      return sba::set_error(SBA_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, \
                            __LINE__);                                                               \
  } while (0)
