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
int wait_for_sequence(const volatile unsigned long long* flag, unsigned long long seq, hipStream_t stream,
                      const char* what);

// Host-side per-sweep state from (rot, tran, depths, delta): SweepParams for the device, and the frame
// (B, J) that maps the factored kernel's moments to normal equations.
void make_sweep_params(size_t n, int depth_mode, const double rot[3], const double tran[3], double d1, double d2,
                       double huber_delta, SweepParams* prm);

// Scratch device memory that is released on every exit path.
struct DeviceBuffer {
  void* ptr = nullptr;
  DeviceBuffer() = default;
  DeviceBuffer(const DeviceBuffer&) = delete;
  DeviceBuffer& operator=(const DeviceBuffer&) = delete;
  ~DeviceBuffer() { if (ptr) (void)hipFree(ptr); }
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
