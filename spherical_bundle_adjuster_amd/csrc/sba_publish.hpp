// Device-side hand-over of a few result words to the host through mapped pinned memory: payload stores, a barrier that makes
// them visible to the host, then (by the caller) the sequence word the host polls.
//
// Two forms of the barrier, selected at build time (measured A/B: profiles/r03_publish_ab.md):
//   SBA_PUBLISH_WRITE_THROUGH=0  plain payload stores + system-scope release fence (buffer_wbl2 sc0 sc1: writes back every
//                                dirty line of the XCD's L2) + s_waitcnt vmcnt(0);
//   SBA_PUBLISH_WRITE_THROUGH=1  payload stores at system scope (global_store ... sc0 sc1: written through, nothing of them
//                                stays dirty in L2) + s_waitcnt vmcnt(0) only -- no L2 write-back on the critical path
//                                (cdna_hip_programming.md Guideline 16, write-through form, at system scope).
#pragma once
#include <hip/hip_runtime.h>

#ifndef SBA_PUBLISH_WRITE_THROUGH
#define SBA_PUBLISH_WRITE_THROUGH 0
#endif

namespace sba {

__device__ __forceinline__ void host_store(double* p, double v) {
#if SBA_PUBLISH_WRITE_THROUGH
  __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
#else
  *p = v;
#endif
}
__device__ __forceinline__ void host_store(unsigned long long* p, unsigned long long v) {
#if SBA_PUBLISH_WRITE_THROUGH
  __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
#else
  *p = v;
#endif
}
// every lane that stored payload calls this before the sequence word is stored
__device__ __forceinline__ void host_release() {
#if !SBA_PUBLISH_WRITE_THROUGH
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");
#endif
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

}  // namespace sba
