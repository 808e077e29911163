// Multi-GPU transports behind the C-ABI (include/sba_hip.h): RCCL bound at run time, the direct peer exchange over HIP
// IPC / xGMI, the user all-reduce hook -- and the SUM all-reduce of the result pack over whichever is attached.
#include <dlfcn.h>

#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <vector>

#include "sba_problem.hpp"

// RCCL is bound at run time (dlopen), so nothing here links against it -- but where its header is installed the
// hand-declared ABI (enum values, the 128-byte id passed by value, argument lists) is checked against it at compile time.
#if __has_include(<rccl/rccl.h>)
#include <rccl/rccl.h>
#include <type_traits>
static_assert(sba::shim::kNcclFloat64 == static_cast<int>(ncclFloat64) && sba::shim::kNcclSum == static_cast<int>(ncclSum),
              "RCCL enum values changed");
static_assert(sizeof(ncclUniqueId) == SBA_COMM_ID_BYTES && sizeof(sba::shim::Rccl::UniqueId) == sizeof(ncclUniqueId),
              "ncclUniqueId is no longer 128 bytes");
static_assert(std::is_pointer<ncclComm_t>::value && sizeof(ncclResult_t) == sizeof(int) && sizeof(ncclDataType_t) == sizeof(int) &&
              sizeof(ncclRedOp_t) == sizeof(int), "RCCL handle / enum representation changed");
static_assert(std::is_same<decltype(&ncclAllReduce), ncclResult_t (*)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t,
                                                                      ncclComm_t, hipStream_t)>::value,
              "ncclAllReduce argument list changed");
static_assert(std::is_same<decltype(&ncclCommInitRank), ncclResult_t (*)(ncclComm_t*, int, ncclUniqueId, int)>::value,
              "ncclCommInitRank argument list changed");
#endif

namespace sba {
namespace shim {

Rccl& rccl() {
  static Rccl r;
  static bool tried = false;
  if (tried) return r;
  tried = true;
  const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
  for (const char* nm : names) {
    r.handle = dlopen(nm, RTLD_NOW | RTLD_NOLOAD);
    if (r.handle) break;
  }
  if (!r.handle)
    for (const char* nm : names) {
      r.handle = dlopen(nm, RTLD_NOW | RTLD_LOCAL);
      if (r.handle) break;
    }
  if (!r.handle) {
    r.why = std::string("cannot load librccl: ") + (dlerror() ? dlerror() : "?");
    return r;
  }
  r.GetUniqueId = reinterpret_cast<decltype(r.GetUniqueId)>(dlsym(r.handle, "ncclGetUniqueId"));
  r.CommInitRank = reinterpret_cast<decltype(r.CommInitRank)>(dlsym(r.handle, "ncclCommInitRank"));
  r.CommDestroy = reinterpret_cast<decltype(r.CommDestroy)>(dlsym(r.handle, "ncclCommDestroy"));
  r.AllReduce = reinterpret_cast<decltype(r.AllReduce)>(dlsym(r.handle, "ncclAllReduce"));
  r.GetErrorString =
      reinterpret_cast<decltype(r.GetErrorString)>(dlsym(r.handle, "ncclGetErrorString"));
  r.ok = r.GetUniqueId && r.CommInitRank && r.CommDestroy && r.AllReduce;
  if (!r.ok) r.why = "librccl is missing ncclGetUniqueId/CommInitRank/CommDestroy/AllReduce";
  return r;
}

// SUM all-reduce of the 24-double pack in p->pack_dev over the attached transport (RCCL or the user hook), then the
// one-wave publish kernel hands the result to the host like the single-GPU path does itself.
int allreduce_pack(sba_problem* p) {
  if (p->peer_ready) {   // direct peer stores + rank-ordered local sum; the exchanging wave publishes to the host
    p->published = p->publish;
    if (p->publish) ++p->seq;
    SBA_TRY_HIP(sba::launch_peer_exchange(p->pack_dev, p->peers, ++p->xseq, p->pack_dev,
                                          p->publish ? p->pack_host_dev : nullptr, p->seq, p->stream));
    return SBA_OK;
  }
  if (p->comm) {
    Rccl& r = rccl();
    const int rc = r.AllReduce(p->pack_dev, p->pack_dev, SBA_PACK_SIZE, kNcclFloat64, kNcclSum,
                               p->comm, p->stream);
    if (rc != 0)
      return sba::set_error(SBA_ERR_COMM, "ncclAllReduce failed: %s",
                  r.GetErrorString ? r.GetErrorString(rc) : "?");
  } else if (p->hook) {
    const int rc = p->hook(p->pack_dev, SBA_PACK_SIZE, p->stream, p->hook_user);
    if (rc != 0) return sba::set_error(SBA_ERR_COMM, "all-reduce hook returned %d", rc);
  }
  p->published = false;
  if (p->publish) {
    SBA_TRY_HIP(sba::launch_publish(p->pack_dev, p->pack_host_dev, ++p->seq, p->stream));
    p->published = true;
  }
  return SBA_OK;
}

// SUM all-reduce of `count` doubles at `dev` (count a multiple of 24) over the attached transport, in stream order.
// Over the peer transport the buffer travels as count / 24 back-to-back exchanges of the 24-double inbox slots; the
// last one publishes to the host, and since every exchange reports the STICKY device-side timeout word, a timeout in any
// of them shows in word 25 of the host pack (checked by the caller).
int allreduce_buffer(sba_problem* p, double* dev, size_t count) {
  if (p->peer_ready) {
    if (count % SBA_PACK_SIZE != 0) return sba::set_error(SBA_ERR_INVALID_ARG, "peer all-reduce needs a multiple of 24 doubles");
    for (size_t off = 0; off < count; off += SBA_PACK_SIZE) {
      const bool last = off + SBA_PACK_SIZE >= count;
      SBA_TRY_HIP(sba::launch_peer_exchange(dev + off, p->peers, ++p->xseq, dev + off,
                                            last ? p->pack_host_dev : nullptr, last ? ++p->seq : 0, p->stream));
    }
    return SBA_OK;
  }
  if (p->comm) {
    Rccl& r = rccl();
    const int rc = r.AllReduce(dev, dev, count, kNcclFloat64, kNcclSum, p->comm, p->stream);
    if (rc != 0) return sba::set_error(SBA_ERR_COMM, "ncclAllReduce failed: %s", r.GetErrorString ? r.GetErrorString(rc) : "?");
  } else if (p->hook) {
    const int rc = p->hook(dev, count, p->stream, p->hook_user);
    if (rc != 0) return sba::set_error(SBA_ERR_COMM, "all-reduce hook returned %d", rc);
  }
  return SBA_OK;
}

}  // namespace shim
}  // namespace sba

using sba::shim::Rccl;
using sba::shim::rccl;
using sba::shim::kNcclFloat64;
using sba::shim::kNcclSum;
using sba::shim::fetch_pack_raw;

extern "C" {

// ---- multi-GPU -----------------------------------------------------------------------------------
int sba_comm_unique_id(char id[SBA_COMM_ID_BYTES]) {
  if (!id) return sba::set_error(SBA_ERR_INVALID_ARG, "id is null");
  Rccl& r = rccl();
  if (!r.ok) return sba::set_error(SBA_ERR_COMM, "%s", r.why.c_str());
  Rccl::UniqueId u;
  const int rc = r.GetUniqueId(&u);
  if (rc != 0) return sba::set_error(SBA_ERR_COMM, "ncclGetUniqueId failed (%d)", rc);
  std::memcpy(id, u.internal, SBA_COMM_ID_BYTES);
  return SBA_OK;
}

int sba_problem_comm_init_rank(sba_problem* p, int nranks, int rank, const char id[SBA_COMM_ID_BYTES]) {
  if (!p || !id) return sba::set_error(SBA_ERR_INVALID_ARG, "null argument");
  SBA_REFUSE_POISONED(p);
  if (nranks < 1 || rank < 0 || rank >= nranks) return sba::set_error(SBA_ERR_INVALID_ARG, "bad rank %d/%d", rank, nranks);
  Rccl& r = rccl();
  if (!r.ok) return sba::set_error(SBA_ERR_COMM, "%s", r.why.c_str());
  SBA_TRY_HIP(hipSetDevice(p->device));
  Rccl::UniqueId u;
  std::memcpy(u.internal, id, SBA_COMM_ID_BYTES);
  void* comm = nullptr;
  const int rc = r.CommInitRank(&comm, nranks, u, rank);
  if (rc != 0)
    return sba::set_error(SBA_ERR_COMM, "ncclCommInitRank failed: %s", r.GetErrorString ? r.GetErrorString(rc) : "?");
  if (p->comm) r.CommDestroy(p->comm);
  p->comm = comm;
  p->nranks = nranks;
  p->shard_rank = rank;
  p->shard_count = nranks;
  return SBA_OK;
}

int sba_rccl_available(void) {
  Rccl& r = rccl();
  if (!r.ok) { (void)sba::set_error(SBA_ERR_COMM, "%s", r.why.c_str()); return 0; }
  return 1;
}

int sba_problem_comm_destroy(sba_problem* p) {
  if (!p) return SBA_OK;
  if (p->comm) {
    (void)hipSetDevice(p->device);
    if (!p->poisoned && p->stream) (void)sba::stream_wait(p->stream, "communicator tear-down", &p->poisoned);
    SBA_REFUSE_POISONED(p);     // ncclCommDestroy waits for the communicator's stream: leak it
    Rccl& r = rccl();
    if (r.ok) r.CommDestroy(p->comm);
    p->comm = nullptr;
  }
  return SBA_OK;
}

// ---- direct peer exchange ---------------------------------------------------------------------------------------
namespace {
// Inboxes exported by THIS process.  A process may host several ranks (one host thread and one sba_problem per GPU of a
// node -- or, in tests, several ranks on one GPU): hipIpcOpenMemHandle refuses a handle of the opening process itself, so
// peers that live in the same process are connected by plain device pointer (with peer access enabled when they sit on
// different devices); everything else about the exchange is unchanged.
struct LocalInbox { hipIpcMemHandle_t handle; double* ptr; int device; };
std::mutex g_local_mutex;
std::vector<LocalInbox> g_local_inboxes;

void register_local_inbox(const hipIpcMemHandle_t& h, double* ptr, int device) {
  std::lock_guard<std::mutex> lock(g_local_mutex);
  g_local_inboxes.push_back(LocalInbox{h, ptr, device});
}
void unregister_local_inbox(const double* ptr) {
  std::lock_guard<std::mutex> lock(g_local_mutex);
  g_local_inboxes.erase(std::remove_if(g_local_inboxes.begin(), g_local_inboxes.end(),
                                       [ptr](const LocalInbox& b) { return b.ptr == ptr; }), g_local_inboxes.end());
}
bool find_local_inbox(const hipIpcMemHandle_t& h, LocalInbox* out) {
  std::lock_guard<std::mutex> lock(g_local_mutex);
  for (const LocalInbox& b : g_local_inboxes)
    if (std::memcmp(&b.handle, &h, sizeof(h)) == 0) { *out = b; return true; }
  return false;
}
}  // namespace

int sba_problem_peer_export(sba_problem* p, int nranks, int rank, char handle[SBA_PEER_HANDLE_BYTES]) {
  if (!p || !handle) return sba::set_error(SBA_ERR_INVALID_ARG, "null argument");
  SBA_REFUSE_POISONED(p);
  if (nranks < 1 || nranks > sba::kMaxPeers || rank < 0 || rank >= nranks)
    return sba::set_error(SBA_ERR_INVALID_ARG, "bad rank %d/%d (at most %d ranks)", rank, nranks, sba::kMaxPeers);
  static_assert(sizeof(hipIpcMemHandle_t) == SBA_PEER_HANDLE_BYTES, "IPC handle size");
  SBA_TRY_HIP(hipSetDevice(p->device));
  (void)sba_problem_peer_disable(p);
  const size_t bytes = sba::kInboxDoubles * sizeof(double);
  // fine-grained / uncached device memory: remote stores and local polls must not sit in a non-coherent cache
  void* mem = nullptr;
  if (hipExtMallocWithFlags(&mem, bytes, hipDeviceMallocUncached) != hipSuccess) {
    (void)hipGetLastError();
    if (hipExtMallocWithFlags(&mem, bytes, hipDeviceMallocFinegrained) != hipSuccess) {
      (void)hipGetLastError();
      SBA_TRY_HIP(hipMalloc(&mem, bytes));
    }
  }
  p->inbox = static_cast<double*>(mem);
  SBA_TRY_HIP(hipMemset(p->inbox, 0, bytes));
  if (!p->peer_sticky) SBA_TRY_HIP(hipMalloc(reinterpret_cast<void**>(&p->peer_sticky), 64));
  SBA_TRY_HIP(hipMemset(p->peer_sticky, 0, 64));
  if (const char* env = std::getenv("SBA_PEER_TIMEOUT_S")) { const double v = std::atof(env); if (v > 0.0) p->peer_timeout_s = v; }
  p->peers.timeout_ticks = static_cast<unsigned long long>(p->peer_timeout_s * 1e3 * p->wall_clock_khz);
  p->peers.sticky = p->peer_sticky;
  SBA_TRY_HIP(hipDeviceSynchronize());
  hipIpcMemHandle_t h;
  const hipError_t e = hipIpcGetMemHandle(&h, p->inbox);
  if (e != hipSuccess) {
    (void)hipFree(p->inbox);
    p->inbox = nullptr;
    return sba::set_error(SBA_ERR_COMM, "hipIpcGetMemHandle failed: %s", hipGetErrorString(e));
  }
  std::memcpy(handle, &h, SBA_PEER_HANDLE_BYTES);
  register_local_inbox(h, p->inbox, p->device);
  p->peers.nranks = nranks;
  p->peers.rank = rank;
  p->shard_rank = rank;
  p->shard_count = nranks;
  return SBA_OK;
}

int sba_problem_peer_connect(sba_problem* p, const char* handles) {
  if (!p || !handles) return sba::set_error(SBA_ERR_INVALID_ARG, "null argument");
  SBA_REFUSE_POISONED(p);
  if (!p->inbox) return sba::set_error(SBA_ERR_INVALID_ARG, "call sba_problem_peer_export first");
  SBA_TRY_HIP(hipSetDevice(p->device));
  for (int r = 0; r < p->peers.nranks; ++r) {
    if (r == p->peers.rank) { p->peers.inbox[r] = p->inbox; continue; }
    hipIpcMemHandle_t h;
    std::memcpy(&h, handles + static_cast<size_t>(r) * SBA_PEER_HANDLE_BYTES, SBA_PEER_HANDLE_BYTES);
    LocalInbox local;
    if (find_local_inbox(h, &local)) {      // rank r lives in this process: its inbox by pointer, nothing to open
      if (local.device != p->device) {
        const hipError_t pe = hipDeviceEnablePeerAccess(local.device, 0);
        if (pe != hipSuccess && pe != hipErrorPeerAccessAlreadyEnabled) {
          (void)sba_problem_peer_disable(p);
          return sba::set_error(SBA_ERR_COMM, "hipDeviceEnablePeerAccess(device %d, for rank %d) failed: %s", local.device, r,
                                hipGetErrorString(pe));
        }
        (void)hipGetLastError();
      }
      p->peers.inbox[r] = local.ptr;
      continue;
    }
    void* ptr = nullptr;
    const hipError_t e = hipIpcOpenMemHandle(&ptr, h, hipIpcMemLazyEnablePeerAccess);
    if (e != hipSuccess) {
      (void)sba_problem_peer_disable(p);
      return sba::set_error(SBA_ERR_COMM, "hipIpcOpenMemHandle(rank %d) failed: %s", r, hipGetErrorString(e));
    }
    p->peer_opened[r] = ptr;
    p->peers.inbox[r] = static_cast<double*>(ptr);
  }
  p->xseq = 0;
  p->peer_ready = true;
  return SBA_OK;
}

int sba_problem_peer_disable(sba_problem* p) {
  if (!p) return SBA_OK;
  (void)hipSetDevice(p->device);
  if (!p->poisoned && p->stream) (void)sba::stream_wait(p->stream, "peer tear-down", &p->poisoned);
  if (p->poisoned) {            // closing / freeing the inboxes would wait for the wedged device: leak them
    p->peer_ready = false;
    SBA_REFUSE_POISONED(p);
  }
  for (auto& o : p->peer_opened) {
    if (o) (void)hipIpcCloseMemHandle(o);
    o = nullptr;
  }
  if (p->inbox) { unregister_local_inbox(p->inbox); (void)hipFree(p->inbox); }
  p->inbox = nullptr;
  p->peer_ready = false;
  return SBA_OK;
}

// `rounds` exchanges of a known pack (rank + 1 in every slot, plus the round number): every rank must obtain
// nranks (nranks + 1) / 2 + nranks * round.  *ok = 1 on success.  All ranks must call it together.
int sba_problem_peer_selftest(sba_problem* p, int rounds, int* ok) {
  if (!p || !ok) return sba::set_error(SBA_ERR_INVALID_ARG, "null argument");
  *ok = 0;
  SBA_REFUSE_POISONED(p);
  if (!p->peer_ready) return sba::set_error(SBA_ERR_INVALID_ARG, "peer exchange is not connected");
  SBA_TRY_HIP(hipSetDevice(p->device));
  const int n = p->peers.nranks;
  const unsigned long long limit = p->peers.timeout_ticks;
  p->peers.timeout_ticks = static_cast<unsigned long long>(3.0 * 1e3 * p->wall_clock_khz);   // 3 s per round in the self-test
  int good = 1;
  for (int k = 0; k < rounds && good; ++k) {
    double v[32];
    for (int i = 0; i < 32; ++i) v[i] = static_cast<double>(p->peers.rank + 1 + k);
    SBA_TRY_HIP(hipMemcpyAsync(p->pack_dev, v, 24 * sizeof(double), hipMemcpyHostToDevice, p->stream));
    ++p->seq;
    p->published = true;
    SBA_TRY_HIP(sba::launch_peer_exchange(p->pack_dev, p->peers, ++p->xseq, p->pack_dev, p->pack_host_dev, p->seq,
                                          p->stream));
    double got[24];
    const int rc = fetch_pack_raw(p, got);
    if (rc != SBA_OK) { good = 0; break; }
    const double want = 0.5 * n * (n + 1) + static_cast<double>(n) * k;
    for (int i = 0; i < 24; ++i)
      if (got[i] != want) good = 0;
  }
  p->peers.timeout_ticks = limit;
  *ok = good;
  return SBA_OK;
}

int sba_problem_set_allreduce(sba_problem* p, sba_allreduce_fn fn, void* user) {
  if (!p) return sba::set_error(SBA_ERR_INVALID_ARG, "null problem handle");
  p->hook = fn;
  p->hook_user = user;
  return SBA_OK;
}

int sba_problem_set_shard(sba_problem* p, int rank, int nranks) {
  if (!p) return sba::set_error(SBA_ERR_INVALID_ARG, "null problem handle");
  if (nranks < 1 || rank < 0 || rank >= nranks) return sba::set_error(SBA_ERR_INVALID_ARG, "bad shard %d/%d", rank, nranks);
  p->shard_rank = rank;
  p->shard_count = nranks;
  return SBA_OK;
}

int sba_problem_pack_device_ptr(sba_problem* p, void** dev_ptr) {
  if (!p || !dev_ptr) return sba::set_error(SBA_ERR_INVALID_ARG, "null argument");
  *dev_ptr = p->pack_dev;
  return SBA_OK;
}

}  // extern "C"
