// TEST HARNESS (tests/ only): the C-ABI's handle logic against a MOCK device whose stream can be wedged.
// Links the product's host translation units (sba_shim / sba_transport / sba_stages / sba_batch .cpp, compiled with g++
// against tests/harness/fake_hip) with a fake runtime (device memory = host memory) and stub kernel launchers:
//   healthy: a launch that would publish stores its sequence number at once, the stream reads "drained";
//   wedged : launches do nothing, hipStreamQuery says hipErrorNotReady for ever, and any call that would WAIT for the
//            device on the real runtime (hipStreamSynchronize, hipFree, hipHostFree, hipStreamDestroy, hipDeviceSynchronize,
//            hipEventSynchronize, hipMemcpy) is counted as a violation -- on a real wedged GPU it never returns
//            (gpurun_out/r2_gputest10.log: blocked inside sba_batch_destroy).
// Checks: a wait that times out (SBA_WAIT_TIMEOUT_S=1) returns SBA_ERR_HIP, the handle is then refused by every entry
// point at once, and destroy returns promptly, non-zero, without a single blocking call.  Exit code 0 = all held.
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include <atomic>
#include <thread>

#include "../../include/sba_hip.h"
#include "../../spherical_bundle_adjuster_amd/csrc/sba_device.hpp"
#include "../../spherical_bundle_adjuster_amd/csrc/sba_problem.hpp"

namespace {
bool g_wedged = false;
int g_violations = 0;
std::string g_violation_names;
void blocking_call(const char* name) {
  if (g_wedged) { ++g_violations; g_violation_names += std::string(name) + " "; }
}
void publish(double* host_dev, size_t word, unsigned long long seq) {
  if (!g_wedged && host_dev) reinterpret_cast<volatile unsigned long long*>(host_dev)[word] = seq;
}
}  // namespace

// ---- fake runtime ---------------------------------------------------------------------------------------------------
const char* hipGetErrorString(hipError_t e) { return e == hipSuccess ? "success" : (e == hipErrorNotReady ? "not ready" : "fake error"); }
hipError_t hipGetLastError() { return hipSuccess; }
hipError_t hipGetDeviceCount(int* c) { *c = 1; return hipSuccess; }
hipError_t hipSetDevice(int) { return hipSuccess; }
hipError_t hipGetDeviceProperties(hipDeviceProp_t* p, int) { std::memset(p, 0, sizeof(*p)); std::strcpy(p->name, "mock"); p->multiProcessorCount = 4; return hipSuccess; }
hipError_t hipDeviceGetAttribute(int* v, hipDeviceAttribute_t, int) { *v = 100000; return hipSuccess; }
hipError_t hipDeviceSynchronize() { blocking_call("hipDeviceSynchronize"); return hipSuccess; }
hipError_t hipMalloc(void** p, size_t n) { *p = std::calloc(1, n ? n : 1); return *p ? hipSuccess : hipErrorUnknown; }
hipError_t hipExtMallocWithFlags(void** p, size_t n, unsigned) { return hipMalloc(p, n); }
hipError_t hipFree(void* p) { blocking_call("hipFree"); std::free(p); return hipSuccess; }
hipError_t hipHostMalloc(void** p, size_t n, unsigned) { return hipMalloc(p, n); }
hipError_t hipHostGetDevicePointer(void** d, void* h, unsigned) { *d = h; return hipSuccess; }
hipError_t hipHostFree(void* p) { blocking_call("hipHostFree"); std::free(p); return hipSuccess; }
hipError_t hipMemset(void* p, int v, size_t n) { blocking_call("hipMemset"); std::memset(p, v, n); return hipSuccess; }
hipError_t hipMemsetAsync(void* p, int v, size_t n, hipStream_t) { if (!g_wedged) std::memset(p, v, n); return hipSuccess; }
hipError_t hipMemcpy(void* d, const void* s, size_t n, hipMemcpyKind) { blocking_call("hipMemcpy"); std::memcpy(d, s, n); return hipSuccess; }
hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, hipMemcpyKind, hipStream_t) { if (!g_wedged) std::memcpy(d, s, n); return hipSuccess; }
hipError_t hipStreamCreateWithFlags(hipStream_t* s, unsigned) { *s = reinterpret_cast<hipStream_t>(std::malloc(8)); return hipSuccess; }
hipError_t hipStreamDestroy(hipStream_t s) { blocking_call("hipStreamDestroy"); std::free(s); return hipSuccess; }
hipError_t hipStreamSynchronize(hipStream_t) { blocking_call("hipStreamSynchronize"); return hipSuccess; }
namespace sba { namespace { extern std::atomic<int> g_kernels_running; } }
hipError_t hipStreamQuery(hipStream_t);
hipError_t hipEventCreate(hipEvent_t* e) { *e = reinterpret_cast<hipEvent_t>(std::malloc(8)); return hipSuccess; }
hipError_t hipEventDestroy(hipEvent_t e) { std::free(e); return hipSuccess; }
hipError_t hipEventRecord(hipEvent_t, hipStream_t) { return hipSuccess; }
hipError_t hipEventCreateWithFlags(hipEvent_t* e, unsigned) { return hipEventCreate(e); }
hipError_t hipEventQuery(hipEvent_t) { return g_wedged ? hipErrorNotReady : hipSuccess; }
hipError_t hipEventSynchronize(hipEvent_t) { blocking_call("hipEventSynchronize"); return hipSuccess; }
hipError_t hipEventElapsedTime(float* ms, hipEvent_t, hipEvent_t) { *ms = 1.f; return hipSuccess; }
hipError_t hipIpcGetMemHandle(hipIpcMemHandle_t*, void*) { return hipErrorUnknown; }
hipError_t hipIpcOpenMemHandle(void**, hipIpcMemHandle_t, unsigned) { return hipErrorUnknown; }
hipError_t hipIpcCloseMemHandle(void*) { return hipSuccess; }
hipError_t hipDeviceEnablePeerAccess(int, unsigned) { return hipSuccess; }

// ---- stub kernel launchers (the .hip files are not part of this build) ---------------------------------------------------
namespace sba {
int points_per_lane(int store) { return store == 0 ? 2 : 4; }
hipError_t sweep_blocks_per_cu(int, int, int, int, bool, int* b) { *b = 2; return hipSuccess; }
hipError_t batch_blocks_per_cu(int, int, int, int, bool, int* b) { *b = 2; return hipSuccess; }
hipError_t depth_blocks_per_cu(int, int* b) { *b = 2; return hipSuccess; }
hipError_t launch_sweep(int, int, int, int, const Planes&, const SweepParams&, const SweepOut& o, int, hipStream_t) {
  publish(o.pack_host, 24, o.seq); return hipSuccess;
}
hipError_t launch_finalize(const double*, int, double*, double* host, unsigned long long seq, const PeerInboxes*, unsigned long long, hipStream_t) {
  publish(host, 24, seq); return hipSuccess;
}
hipError_t launch_publish(const double*, double* host, unsigned long long seq, hipStream_t) { publish(host, 24, seq); return hipSuccess; }
hipError_t launch_peer_exchange(const double*, const PeerInboxes&, unsigned long long, double*, double* host, unsigned long long seq, hipStream_t) {
  publish(host, 24, seq); return hipSuccess;
}
hipError_t launch_batch_step_fused(int, int, int, int, double, const Planes&, const BatchState*, const PairDesc*, int np, double*,
                                   double* host, unsigned int*, unsigned long long seq, hipStream_t) {
  publish(host, static_cast<size_t>(24) * np, seq); return hipSuccess;
}
hipError_t launch_batch_lm(int, int, int, int, const Planes&, const PairDesc*, BatchLmIo* io, const sba_lm_options&, int np, unsigned int*,
                           unsigned long long* seq_host, unsigned long long seq, hipStream_t, int, void*, SweepParams*, double*, int*) {
  if (!g_wedged) { for (int g = 0; g < np; ++g) io[g].status = SBA_OK; if (seq_host) *reinterpret_cast<volatile unsigned long long*>(seq_host) = seq; }
  return hipSuccess;
}
hipError_t launch_batch_sweep(int, int, int, int, bool, const Planes&, const SweepParams*, const PairDesc*, int np, int, double*, double*,
                              double* host, unsigned long long seq, hipStream_t) {
  publish(host, static_cast<size_t>(24) * np, seq); return hipSuccess;
}
hipError_t launch_batch_sweep_only(int, int, int, int, bool, const Planes&, const SweepParams*, const PairDesc*, int, int, double*, hipStream_t) { return hipSuccess; }
hipError_t launch_batch_step(int, int, int, int, double, const Planes&, const BatchState*, SweepParams*, double*, const PairDesc*, int np, int,
                             double*, double*, double* host, unsigned long long seq, hipStream_t) {
  publish(host, static_cast<size_t>(24) * np, seq); return hipSuccess;
}
hipError_t launch_aos_to_planes(const double*, size_t, size_t, void*, void*, void*, int, hipStream_t, size_t, size_t) { return hipSuccess; }
hipError_t launch_d12_to_planes(const double*, size_t, size_t, double*, double*, hipStream_t, size_t, size_t) { return hipSuccess; }
hipError_t launch_planes_to_d12(const double*, const double*, size_t, double*, hipStream_t) { return hipSuccess; }
hipError_t launch_batch_aos_to_planes(const double*, size_t, size_t, const unsigned long long*, int, const PairDesc*, size_t, void*, void*, void*,
                                      int, hipStream_t) { return hipSuccess; }
hipError_t launch_batch_d12_to_planes(const double*, size_t, size_t, const unsigned long long*, int, const PairDesc*, size_t, double*, double*,
                                      hipStream_t) { return hipSuccess; }
hipError_t launch_depth_step(int, const Planes&, const double*, const double*, double*, double*, double*, double*, const DepthParams&, double*,
                             int, double*, double* host, unsigned long long seq, int, hipStream_t) {
  publish(host, 24, seq); return hipSuccess;
}
hipError_t launch_epipolar_moments(int, const Planes&, size_t, double*, int, double*, hipStream_t) { return hipSuccess; }
hipError_t launch_batch_epipolar_moments(int, const Planes&, const PairDesc*, int, double*, hipStream_t) { return hipSuccess; }
hipError_t launch_batch_guess(const double*, int, int, double, unsigned long long, BatchGuessOut*, hipStream_t) { return hipSuccess; }
hipError_t launch_batch_depth_step(int, const Planes&, const PairDesc*, const BatchDepthConst*, const BatchDepthPass*, int np, double, double,
                                   double, double, double*, double*, double*, double*, double*, double*, double* host, unsigned int*,
                                   unsigned long long seq, hipStream_t) {
  publish(host, static_cast<size_t>(DEPTH_ROW) * np, seq); return hipSuccess;
}
hipError_t launch_batch_depth_solve(int, const Planes&, const PairDesc*, const BatchDepthConst*, int, double, double, const sba_lm_options&,
                                    double*, double*, double*, double*, double*, double*, const unsigned long long*, double*, BatchLmIo*,
                                    unsigned int*, unsigned long long* seq_host, unsigned long long seq, hipStream_t, int, void*, BatchDepthPass*,
                                    int*, unsigned char*) {
  publish(reinterpret_cast<double*>(seq_host), 0, seq); return hipSuccess;
}
size_t batch_depth_dyn_state_bytes() { return 1024; }
hipError_t launch_batch_depth_dyn_pass(int, const Planes&, const PairDesc*, const BatchDepthConst*, int, double, double, const sba_lm_options&,
                                       double*, double*, double*, double*, double*, double*, int, int, void*, BatchDepthPass*, const BatchDynCtl*,
                                       const unsigned int*, int*, unsigned char*, double*, BatchLmIo*, hipStream_t) { return hipSuccess; }
hipError_t launch_batch_dyn_compact(BatchDynCtl*, unsigned int*, const int*, int, int, unsigned long long* host_words, unsigned long long seq,
                                    hipStream_t) {
  if (host_words && !g_wedged) host_words[0] = 0;
  publish(reinterpret_cast<double*>(host_words), 1, seq); return hipSuccess;
}
size_t batch_lm_dyn_state_bytes() { return 2048; }
hipError_t launch_batch_dyn_first_list(BatchDynCtl*, unsigned int*, const int*, int, hipStream_t) { return hipSuccess; }
hipError_t launch_batch_lm_dyn_init(int, int, int, const PairDesc*, const BatchLmIo*, const sba_lm_options&, int, void*, SweepParams*, double*,
                                    BatchDynCtl*, unsigned int*, int*, hipStream_t) { return hipSuccess; }
hipError_t launch_batch_lm_dyn_pass(int, int, int, int, const Planes&, const PairDesc*, const sba_lm_options&, int, int, int, void*, SweepParams*,
                                    double*, BatchDynCtl*, unsigned int*, int*, double*, BatchLmIo*, unsigned long long* host_words,
                                    unsigned long long seq, hipStream_t) {
  if (host_words && !g_wedged) host_words[0] = 0;          // "no pair is still iterating"
  publish(reinterpret_cast<double*>(host_words), 1, seq); return hipSuccess;
}
hipError_t launch_batch_depth_finish(int, const PairDesc*, const unsigned char*, int, double*, double*, const double*, const double*,
                                     const unsigned long long*, double*, hipStream_t) { return hipSuccess; }
// ---- the resident kernels, emulated by a host thread that speaks the device side of the protocol (sba_resident.hpp) -------
// Same record decoding as resident_wait_command (check word per line), same publication order as resident_publish, same
// ends (QUIT command, idle time-out, trip budget).  The "sweep" it answers with is a fixed function of the command's
// payload, so the caller can tell that command k got answer k.  A launch on the mock's one stream runs in launch order:
// a new emulated kernel first joins the previous one (a real stream serialises them the same way).
namespace {
std::thread g_kernel;
std::atomic<int> g_kernels_launched{0}, g_kernels_running{0};
int g_mock_trip_budget = kResidentMaxTrips;
bool read_record(const ResidentRecord* rec, unsigned long long expect, double payload[kResidentPayload]) {
  for (int l = 0; l < kResidentLines; ++l) {
    const volatile unsigned long long* line = reinterpret_cast<const volatile unsigned long long*>(rec->w[l]);
    unsigned long long w[8];
    for (int k = 0; k < 8; ++k) w[k] = line[k];
    unsigned long long x = w[7];
    for (int k = 0; k < 7; ++k) x ^= resident_fold(w[k], k);
    if (x != expect) return false;
    for (int k = 0; k < 7; ++k) std::memcpy(&payload[7 * l + k], &w[k], 8);
  }
  return true;
}
void emulated_resident_kernel(const ResidentRecord* rec, double* host_pack, unsigned long long first_cmd, unsigned long long first_pack,
                              unsigned long long idle_ticks, int khz, int want_op) {
  volatile unsigned long long* words = reinterpret_cast<volatile unsigned long long*>(host_pack);
  const double idle_s = static_cast<double>(idle_ticks) / (1e3 * khz);
  int end = RESIDENT_END_TRIPS;
  for (int trip = 0; trip < g_mock_trip_budget; ++trip) {
    double payload[kResidentPayload];
    const auto t0 = std::chrono::steady_clock::now();
    bool got = false;
    while (!(got = read_record(rec, first_cmd + trip, payload))) {
      if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > idle_s) break;
      std::this_thread::yield();
    }
    if (!got) { end = RESIDENT_END_IDLE; break; }
    const int op = static_cast<int>(payload[0]);
    if (op == RESIDENT_OP_QUIT) { end = RESIDENT_END_QUIT; break; }
    if (op != want_op) { end = RESIDENT_END_BAD_OP; break; }
    for (int k = 0; k < 24; ++k) host_pack[k] = payload[1 + k] + 1000.0 * k;      // the "sums"
    __atomic_thread_fence(__ATOMIC_RELEASE);
    words[24] = first_pack + trip;
  }
  __atomic_thread_fence(__ATOMIC_RELEASE);
  words[kResidentEndWord] = static_cast<unsigned long long>(end);
  g_kernels_running.fetch_sub(1);
}
hipError_t launch_emulated(const ResidentRecord* rec, double* host_pack, unsigned long long first_cmd, unsigned long long first_pack,
                           unsigned long long idle_ticks, int want_op) {
  if (g_wedged) return hipSuccess;            // a wedged device never runs what is launched on it
  if (g_kernel.joinable()) g_kernel.join();   // stream order
  ++g_kernels_launched;
  g_kernels_running.fetch_add(1);
  g_kernel = std::thread(emulated_resident_kernel, rec, host_pack, first_cmd, first_pack, idle_ticks, 100000, want_op);
  return hipSuccess;
}
}  // namespace
hipError_t launch_resident_sweep(int, int, int, int, bool, const Planes&, size_t, const ResidentRecord* rec, double* host_pack,
                                 unsigned long long first_cmd, unsigned long long first_pack, unsigned long long idle_ticks, hipStream_t) {
  return launch_emulated(rec, host_pack, first_cmd, first_pack, idle_ticks, RESIDENT_OP_SWEEP);
}
hipError_t launch_resident_depth(int, const Planes&, size_t, double*, double*, double*, double*, double*, double*, const ResidentRecord* rec,
                                 double* host_pack, unsigned long long first_cmd, unsigned long long first_pack,
                                 unsigned long long idle_ticks, hipStream_t) {
  return launch_emulated(rec, host_pack, first_cmd, first_pack, idle_ticks, RESIDENT_OP_DEPTH);
}
hipError_t launch_epipolar_subset_moments(int, const Planes&, size_t, const int*, int, int, double*, hipStream_t) { return hipSuccess; }
hipError_t launch_keypoints_to_sphere(const uint8_t*, size_t, size_t, double, double, double*, hipStream_t) { return hipSuccess; }
hipError_t launch_keypoints_to_planes(const uint8_t*, const uint8_t*, size_t, size_t, double, double, void* const*, int, hipStream_t) { return hipSuccess; }
}  // namespace sba

hipError_t hipStreamQuery(hipStream_t) { return (g_wedged || sba::g_kernels_running.load() > 0) ? hipErrorNotReady : hipSuccess; }

#define REQUIRE(c) do { if (!(c)) { std::fprintf(stderr, "wedge_harness: check failed: %s (line %d); last error: %s\n", #c, __LINE__, sba_last_error()); return 1; } } while (0)

static double seconds_since(std::chrono::steady_clock::time_point t0) {
  return std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
}

int main() {
  const size_t n = 100;
  std::vector<double> x(3 * n, 0.5), d12(2 * n, 1.0), pack(24);
  const double rot[3] = {0.1, 0.2, 0.3}, tran[3] = {0, 0, 1};

  // ---- single problem: healthy sweep, then the device stops answering in the middle of the next one ----------------------------
  {
    sba_problem* p = nullptr;
    REQUIRE(sba_problem_create(&p, 0, nullptr) == SBA_OK);
    REQUIRE(sba_problem_upload(p, x.data(), x.data(), d12.data(), n, SBA_STORE_F64) == SBA_OK);
    REQUIRE(sba_problem_eval_pack(p, SBA_MODE_RT, SBA_DEPTH_PER_MATCH, rot, tran, 1, 1, 1.0, pack.data()) == SBA_OK);
    g_wedged = true;
    auto t0 = std::chrono::steady_clock::now();
    REQUIRE(sba_problem_eval_pack(p, SBA_MODE_RT, SBA_DEPTH_PER_MATCH, rot, tran, 1, 1, 1.0, pack.data()) == SBA_ERR_HIP);
    REQUIRE(std::strstr(sba_last_error(), "SBA_WAIT_TIMEOUT_S") != nullptr);
    REQUIRE(seconds_since(t0) < 10.0);
    // refused at once from now on, whatever the entry point
    t0 = std::chrono::steady_clock::now();
    REQUIRE(sba_problem_eval_pack(p, SBA_MODE_RT, SBA_DEPTH_PER_MATCH, rot, tran, 1, 1, 1.0, pack.data()) == SBA_ERR_HIP);
    REQUIRE(std::strstr(sba_last_error(), "poisoned") != nullptr);
    double r2[3] = {0.1, 0.2, 0.3}, t2[3] = {0, 0, 1};
    REQUIRE(sba_problem_solve(p, SBA_MODE_RT, SBA_DEPTH_PER_MATCH, r2, t2, 1, 1, nullptr, nullptr) == SBA_ERR_HIP);
    REQUIRE(sba_problem_upload(p, x.data(), x.data(), nullptr, n, SBA_STORE_F64) == SBA_ERR_HIP);
    REQUIRE(sba_problem_set_depths(p, d12.data()) == SBA_ERR_HIP);
    REQUIRE(sba_problem_solve_depths(p, rot, tran, 1, 1, nullptr, nullptr, nullptr) == SBA_ERR_HIP);
    std::vector<double> groups(64 * 45);
    REQUIRE(sba_problem_epipolar_moments(p, groups.data()) == SBA_ERR_HIP);
    char handle[SBA_PEER_HANDLE_BYTES];
    REQUIRE(sba_problem_peer_export(p, 2, 0, handle) == SBA_ERR_HIP);
    REQUIRE(sba_problem_destroy(p) == SBA_ERR_HIP);          // leaks, says so, returns
    REQUIRE(std::strstr(sba_last_error(), "leaked") != nullptr);
    REQUIRE(seconds_since(t0) < 0.5);
    REQUIRE(g_violations == 0);
    g_wedged = false;
  }
  // ---- the d-only stage wedges in its first pass: its scratch planes must be leaked, not freed ---------------------------------
  {
    sba_problem* p = nullptr;
    REQUIRE(sba_problem_create(&p, 0, nullptr) == SBA_OK);
    REQUIRE(sba_problem_upload(p, x.data(), x.data(), d12.data(), n, SBA_STORE_F64) == SBA_OK);
    g_wedged = true;
    REQUIRE(sba_problem_solve_depths(p, rot, tran, 1, 1, nullptr, nullptr, nullptr) == SBA_ERR_HIP);
    REQUIRE(sba_problem_destroy(p) == SBA_ERR_HIP);
    REQUIRE(g_violations == 0);
    g_wedged = false;
  }
  // ---- a handle that is healthy until its destroy: the bounded drain inside destroy poisons it, destroy still returns -----
  {
    sba_problem* p = nullptr;
    REQUIRE(sba_problem_create(&p, 0, nullptr) == SBA_OK);
    REQUIRE(sba_problem_upload(p, x.data(), x.data(), nullptr, n, SBA_STORE_F64) == SBA_OK);
    g_wedged = true;
    const auto t0 = std::chrono::steady_clock::now();
    REQUIRE(sba_problem_destroy(p) == SBA_ERR_HIP);
    REQUIRE(seconds_since(t0) < 10.0 && g_violations == 0);
    g_wedged = false;
  }
  // ---- batch: one-launch step and the one-launch per-pair solve ---------------------------------------------------------------------
  for (int which = 0; which < 2; ++which) {
    sba_batch* b = nullptr;
    const int B = 4;      // = the mock's CU count -> one block per pair, fused step, device LM
    std::vector<size_t> off(B + 1);
    for (int g = 0; g <= B; ++g) off[g] = static_cast<size_t>(g) * (n / B);
    std::vector<double> rots(3 * B, 0.1), trans(3 * B, 0.2), packs(24 * B);
    REQUIRE(sba_batch_create(&b, 0, nullptr) == SBA_OK);
    REQUIRE(sba_batch_upload(b, x.data(), x.data(), d12.data(), off.data(), B, SBA_STORE_F64) == SBA_OK);
    REQUIRE(sba_batch_eval(b, SBA_MODE_RT, SBA_DEPTH_PER_MATCH, rots.data(), trans.data(), nullptr, nullptr, 1.0, packs.data()) == SBA_OK);
    g_wedged = true;
    if (which == 0)
      REQUIRE(sba_batch_eval(b, SBA_MODE_RT, SBA_DEPTH_PER_MATCH, rots.data(), trans.data(), nullptr, nullptr, 1.0, packs.data()) == SBA_ERR_HIP);
    else
      REQUIRE(sba_batch_solve(b, SBA_MODE_RT, SBA_DEPTH_PER_MATCH, rots.data(), trans.data(), nullptr, nullptr, nullptr, nullptr, nullptr) == SBA_ERR_HIP);
    const auto t0 = std::chrono::steady_clock::now();
    REQUIRE(sba_batch_eval(b, SBA_MODE_RT, SBA_DEPTH_PER_MATCH, rots.data(), trans.data(), nullptr, nullptr, 1.0, packs.data()) == SBA_ERR_HIP);
    REQUIRE(std::strstr(sba_last_error(), "poisoned") != nullptr);
    REQUIRE(sba_batch_upload(b, x.data(), x.data(), d12.data(), off.data(), B, SBA_STORE_F64) == SBA_ERR_HIP);
    REQUIRE(sba_batch_destroy(b) == SBA_ERR_HIP);
    REQUIRE(seconds_since(t0) < 0.5);
    REQUIRE(g_violations == 0);
    g_wedged = false;
  }
  // ---- resident evaluator: the host side of the protocol against a host thread that speaks the device side -----------------------
  {
    using sba::shim::ResidentSession;
    sba_problem* p = nullptr;
    REQUIRE(sba_problem_create(&p, 0, nullptr) == SBA_OK);
    REQUIRE(sba_problem_upload(p, x.data(), x.data(), d12.data(), n, SBA_STORE_F64) == SBA_OK);
    REQUIRE(sba::shim::resident_eligible(p, false) && sba::shim::resident_eligible(p, true));
    p->resident_idle_s = 0.05;
    const int launched0 = sba::g_kernels_launched.load();
    double payload[44], got[24];
    auto check = [&](ResidentSession& s, int k) -> bool {
      payload[0] = sba::RESIDENT_OP_SWEEP;
      for (int i = 1; i < 44; ++i) payload[i] = 0.25 * i + k;
      if (s.call(payload, 44, got, 24) != SBA_OK) return false;
      for (int i = 0; i < 24; ++i) if (got[i] != payload[1 + i] + 1000.0 * i) return false;
      return true;
    };
    {
      ResidentSession s(p);
      REQUIRE(s.start_sweep(SBA_MODE_RT, SBA_DEPTH_PER_MATCH, true) == SBA_OK);
      for (int k = 0; k < 200; ++k) REQUIRE(check(s, k));                       // back to back: one kernel serves them all
      REQUIRE(sba::g_kernels_launched.load() == launched0 + 1);
      std::this_thread::sleep_for(std::chrono::milliseconds(200));             // the host goes away: the kernel ends itself ...
      REQUIRE(sba::g_kernels_running.load() == 0);
      for (int k = 0; k < 5; ++k) REQUIRE(check(s, 1000 + k));                  // ... and is restarted on the pending command
      REQUIRE(sba::g_kernels_launched.load() == launched0 + 2);
      REQUIRE(s.end() == SBA_OK);
      REQUIRE(sba::g_kernels_running.load() == 0);                              // QUIT was seen, the stream has drained
    }
    {
      sba::g_mock_trip_budget = 7;                                             // trip budget used up mid-session: restarted too
      ResidentSession s(p);
      REQUIRE(s.start_sweep(SBA_MODE_ROT, SBA_DEPTH_UNIFORM, true) == SBA_OK);
      for (int k = 0; k < 30; ++k) REQUIRE(check(s, k));
      REQUIRE(sba::g_kernels_launched.load() >= launched0 + 2 + 4);
      sba::g_mock_trip_budget = sba::kResidentMaxTrips;
    }                                                                          // destructor ends the session
    REQUIRE(sba::g_kernels_running.load() == 0);
    {
      ResidentSession s(p);                                                     // a kernel that never answers: bounded, poisons
      REQUIRE(s.start_sweep(SBA_MODE_ROT, SBA_DEPTH_UNIFORM, true) == SBA_OK);
      REQUIRE(check(s, 1));
      g_wedged = true;                                                         // (restarts are swallowed like every launch)
      std::this_thread::sleep_for(std::chrono::milliseconds(200));
      payload[0] = sba::RESIDENT_OP_SWEEP;
      REQUIRE(s.call(payload, 44, got, 24) == SBA_ERR_HIP);
      REQUIRE(p->poisoned == 1);
    }
    REQUIRE(sba_problem_destroy(p) == SBA_ERR_HIP);
    REQUIRE(g_violations == 0);
    g_wedged = false;
    if (sba::g_kernel.joinable()) sba::g_kernel.join();
  }
  // ---- and a healthy life cycle still frees everything (destroy returns SBA_OK) --------------------------------------------------
  {
    sba_problem* p = nullptr;
    REQUIRE(sba_problem_create(&p, 0, nullptr) == SBA_OK);
    REQUIRE(sba_problem_upload(p, x.data(), x.data(), d12.data(), n, SBA_STORE_F64) == SBA_OK);
    REQUIRE(sba_problem_eval_pack(p, SBA_MODE_ROT, SBA_DEPTH_UNIFORM, rot, tran, 1, 1, 1.0, pack.data()) == SBA_OK);
    REQUIRE(sba_problem_destroy(p) == SBA_OK);
  }
  if (g_violations) { std::fprintf(stderr, "blocking calls on a wedged device: %s\n", g_violation_names.c_str()); return 1; }
  std::printf("wedge_harness: ok\n");
  return 0;
}
