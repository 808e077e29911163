// Resident evaluator (small problems: the reference's real workload, ~2 k matches, BASELINE config C1).
//
// A host-synchronous LM iteration of a 2 048-match problem is ~11-17 us on the launch-per-sweep path, of which the sweep
// itself is ~2 us: the rest is two kernel launches and their completion latency, every iteration (and the d-only stage
// makes ~27 such passes before the first sweep).  For problems one block can sweep in a few microseconds the solve
// entry points therefore start ONE single-block kernel per stage that stays resident and is driven by the host through
// mapped pinned memory: the host (which keeps running the very same LM / d-only state machines) writes a command
// record -- the sweep state of the next query point -- and the kernel answers with the reduced pack, exactly as a
// sweep launch would have.  No launch, no kernel boundary per iteration; the host <-> device hand-off is one PCIe read
// (the kernel polling the record) and one posted PCIe write burst (the pack + its sequence word).
//
// Command record (host -> device): kResidentLines cache lines of 8 doubles; the first 7 of every line are payload, the
// 8th is a CHECK word = the command's sequence number XOR a position-dependent fold of the line's 7 payload words.  The
// kernel's wave 0 reads the whole record with ONE wave-wide system-scope load and accepts it when every line's check
// word matches the sequence number it waits for -- no separate doorbell round trip, and no assumption about the order or
// granularity in which the host's stores become visible to a PCIe read: a line that is half old, half new fails its
// check (unless old and new payload coincide, in which case it is the new payload) and is simply read again.  Payload
// word 0 is the opcode.
//
// Safety: the kernel cannot outlive its use.  Every poll loop is bounded in TIME (wall_clock64: `idle_ticks` without a
// command -> the kernel ends itself and says so in the host pack), the trip count is bounded by a counter all threads
// keep, the host ends a session with an explicit QUIT command and waits for the stream to drain (bounded), and a
// kernel that ended on its own is restarted transparently by the host on the next command.
#pragma once
#include "sba_device.hpp"
#if defined(__HIPCC__)
#include "sba_publish.hpp"
#endif

namespace sba {

constexpr int kResidentLines = 8;
constexpr int kResidentPayload = 7 * kResidentLines;     // 56 doubles
struct alignas(64) ResidentRecord { double w[kResidentLines][8]; };
static_assert(sizeof(ResidentRecord) == 512, "one wave-wide 8-byte load covers the record");

enum : int { RESIDENT_OP_SWEEP = 1, RESIDENT_OP_DEPTH = 2, RESIDENT_OP_QUIT = 3 };
// why a resident kernel ended; published in host pack word kResidentEndWord (0 while it runs)
enum : int { RESIDENT_END_QUIT = 1, RESIDENT_END_IDLE = 2, RESIDENT_END_TRIPS = 3, RESIDENT_END_BAD_OP = 4 };
constexpr int kResidentEndWord = 26;      // of the 32-double mapped host pack: [0..23] pack, [24] sequence, [25] peer flag
constexpr int kResidentMaxTrips = 4096;   // commands per session; the host starts a new session before it runs out

// Host side: write a command (payload[0] = opcode) with sequence number `seq` into the mapped record.
SBA_HD inline unsigned long long resident_fold(unsigned long long bits, int k) {     // k = 0..6: position in the line
  return (bits << (k + 1)) | (bits >> (63 - k));
}
inline void resident_write_command(ResidentRecord* rec, const double* payload, int count, unsigned long long seq) {
  for (int l = 0; l < kResidentLines; ++l) {
    volatile unsigned long long* line = reinterpret_cast<volatile unsigned long long*>(rec->w[l]);
    unsigned long long check = seq;
    for (int k = 0; k < 7; ++k) {
      const int idx = 7 * l + k;
      const double v = idx < count ? payload[idx] : 0.0;
      unsigned long long bits;
      __builtin_memcpy(&bits, &v, sizeof(bits));
      line[k] = bits;
      check ^= resident_fold(bits, k);
    }
    line[7] = check;
  }
  __atomic_thread_fence(__ATOMIC_RELEASE);
}

#if defined(__HIPCC__)
// Device side, wave 0 only (all 64 lanes): wait for command `expect` and copy its payload to cmd_s[kResidentPayload] (LDS;
// the opcode is cmd_s[0], to be read after the caller's barrier).  Returns true when the command arrived, false -- with
// *end = RESIDENT_END_IDLE -- when none did within idle_ticks of the wall clock.
__device__ __forceinline__ bool resident_wait_command(const ResidentRecord* __restrict__ rec, unsigned long long expect,
                                                      unsigned long long idle_ticks, double* __restrict__ cmd_s, int* end) {
  const int lane = threadIdx.x & 63;
  const unsigned long long* words = reinterpret_cast<const unsigned long long*>(rec);
  const long long t0 = wall_clock64();
  unsigned long long v = 0;
  bool got = false;
  for (;;) {
    v = __hip_atomic_load(words + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);     // 512 B: the whole record
    // per line (8 consecutive lanes): check word XOR the folds of the 7 payload words must give the sequence number
    unsigned long long x = (lane & 7) == 7 ? v : resident_fold(v, lane & 7);
    x ^= __shfl_xor(x, 1); x ^= __shfl_xor(x, 2); x ^= __shfl_xor(x, 4);
    if (__builtin_amdgcn_ballot_w64(x == expect) == ~0ull) { got = true; break; }
    if (static_cast<unsigned long long>(wall_clock64() - t0) > idle_ticks) break;
    __builtin_amdgcn_s_sleep(8);
  }
  if (!got) { *end = RESIDENT_END_IDLE; return false; }
  if ((lane & 7) != 7) {
    double d;
    __builtin_memcpy(&d, &v, sizeof(d));
    cmd_s[7 * (lane >> 3) + (lane & 7)] = d;
  }
  return true;
}

// Device side, wave 0: `count` (<= 24) result doubles from LDS -> mapped host pack, system-scope release, then `seq`.
__device__ __forceinline__ void resident_publish(double* __restrict__ host_pack, const double* __restrict__ res_s, int count,
                                                 unsigned long long seq) {
  const int lane = threadIdx.x & 63;
  if (lane < count) host_store(host_pack + lane, res_s[lane]);
  host_release();
  if (lane == 0)
    __hip_atomic_store(reinterpret_cast<unsigned long long*>(host_pack + 24), seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
__device__ __forceinline__ void resident_publish_end(double* __restrict__ host_pack, int reason) {
  if ((threadIdx.x & 63) == 0) {
    __hip_atomic_store(reinterpret_cast<unsigned long long*>(host_pack + kResidentEndWord), static_cast<unsigned long long>(reason),
                       __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");
  }
}
#endif

// launchers (sba_batch_kernels.hip, sba_depth.hip).  first_cmd_seq: sequence number of the first command the kernel waits
// for; first_pack_seq: the number it publishes with its first answer (both count up by one per command).
hipError_t launch_resident_sweep(int mode, int depth, int store, int kind, bool loss, const Planes& pl, size_t n,
                                 const ResidentRecord* rec_dev, double* host_pack_dev, unsigned long long first_cmd_seq,
                                 unsigned long long first_pack_seq, unsigned long long idle_ticks, hipStream_t stream);
// d-only stage: planes a1/a2 and b1/b2 alternate as current / candidate depths (the command says which), sc1/sc2 hold the
// Jacobi scaling.
hipError_t launch_resident_depth(int store, const Planes& pl, size_t n, double* a1, double* a2, double* b1, double* b2,
                                 double* sc1, double* sc2, const ResidentRecord* rec_dev, double* host_pack_dev,
                                 unsigned long long first_cmd_seq, unsigned long long first_pack_seq,
                                 unsigned long long idle_ticks, hipStream_t stream);

}  // namespace sba
