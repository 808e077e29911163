// Host side of the resident evaluator (sba_resident.hpp): session life cycle and the command / answer hand-shake.
#include <chrono>
#include <cstdlib>
#include <cstring>

#include "sba_problem.hpp"

namespace sba {
namespace shim {

bool resident_eligible(const sba_problem* p, bool depth_stage) {
  return p->res_rec != nullptr && p->publish && !is_collective(p) && p->uploaded && p->n > 0 &&
         p->n <= (depth_stage ? p->resident_max_n_depth : p->resident_max_n);
}

int ResidentSession::launch() {
  sba_problem* p = p_;
  sba::Planes pl;
  for (int k = 0; k < 3; ++k) { pl.x1[k] = p->coord[k]; pl.x2[k] = p->coord[3 + k]; }
  pl.d1 = p->dplane[0]; pl.d2 = p->dplane[1];
  reinterpret_cast<volatile unsigned long long*>(p->pack_host)[kResidentEndWord] = 0;
  __atomic_thread_fence(__ATOMIC_SEQ_CST);
  const unsigned long long idle_ticks = static_cast<unsigned long long>(p->resident_idle_s * 1e3 * p->wall_clock_khz);
  // the kernel waits for the NEXT command to be written (res_cmd_seq + 1) -- or, when a session is restarted with a
  // command pending, for that very command -- and numbers its answers from the next pack sequence number
  if (depth_)
    SBA_TRY_HIP(launch_resident_depth(p->store, pl, p->n, a1_, a2_, b1_, b2_, sc1_, sc2_, p->res_rec_dev, p->pack_host_dev,
                                      pending_cmd_, pending_pack_, idle_ticks, p->stream));
  else
    SBA_TRY_HIP(launch_resident_sweep(mode_, depth_mode_, p->store, p->kind, loss_, pl, p->n, p->res_rec_dev, p->pack_host_dev,
                                      pending_cmd_, pending_pack_, idle_ticks, p->stream));
  return SBA_OK;
}

int ResidentSession::start_sweep(int mode, int depth_mode, bool loss) {
  depth_ = false; mode_ = mode; depth_mode_ = depth_mode; loss_ = loss;
  pending_cmd_ = p_->res_cmd_seq + 1;
  pending_pack_ = p_->seq + 1;
  const int rc = launch();
  active_ = rc == SBA_OK;
  return rc;
}

int ResidentSession::start_depth(double* a1, double* a2, double* b1, double* b2, double* sc1, double* sc2) {
  depth_ = true; a1_ = a1; a2_ = a2; b1_ = b1; b2_ = b2; sc1_ = sc1; sc2_ = sc2;
  pending_cmd_ = p_->res_cmd_seq + 1;
  pending_pack_ = p_->seq + 1;
  const int rc = launch();
  active_ = rc == SBA_OK;
  return rc;
}

// Send one command, wait for its answer.  While waiting also watch the kernel's end word: a kernel that ended itself (no
// command for resident_idle_s -- the host thread was descheduled -- or its trip budget used up) is restarted on the
// pending command; the wait as a whole is bounded by SBA_WAIT_TIMEOUT_S like every other wait for the device.
int ResidentSession::call(const double* payload, int count, double* out, int out_count) {
  sba_problem* p = p_;
  if (!active_) return sba::set_error(SBA_ERR_INVALID_ARG, "resident session is not open");
  const unsigned long long cmd = ++p->res_cmd_seq;
  const unsigned long long ans = ++p->seq;
  resident_write_command(p->res_rec, payload, count, cmd);
  volatile unsigned long long* words = reinterpret_cast<volatile unsigned long long*>(p->pack_host);
  static const double limit_s = [] {
    const char* env = std::getenv("SBA_WAIT_TIMEOUT_S");
    const double v = env ? std::atof(env) : 60.0;
    return v > 0.0 ? v : 60.0;
  }();
  std::chrono::steady_clock::time_point t0;
  bool timing = false;
  for (unsigned long spins = 0; words[24] != ans; ++spins) {
    if ((spins & 0x3ff) == 0x3ff) {
      const unsigned long long ended = words[kResidentEndWord];
      if (ended != 0 && words[24] != ans) {
        if (ended != RESIDENT_END_IDLE && ended != RESIDENT_END_TRIPS) {
          active_ = false;
          return sba::set_error(SBA_ERR_HIP, "resident kernel ended unexpectedly (reason %llu)", ended);
        }
        const int rc = sba::stream_wait(p->stream, "resident kernel restart", &p->poisoned);
        if (rc) { active_ = false; return rc; }
        if (words[24] == ans) break;             // it answered just before it ended
        pending_cmd_ = cmd; pending_pack_ = ans;
        const int lrc = launch();
        if (lrc) { active_ = false; return lrc; }
      }
      const auto now = std::chrono::steady_clock::now();
      if (!timing) { t0 = now; timing = true; }
      else if (std::chrono::duration<double>(now - t0).count() > limit_s) {
        p->poisoned = 1;
        active_ = false;
        return sba::set_error(SBA_ERR_HIP, "resident kernel: no answer after %.0f s (SBA_WAIT_TIMEOUT_S)", limit_s);
      }
    }
    __builtin_ia32_pause();
  }
  __atomic_thread_fence(__ATOMIC_ACQUIRE);
  std::memcpy(out, p->pack_host, sizeof(double) * static_cast<size_t>(out_count));
  return SBA_OK;
}

int ResidentSession::end() {
  if (!active_) return SBA_OK;
  active_ = false;
  sba_problem* p = p_;
  if (p->poisoned) return SBA_OK;          // nothing may wait on that stream again; the kernel ends itself after idle_s
  const double quit[1] = {static_cast<double>(RESIDENT_OP_QUIT)};
  resident_write_command(p->res_rec, quit, 1, ++p->res_cmd_seq);
  return sba::stream_wait(p->stream, "resident kernel shut-down", &p->poisoned);
}

}  // namespace shim
}  // namespace sba
