// Which trials of the batched initial guess are slow on the device?  One block of 128 lanes on a synthetic pair's moments;
// every wave runs epi::group_trial for its 64 trial ids, thread 0 of each wave clocks it.  Second launch: trial ids swapped
// between the waves; third: wave 1 alone.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <random>
#include <vector>
#include "sba_epipolar.hpp"
using namespace sba::epi;

__global__ void probe(const double* groups, unsigned long long seed, int mode, long long* ticks, float* sink) {
  __shared__ GroupOccupancy occ;
  __shared__ double g_s[kGroups * kMom];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  for (int k = tid; k < kGroups * kMom; k += blockDim.x) g_s[k] = groups[k];
  __syncthreads();
  if (tid == 0) group_occupancy(g_s, 0.25, &occ);
  __syncthreads();
  int trial = mode == 1 ? ((1 - wave) * 64 + lane) : tid;
  const bool active = mode == 2 ? wave == 1 : (mode == 3 ? tid < 80 : true);
  const long long t0 = wall_clock64();
  TrialOut o{};
  if (active) group_trial(g_s, occ, seed, trial, &o);
  const long long t1 = wall_clock64();
  sink[tid] = o.c1.euler[0] + o.c2.euler[1];
  if (lane == 0) ticks[wave] = t1 - t0;
}

int main() {
  const int n = 50000;
  std::mt19937_64 rng(5);
  std::normal_distribution<double> nd;
  std::vector<double> groups(kGroups * kMom, 0.0);
  for (int i = 0; i < n; ++i) {
    double x[3] = {nd(rng), nd(rng), nd(rng)}, nn = std::sqrt(x[0] * x[0] + x[1] * x[1] + x[2] * x[2]);
    for (double& v : x) v /= nn;
    const double d = 2.0 + (rng() % 1000) / 250.0;
    double y[3] = {x[0] * d + 0.3, x[1] * d - 0.2 + 0.05 * x[2] * d, x[2] * d + 0.1 - 0.05 * x[1] * d};
    nn = std::sqrt(y[0] * y[0] + y[1] * y[1] + y[2] * y[2]);
    for (double& v : y) v = v / nn + 1e-3 * nd(rng);
    double row[9];
    for (int a = 0; a < 3; ++a) for (int b = 0; b < 3; ++b) row[3 * a + b] = x[a] * y[b];
    const int g = (i / 2) % kGroups;
    int k = 0;
    for (int a = 0; a < 9; ++a) for (int b = a; b < 9; ++b) groups[g * kMom + k++] += row[a] * row[b];
  }
  double* gd; long long* td; float* sd;
  hipMalloc(&gd, groups.size() * 8); hipMalloc(&td, 64); hipMalloc(&sd, 1024);
  hipMemcpy(gd, groups.data(), groups.size() * 8, hipMemcpyHostToDevice);
  const char* what[4] = {"wave w runs trials 64 w + lane", "trial ids swapped between the waves", "wave 1 alone (trials 64..127)", "80 lanes (as the kernel)"};
  for (int rep = 0; rep < 2; ++rep)
    for (int mode = 0; mode < 4; ++mode) {
      hipLaunchKernelGGL(probe, dim3(1), dim3(128), 0, 0, gd, 1ull, mode, td, sd);
      hipDeviceSynchronize();
      long long t[2];
      hipMemcpy(t, td, sizeof(t), hipMemcpyDeviceToHost);
      std::printf("%-40s wave 0: %7.1f us   wave 1: %7.1f us\n", what[mode], t[0] / 100.0, t[1] / 100.0);
    }
  return 0;
}
