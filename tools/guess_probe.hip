// Where does a trial of the batched initial guess spend its time on the device?  One block, 80 lanes, each running
// epi::group_trial's steps on the moments of a synthetic pair, thread 0's 100 MHz wall clock around every step.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -I spherical_bundle_adjuster_amd/csrc tools/guess_probe.hip -o tools/build/guess_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <random>
#include <vector>
#include "sba_epipolar.hpp"

using namespace sba::epi;

__global__ void probe(const double* groups, unsigned long long seed, long long* ticks, float* sink) {
  __shared__ GroupOccupancy occ;
  const int tid = threadIdx.x;
  long long t0 = wall_clock64();
  if (tid == 0) group_occupancy(groups, 0.25, &occ);
  __syncthreads();
  long long t1 = wall_clock64();
  int sel[kGroups];
  trial_groups(seed, tid, occ.ne, sel, occ.nonempty, occ.ne);
  int used = occ.take;
  for (int i = 1; i < used; ++i) { const int v = sel[i]; int j = i - 1; for (; j >= 0 && sel[j] > v; --j) sel[j + 1] = sel[j]; sel[j + 1] = v; }
  double mom[kMom];
  for (int k = 0; k < kMom; ++k) mom[k] = 0.0;
  for (int s = 0; s < used; ++s) for (int k = 0; k < kMom; ++k) mom[k] += groups[sel[s] * kMom + k];
  long long t2 = wall_clock64();
  double S[81];
  { int k = 0; for (int a = 0; a < 9; ++a) for (int b = a; b < 9; ++b) { S[9 * a + b] = S[9 * b + a] = mom[k]; ++k; } }
  double E[9], lam;
  const bool ok = smallest_eigvec(9, S, E, &lam);
  long long t3 = wall_clock64();
  double U[9], sv[3], Vt[9], T[9], Ec[9];
  svd3(E, U, sv, Vt);
  const double D[9] = {sv[0], 0, 0, 0, sv[1], 0, 0, 0, 0};
  mul3(U, D, T); mul3(T, Vt, Ec);
  long long t4 = wall_clock64();
  double R1[9], R2[9], t[3];
  decompose_essential(Ec, R1, R2, t);
  long long t5 = wall_clock64();
  float e1[3], e2[3];
  rot_to_euler(R1, e1); rot_to_euler(R2, e2);
  long long t6 = wall_clock64();
  sink[tid] = e1[0] + e2[1] + static_cast<float>(t[0]) + (ok ? 1.f : 0.f);
  if (tid == 0) { ticks[0] = t1 - t0; ticks[1] = t2 - t1; ticks[2] = t3 - t2; ticks[3] = t4 - t3; ticks[4] = t5 - t4; ticks[5] = t6 - t5; }
}

int main() {
  const int n = 50000;
  std::mt19937_64 rng(5);
  std::normal_distribution<double> nd;
  std::vector<double> groups(kGroups * kMom, 0.0);
  // matches of a true relative pose, so that A^T A has its small eigenvalue
  for (int i = 0; i < n; ++i) {
    double x[3] = {nd(rng), nd(rng), nd(rng)}, nn = std::sqrt(x[0] * x[0] + x[1] * x[1] + x[2] * x[2]);
    for (double& v : x) v /= nn;
    const double d = 2.0 + (rng() % 1000) / 250.0;
    double y[3] = {x[0] * d + 0.3, x[1] * d - 0.2 + 0.05 * x[2] * d, x[2] * d + 0.1 - 0.05 * x[1] * d};
    nn = std::sqrt(y[0] * y[0] + y[1] * y[1] + y[2] * y[2]);
    for (double& v : y) v = v / nn + 2e-4 * nd(rng);
    double row[9];
    for (int a = 0; a < 3; ++a) for (int b = 0; b < 3; ++b) row[3 * a + b] = x[a] * y[b];
    const int g = (i / 2) % kGroups;
    int k = 0;
    for (int a = 0; a < 9; ++a) for (int b = a; b < 9; ++b) groups[g * kMom + k++] += row[a] * row[b];
  }
  double* gd; long long* td; float* sd;
  hipMalloc(&gd, groups.size() * 8); hipMalloc(&td, 64); hipMalloc(&sd, 1024);
  hipMemcpy(gd, groups.data(), groups.size() * 8, hipMemcpyHostToDevice);
  for (int rep = 0; rep < 3; ++rep) {
    hipLaunchKernelGGL(probe, dim3(1), dim3(80), 0, 0, gd, 1ull, td, sd);
    hipDeviceSynchronize();
    long long t[6];
    hipMemcpy(t, td, sizeof(t), hipMemcpyDeviceToHost);
    std::printf("us: occupancy %.1f  groups+moments %.1f  smallest_eigvec %.1f  svd3+rank2 %.1f  decompose %.1f  euler %.1f\n", t[0] / 100.0,
                t[1] / 100.0, t[2] / 100.0, t[3] / 100.0, t[4] / 100.0, t[5] / 100.0);
  }
  return 0;
}
