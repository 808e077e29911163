// What does a kernel's scratch (private segment) cost per launch?  Kernels that only differ in the size of a per-lane
// array the compiler cannot keep in registers, 256 blocks x 256 threads, launched back to back; host wall clock per
// launch + synchronise.  (ROCr keeps a device-wide scratch allocation per queue while the request stays below a limit;
// above it every dispatch allocates and releases its own.)
// Build: hipcc --offload-arch=gfx950 -O3 tools/scratch_probe.hip -o tools/build/scratch_probe
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>

template <int WORDS>
__global__ __launch_bounds__(256) void probe(const int* idx, double* out) {
  double a[WORDS];
  for (int i = 0; i < WORDS; ++i) a[i] = i * 0.5 + threadIdx.x;
  double s = 0;
  for (int i = 0; i < 4; ++i) s += a[idx[i] % WORDS];      // dynamic index: the array lives in scratch
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int WORDS>
void run(const int* idx, double* out, int blocks) {
  for (int i = 0; i < 3; ++i) { hipLaunchKernelGGL(probe<WORDS>, dim3(blocks), dim3(256), 0, 0, idx, out); hipDeviceSynchronize(); }
  const int reps = 20;
  const auto t0 = std::chrono::steady_clock::now();
  for (int i = 0; i < reps; ++i) { hipLaunchKernelGGL(probe<WORDS>, dim3(blocks), dim3(256), 0, 0, idx, out); hipDeviceSynchronize(); }
  const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / reps;
  std::printf("scratch %5d B/lane, %4d blocks: %8.1f us per launch + sync\n", WORDS * 8, blocks, us);
}

int main() {
  int* idx; double* out;
  hipMalloc(&idx, 16); hipMalloc(&out, 512 * 256 * 8);
  const int h[4] = {1, 5, 2, 7};
  hipMemcpy(idx, h, 16, hipMemcpyHostToDevice);
  for (int blocks : {1, 256}) {
    run<8>(idx, out, blocks); run<16>(idx, out, blocks); run<32>(idx, out, blocks); run<40>(idx, out, blocks); run<48>(idx, out, blocks);
    run<64>(idx, out, blocks); run<96>(idx, out, blocks); run<128>(idx, out, blocks); run<256>(idx, out, blocks);
  }
  // and mixed: does a small-scratch kernel after a large one pay again?
  run<16>(idx, out, 256); run<256>(idx, out, 256); run<16>(idx, out, 256);
  return 0;
}
