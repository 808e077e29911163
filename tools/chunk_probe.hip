// Chunk probe: what does HBM give a kernel that reads SCATTERED aligned chunks of a buffer far larger than the caches --
// the access pattern of the remap's gather (csrc/sba_maps.hip: 16-byte source chunks of a 22 MB frame, hundreds of frames),
// without any of its other work?  Each wave instruction reads 64 x 16 B; the lanes of a group of G consecutive lanes read
// one contiguous run of G x 16 B at a pseudo-random, run-aligned offset.  G = 64: one 1 KiB run per instruction (a stream);
// G = 4: sixteen 64-byte runs per instruction (what a staged tile's scattered chunks look like); G = 1: 64 separate chunks.
// Reports GB/s of useful bytes (16 B per lane) per run length: the ceiling to hold `gather_tiled_kernel`'s 3.1 TB/s of
// 64-byte requests against (profiles/r03_gather_subtiles_ab.md).
//
//   build:  hipcc --offload-arch=gfx950 -O3 -o chunk_probe tools/chunk_probe.hip      run: ./chunk_probe [GiB=8] [launches=5]
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); std::exit(1); } } while (0)

__device__ __forceinline__ unsigned long long mix(unsigned long long z) {
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}

// every lane issues `per_lane` loads of 16 B, 8 in flight; run r (G lanes x 16 B) sits at a random G*16-aligned offset
template <int G>
__global__ __launch_bounds__(256) void chunk_kernel(const uint4* __restrict__ buf, unsigned long long nruns_in_buf, int per_lane,
                                                    unsigned* __restrict__ sink) {
  const unsigned long long lane_global = static_cast<unsigned long long>(blockIdx.x) * 256 + threadIdx.x;
  const unsigned long long group = lane_global / G, in_group = lane_global % G;
  unsigned acc = 0;
  for (int k = 0; k < per_lane; k += 8) {
    uint4 v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const unsigned long long run = mix(group * 0x9E3779B97F4A7C15ull + static_cast<unsigned long long>(k + j)) % nruns_in_buf;
      v[j] = buf[run * G + in_group];
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) acc += v[j].x ^ v[j].y ^ v[j].z ^ v[j].w;
  }
  if (acc == 0x12345678u) sink[0] = acc;     // keep the loads alive
}

template <int G>
void run(const uint4* buf, size_t bytes, int launches, unsigned* sink, int cus) {
  const int per_lane = 64, grid = cus * 8 * 16;
  const unsigned long long nruns = bytes / (static_cast<size_t>(G) * 16);
  hipEvent_t a, b;
  CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
  hipLaunchKernelGGL(chunk_kernel<G>, dim3(grid), dim3(256), 0, nullptr, buf, nruns, per_lane, sink);
  CHECK(hipDeviceSynchronize());
  CHECK(hipEventRecord(a, nullptr));
  for (int i = 0; i < launches; ++i) hipLaunchKernelGGL(chunk_kernel<G>, dim3(grid), dim3(256), 0, nullptr, buf, nruns, per_lane, sink);
  CHECK(hipEventRecord(b, nullptr));
  CHECK(hipEventSynchronize(b));
  float ms = 0;
  CHECK(hipEventElapsedTime(&ms, a, b));
  const double useful = static_cast<double>(grid) * 256 * per_lane * 16 * launches;
  std::printf("run length %5d B (%2d lanes): %8.1f GB/s useful, %.1f us per launch\n", G * 16, G, useful / (ms * 1e-3) / 1e9, ms * 1e3 / launches);
}

int main(int argc, char** argv) {
  const size_t gib = argc > 1 ? std::strtoull(argv[1], nullptr, 10) : 8;
  const int launches = argc > 2 ? std::atoi(argv[2]) : 5;
  const size_t bytes = gib << 30;
  hipDeviceProp_t prop;
  CHECK(hipGetDeviceProperties(&prop, 0));
  uint4* buf; unsigned* sink;
  CHECK(hipMalloc(reinterpret_cast<void**>(&buf), bytes));
  CHECK(hipMemset(buf, 0x5a, bytes));
  CHECK(hipMalloc(reinterpret_cast<void**>(&sink), 64));
  std::printf("# %s, %d CUs, %zu GiB buffer, random aligned runs, 8 loads of 16 B in flight per lane\n", prop.name, prop.multiProcessorCount, gib);
  run<64>(buf, bytes, launches, sink, prop.multiProcessorCount);
  run<16>(buf, bytes, launches, sink, prop.multiProcessorCount);
  run<8>(buf, bytes, launches, sink, prop.multiProcessorCount);
  run<4>(buf, bytes, launches, sink, prop.multiProcessorCount);
  run<2>(buf, bytes, launches, sink, prop.multiProcessorCount);
  run<1>(buf, bytes, launches, sink, prop.multiProcessorCount);
  CHECK(hipFree(buf)); CHECK(hipFree(sink));
  return 0;
}
