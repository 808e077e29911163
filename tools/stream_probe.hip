// Stream probe: what does the memory system give a kernel that reads the sweep kernel's 8 coordinate/depth planes
// exactly once with the sweep kernel's access pattern and does (almost) no arithmetic?  The answer is the ceiling the
// sweep kernel is measured against in DESIGN.md section 7, next to the 8 TB/s datasheet figure.
//
//   build:  hipcc --offload-arch=gfx950 -O3 -o build/stream_probe tools/stream_probe.hip
//   run:    build/stream_probe [n_matches=10000000] [launches=50]
//
// Variants: grid size (blocks per CU), grid-stride vs per-block contiguous chunks, nt vs plain loads, 1 or 2 vectors
// in flight per plane.  Per-block start/end times (s_memrealtime, 100 MHz) of the last launch show the ramp and the tail.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x)                                                                                       \
  do {                                                                                                 \
    hipError_t e_ = (x);                                                                               \
    if (e_ != hipSuccess) {                                                                            \
      std::fprintf(stderr, "%s:%d %s: %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e_));          \
      std::exit(1);                                                                                    \
    }                                                                                                  \
  } while (0)

constexpr int kPlanes = 8;
constexpr int kBlock = 256;
struct PlanePtrs { const double2* p[kPlanes]; };

template <bool NT>
__device__ __forceinline__ double2 ld(const double2* ptr) {
  if (NT) {
    typedef float f4 __attribute__((ext_vector_type(4)));
    const f4 r = __builtin_nontemporal_load(reinterpret_cast<const f4*>(ptr));
    return *reinterpret_cast<const double2*>(&r);
  }
  return *ptr;
}

// CHUNKED = false: grid-stride (block b reads vectors b*256+tid + k*grid*256) -- the sweep kernel's pattern.
// CHUNKED = true : block b owns one contiguous range of every plane.
template <bool NT, bool CHUNKED, int DEPTH>
__global__ __launch_bounds__(kBlock) void probe_kernel(PlanePtrs pl, size_t nvec, double* __restrict__ out,
                                                        unsigned long long* __restrict__ clocks) {
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
  size_t p, end, stride;
  if (CHUNKED) {
    const size_t per = (nvec + gridDim.x - 1) / gridDim.x;
    p = per * blockIdx.x + threadIdx.x;
    end = std::min(nvec, per * (blockIdx.x + 1));
    stride = kBlock;
  } else {
    p = static_cast<size_t>(blockIdx.x) * kBlock + threadIdx.x;
    end = nvec;
    stride = static_cast<size_t>(gridDim.x) * kBlock;
  }
  double acc = 0.0;
  double2 cur[DEPTH][kPlanes];
#pragma unroll
  for (int d = 0; d < DEPTH; ++d)
    if (p + d * stride < end)
#pragma unroll
      for (int k = 0; k < kPlanes; ++k) cur[d][k] = ld<NT>(pl.p[k] + p + d * stride);
  while (p < end) {
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) {
      if (p < end) {
        double s = 0.0;
#pragma unroll
        for (int k = 0; k < kPlanes; ++k) s += cur[d][k].x * cur[d][k].y;
        acc += s;
        const size_t pn = p + DEPTH * stride;
        if (pn < end)
#pragma unroll
          for (int k = 0; k < kPlanes; ++k) cur[d][k] = ld<NT>(pl.p[k] + pn);
      }
      p += stride;
    }
  }
  // keep the result alive: one store per wave
  for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off);
  if ((threadIdx.x & 63) == 0) out[blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6)] = acc;
  if (threadIdx.x == 0) {
    clocks[2 * blockIdx.x] = t0;
    clocks[2 * blockIdx.x + 1] = __builtin_amdgcn_s_memrealtime();
  }
}

// Static grid-stride over the first `static_vec` vectors, then wave-granular dynamic chunks (64 vectors = 1 KiB per
// plane) handed out by one atomic counter: waves that finish their static share early take more of the rest.  The
// ticket for the next chunk is requested before the current chunk's loads, so its latency overlaps theirs.
template <bool NT>
__global__ __launch_bounds__(kBlock) void probe_dyn_kernel(PlanePtrs pl, size_t nvec, size_t static_vec,
                                                            unsigned* __restrict__ counter, double* __restrict__ out,
                                                            unsigned long long* __restrict__ clocks) {
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
  const int lane = threadIdx.x & 63;
  const unsigned nd = static_cast<unsigned>((nvec - static_vec + 63) / 64);
  unsigned ticket = 0;
  if (lane == 0) ticket = atomicAdd(counter, 1u);        // first dynamic chunk: latency hidden under the static part
  double acc = 0.0;
  {
    size_t p = static_cast<size_t>(blockIdx.x) * kBlock + threadIdx.x;
    const size_t stride = static_cast<size_t>(gridDim.x) * kBlock;
    double2 cur[kPlanes];
    if (p < static_vec)
#pragma unroll
      for (int k = 0; k < kPlanes; ++k) cur[k] = ld<NT>(pl.p[k] + p);
    while (p < static_vec) {
      double s = 0.0;
#pragma unroll
      for (int k = 0; k < kPlanes; ++k) s += cur[k].x * cur[k].y;
      acc += s;
      p += stride;
      if (p < static_vec)
#pragma unroll
        for (int k = 0; k < kPlanes; ++k) cur[k] = ld<NT>(pl.p[k] + p);
    }
  }
  unsigned c = __builtin_amdgcn_readfirstlane(ticket);
  while (c < nd) {
    unsigned nxt = 0;
    if (lane == 0) nxt = atomicAdd(counter, 1u);
    const size_t p = static_vec + static_cast<size_t>(c) * 64 + lane;
    if (p < nvec) {
      double2 v[kPlanes];
#pragma unroll
      for (int k = 0; k < kPlanes; ++k) v[k] = ld<NT>(pl.p[k] + p);
      double s = 0.0;
#pragma unroll
      for (int k = 0; k < kPlanes; ++k) s += v[k].x * v[k].y;
      acc += s;
    }
    c = __builtin_amdgcn_readfirstlane(nxt);
  }
  for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off);
  if ((threadIdx.x & 63) == 0) out[blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6)] = acc;
  __syncthreads();
  if (threadIdx.x == 0) {
    clocks[2 * blockIdx.x] = t0;
    clocks[2 * blockIdx.x + 1] = __builtin_amdgcn_s_memrealtime();
  }
}

typedef void (*Kern)(PlanePtrs, size_t, double*, unsigned long long*);

struct Variant { const char* name; Kern k; };

int main(int argc, char** argv) {
  const size_t n = argc > 1 ? std::strtoull(argv[1], nullptr, 10) : 10000000ull;
  const int launches = argc > 2 ? std::atoi(argv[2]) : 50;
  const size_t nvec = n / 2;
  const size_t stagger = 4352;
  hipDeviceProp_t prop;
  CHECK(hipGetDeviceProperties(&prop, 0));
  const int cus = prop.multiProcessorCount;
  std::printf("# %s, %d CUs, n = %zu matches, %d planes x %.1f MB = %.1f MB per launch\n", prop.name, cus, n, kPlanes,
              n * 8 / 1e6, n * 8.0 * kPlanes / 1e6);

  PlanePtrs pl;
  void* base[kPlanes];
  for (int k = 0; k < kPlanes; ++k) {
    CHECK(hipMalloc(&base[k], (nvec + 1) * 16 + 8 * stagger));
    CHECK(hipMemset(base[k], 0x3c, (nvec + 1) * 16 + 8 * stagger));
    pl.p[k] = reinterpret_cast<const double2*>(static_cast<char*>(base[k]) + k * stagger);
  }
  const int max_grid = cus * 16;
  double* out;
  unsigned long long* clocks;
  CHECK(hipMalloc(&out, max_grid * 4 * sizeof(double)));
  CHECK(hipMalloc(&clocks, max_grid * 2 * sizeof(unsigned long long)));
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));

  const Variant variants[] = {
      {"stride nt   1-deep", probe_kernel<true, false, 1>},  {"stride nt   2-deep", probe_kernel<true, false, 2>},
      {"stride plain 1-deep", probe_kernel<false, false, 1>}, {"chunk  nt   1-deep", probe_kernel<true, true, 1>},
      {"chunk  nt   2-deep", probe_kernel<true, true, 2>},
  };
  const int per_cu[] = {1, 2, 3, 4, 8};
  std::vector<unsigned long long> h;
  for (const Variant& v : variants)
    for (int bpc : per_cu) {
      const int grid = cus * bpc;
      for (int i = 0; i < 5; ++i) hipLaunchKernelGGL(v.k, dim3(grid), dim3(kBlock), 0, nullptr, pl, nvec, out, clocks);
      CHECK(hipDeviceSynchronize());
      CHECK(hipEventRecord(e0, nullptr));
      for (int i = 0; i < launches; ++i)
        hipLaunchKernelGGL(v.k, dim3(grid), dim3(kBlock), 0, nullptr, pl, nvec, out, clocks);
      CHECK(hipEventRecord(e1, nullptr));
      CHECK(hipEventSynchronize(e1));
      CHECK(hipGetLastError());
      float ms = 0;
      CHECK(hipEventElapsedTime(&ms, e0, e1));
      const double us = ms * 1e3 / launches;
      h.resize(2 * grid);
      CHECK(hipMemcpy(h.data(), clocks, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
      unsigned long long first = ~0ull, last = 0;
      for (int b = 0; b < grid; ++b) { first = std::min(first, h[2 * b]); last = std::max(last, h[2 * b + 1]); }
      std::vector<double> st(grid), en(grid);
      for (int b = 0; b < grid; ++b) { st[b] = (h[2 * b] - first) * 0.01; en[b] = (last - h[2 * b + 1]) * 0.01; }
      std::sort(st.begin(), st.end());
      std::sort(en.begin(), en.end());
      std::printf("%-20s %2d blocks/CU  %7.1f us  %6.0f GB/s | in-kernel span %6.1f us; start lag med %4.1f p90 %4.1f max %4.1f us;"
                  " idle-before-end med %4.1f p90 %4.1f max %4.1f us\n",
                  v.name, bpc, us, n * 8.0 * kPlanes / us * 1e-3, (last - first) * 0.01, st[grid / 2], st[grid * 9 / 10],
                  st[grid - 1], en[grid / 2], en[grid * 9 / 10], en[grid - 1]);
      std::fflush(stdout);
    }
  // dynamic tail
  unsigned* counters;
  const int ncounters = launches + 8;
  CHECK(hipMalloc(&counters, ncounters * 64));
  const double fracs[] = {0.95, 0.9, 0.85, 0.75, 0.5, 0.0};
  for (int bpc : {1, 2})
    for (double f : fracs) {
      const int grid = cus * bpc;
      const size_t round = static_cast<size_t>(grid) * kBlock;
      const size_t static_vec = static_cast<size_t>(nvec * f) / round * round;
      CHECK(hipMemset(counters, 0, ncounters * 64));
      for (int i = 0; i < 5; ++i)
        hipLaunchKernelGGL(probe_dyn_kernel<true>, dim3(grid), dim3(kBlock), 0, nullptr, pl, nvec, static_vec,
                           counters + 16 * (launches + i % 8), out, clocks);
      CHECK(hipDeviceSynchronize());
      CHECK(hipMemset(counters, 0, ncounters * 64));
      CHECK(hipDeviceSynchronize());
      CHECK(hipEventRecord(e0, nullptr));
      for (int i = 0; i < launches; ++i)
        hipLaunchKernelGGL(probe_dyn_kernel<true>, dim3(grid), dim3(kBlock), 0, nullptr, pl, nvec, static_vec,
                           counters + 16 * i, out, clocks);
      CHECK(hipEventRecord(e1, nullptr));
      CHECK(hipEventSynchronize(e1));
      CHECK(hipGetLastError());
      float ms = 0;
      CHECK(hipEventElapsedTime(&ms, e0, e1));
      const double us = ms * 1e3 / launches;
      h.resize(2 * grid);
      CHECK(hipMemcpy(h.data(), clocks, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
      unsigned long long first = ~0ull, last = 0;
      for (int b = 0; b < grid; ++b) { first = std::min(first, h[2 * b]); last = std::max(last, h[2 * b + 1]); }
      std::vector<double> en(grid);
      for (int b = 0; b < grid; ++b) en[b] = (last - h[2 * b + 1]) * 0.01;
      std::sort(en.begin(), en.end());
      std::printf("dyn tail static=%.2f   %2d blocks/CU  %7.1f us  %6.0f GB/s | in-kernel span %6.1f us; idle-before-end med %4.1f"
                  " p90 %4.1f max %4.1f us\n",
                  f, bpc, us, n * 8.0 * kPlanes / us * 1e-3, (last - first) * 0.01, en[grid / 2], en[grid * 9 / 10],
                  en[grid - 1]);
      std::fflush(stdout);
    }
  for (int k = 0; k < kPlanes; ++k) CHECK(hipFree(base[k]));
  return 0;
}
