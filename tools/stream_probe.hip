// Stream probe: what does the memory system give a kernel that reads the sweep kernel's 8 coordinate/depth planes
// exactly once with the sweep kernel's access pattern and does (almost) no arithmetic?  The answer is the ceiling the
// sweep kernel is measured against in DESIGN.md section 7, next to the 8 TB/s datasheet figure.
//
//   build:  make -C spherical_bundle_adjuster_amd/csrc      (-> csrc/build/stream_probe)
//   run:    spherical_bundle_adjuster_amd/csrc/build/stream_probe [n_matches=10000000] [launches=50]   (on the GPU box)
//
// (A dynamically balanced tail -- 64-vector chunks handed out by one agent-scope atomic counter -- was measured with
// an earlier version of this probe: ~10 ns per same-address ticket, 907 us for a fully dynamic 10M launch; dropped.)
// Variants: grid size (blocks per CU), grid-stride vs per-block contiguous chunks, nt vs plain loads, 1 or 2 vectors
// in flight per plane.  Per-block start/end times (s_memrealtime, 100 MHz) of the last launch show the ramp and the tail.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <string>
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

// Mixed read / write stream (the d-only stage's pattern): NR planes read once (nt loads), NW planes written once (nt
// stores with NTS, plain stores otherwise), 16 B per lane, grid-stride, next step's loads in flight while the current
// step is stored.  Every load is consumed: the sums also go into one store per wave at the end (without it the
// read-only variants are dead code and "run" in 4 us).
struct MixedPtrs { const double2* r[12]; double2* w[4]; double* sink; };
template <int NR, int NW, bool NTS>
__global__ __launch_bounds__(kBlock) void mixed_kernel(MixedPtrs pl, size_t nvec) {
  const size_t stride = static_cast<size_t>(gridDim.x) * kBlock;
  size_t p = static_cast<size_t>(blockIdx.x) * kBlock + threadIdx.x;
  double2 cur[NR], nxt[NR];
  double acc = 0.0;
  if (p < nvec)
#pragma unroll
    for (int k = 0; k < NR; ++k) cur[k] = ld<true>(pl.r[k] + p);
  while (p < nvec) {
    const size_t pn = p + stride;
    if (pn < nvec)
#pragma unroll
      for (int k = 0; k < NR; ++k) nxt[k] = ld<true>(pl.r[k] + pn);
    double sx = 0.0, sy = 0.0;
#pragma unroll
    for (int k = 0; k < NR; ++k) { sx += cur[k].x; sy += cur[k].y; }
    acc += sx * sy;
#pragma unroll
    for (int k = 0; k < NW; ++k) {
      typedef float f4 __attribute__((ext_vector_type(4)));
      const double2 q = make_double2(sx + k, sy - k);
      if (NTS) __builtin_nontemporal_store(*reinterpret_cast<const f4*>(&q), reinterpret_cast<f4*>(pl.w[k] + p));
      else pl.w[k][p] = q;
    }
#pragma unroll
    for (int k = 0; k < NR; ++k) cur[k] = nxt[k];
    p = pn;
  }
  for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off);
  if ((threadIdx.x & 63) == 0) pl.sink[blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6)] = acc;
}

typedef void (*Kern)(PlanePtrs, size_t, double*, unsigned long long*);

struct Variant { const char* name; Kern k; };

int main(int argc, char** argv) {
  const size_t n = argc > 1 ? std::strtoull(argv[1], nullptr, 10) : 10000000ull;
  const int launches = argc > 2 ? std::atoi(argv[2]) : 50;
  const bool quick = argc > 3 && std::string(argv[3]) == "quick";   // bench.py: only the sweep kernel's pattern, one JSON line
  const bool mixed = argc > 3 && std::string(argv[3]) == "mixed";   // the d-only stage's read + write pattern
  const size_t nvec = n / 2;
  const size_t stagger = 4352;
  hipDeviceProp_t prop;
  CHECK(hipGetDeviceProperties(&prop, 0));
  const int cus = prop.multiProcessorCount;
  if (!quick)
    std::printf("# %s, %d CUs, n = %zu matches, %d planes x %.1f MB = %.1f MB per launch\n", prop.name, cus, n, kPlanes,
                n * 8 / 1e6, n * 8.0 * kPlanes / 1e6);

  if (mixed) {
    const size_t nv = n / 2;
    MixedPtrs mp;
    void* mb[16];
    for (int k = 0; k < 16; ++k) {
      CHECK(hipMalloc(&mb[k], (nv + 1) * 16 + 16 * stagger));
      CHECK(hipMemset(mb[k], 0x3c, (nv + 1) * 16 + 16 * stagger));
      if (k < 12) mp.r[k] = reinterpret_cast<const double2*>(static_cast<char*>(mb[k]) + k * stagger);
      else mp.w[k - 12] = reinterpret_cast<double2*>(static_cast<char*>(mb[k]) + k * stagger);
    }
    hipEvent_t a, b;
    CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
    CHECK(hipMalloc(reinterpret_cast<void**>(&mp.sink), static_cast<size_t>(cus) * 16 * 4 * sizeof(double)));
    struct { const char* name; void (*k)(MixedPtrs, size_t); int nr, nw; } mv[] = {
        {"10 read + 2 written nt  (the d-only pass)", mixed_kernel<10, 2, true>, 10, 2},
        {"10 read + 2 written plain", mixed_kernel<10, 2, false>, 10, 2},
        {" 8 read + 4 written nt  (pass that also stores the scaling)", mixed_kernel<8, 4, true>, 8, 4},
        {" 8 read + 4 written plain", mixed_kernel<8, 4, false>, 8, 4},
        {"12 read + 0 written", mixed_kernel<12, 0, true>, 12, 0},
        {" 8 read + 0 written (the sweep's planes)", mixed_kernel<8, 0, true>, 8, 0}};
    for (auto& v : mv)
      for (int bpc : {1, 2, 4}) {
        const int grid = cus * bpc;
        for (int i = 0; i < 5; ++i) hipLaunchKernelGGL(v.k, dim3(grid), dim3(kBlock), 0, nullptr, mp, nv);
        CHECK(hipDeviceSynchronize());
        CHECK(hipEventRecord(a, nullptr));
        for (int i = 0; i < launches; ++i) hipLaunchKernelGGL(v.k, dim3(grid), dim3(kBlock), 0, nullptr, mp, nv);
        CHECK(hipEventRecord(b, nullptr));
        CHECK(hipEventSynchronize(b));
        float ms = 0;
        CHECK(hipEventElapsedTime(&ms, a, b));
        const double us = ms * 1e3 / launches, bytes = static_cast<double>(nv) * 16 * (v.nr + v.nw);
        std::printf("mixed %-55s %d blocks/CU  %7.1f us  %6.0f GB/s (read + written)\n", v.name, bpc, us, bytes / us * 1e-3);
      }
    for (int k = 0; k < 16; ++k) CHECK(hipFree(mb[k]));
    CHECK(hipFree(mp.sink));
    return 0;
  }
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
  double quick_gbps[3] = {0, 0, 0};
  for (const Variant& v : variants)
    for (int bpc : per_cu) {
      if (quick && (&v != &variants[0] || bpc > 2)) continue;
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
      if (quick) { quick_gbps[bpc] = n * 8.0 * kPlanes / us * 1e-3; continue; }
      std::printf("%-20s %2d blocks/CU  %7.1f us  %6.0f GB/s | in-kernel span %6.1f us; start lag med %4.1f p90 %4.1f max %4.1f us;"
                  " idle-before-end med %4.1f p90 %4.1f max %4.1f us\n",
                  v.name, bpc, us, n * 8.0 * kPlanes / us * 1e-3, (last - first) * 0.01, st[grid / 2], st[grid * 9 / 10],
                  st[grid - 1], en[grid / 2], en[grid * 9 / 10], en[grid - 1]);
      if (bpc <= 2 && &v == &variants[0]) {   // is the finish-time spread systematic?  mean finish per XCD (block % 8) and per CU slot
        double sum[8] = {0}, lo[8], hi[8];
        int cnt[8] = {0};
        for (int x = 0; x < 8; ++x) { lo[x] = 1e30; hi[x] = -1e30; }
        for (int b = 0; b < grid; ++b) {
          const double t = (h[2 * b + 1] - first) * 0.01;
          sum[b & 7] += t; ++cnt[b & 7];
          lo[b & 7] = std::min(lo[b & 7], t); hi[b & 7] = std::max(hi[b & 7], t);
        }
        std::printf("    finish time by block%%8 (mean [min..max] us):");
        for (int x = 0; x < 8; ++x) std::printf("  %d: %.1f [%.1f..%.1f]", x, sum[x] / cnt[x], lo[x], hi[x]);
        std::printf("\n");
      }
      std::fflush(stdout);
    }
  if (quick) {
    std::printf("{\"n\": %zu, \"bytes_per_launch\": %.0f, \"GBps_1_block_per_cu\": %.1f, \"GBps_2_blocks_per_cu\": %.1f}\n", n,
                n * 8.0 * kPlanes, quick_gbps[1], quick_gbps[2]);
    for (int k = 0; k < kPlanes; ++k) CHECK(hipFree(base[k]));
    return 0;
  }
  // Is the slow XCD the same from launch to launch?  8 consecutive launches, each with its own clock buffer.
  {
    unsigned long long* clk8;
    const int grid = cus * 2;
    CHECK(hipMalloc(&clk8, 8 * grid * 2 * sizeof(unsigned long long)));
    for (int rep = 0; rep < 2; ++rep) {
      for (int i = 0; i < 8; ++i)
        hipLaunchKernelGGL((probe_kernel<true, false, 1>), dim3(grid), dim3(kBlock), 0, nullptr, pl, nvec, out,
                           clk8 + static_cast<size_t>(i) * grid * 2);
      CHECK(hipDeviceSynchronize());
    }
    std::vector<unsigned long long> hh(8 * grid * 2);
    CHECK(hipMemcpy(hh.data(), clk8, hh.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    std::printf("# 8 consecutive launches, 2 blocks/CU: mean finish time (us after the launch's first block start) by block%%8\n");
    for (int i = 0; i < 8; ++i) {
      const unsigned long long* c = hh.data() + static_cast<size_t>(i) * grid * 2;
      unsigned long long first = ~0ull;
      for (int b = 0; b < grid; ++b) first = std::min(first, c[2 * b]);
      double sum[8] = {0};
      for (int b = 0; b < grid; ++b) sum[b & 7] += (c[2 * b + 1] - first) * 0.01;
      std::printf("launch %d:", i);
      for (int x = 0; x < 8; ++x) std::printf(" %6.1f", sum[x] / (grid / 8));
      std::printf("\n");
    }
    CHECK(hipFree(clk8));
  }
  for (int k = 0; k < kPlanes; ++k) CHECK(hipFree(base[k]));
  return 0;
}
