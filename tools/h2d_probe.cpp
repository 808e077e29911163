// H2D probe: how fast can a PAGEABLE host array (what std::vector<cv::Point3d>::data() is) reach the device?
//   (a) hipMemcpy straight from the pageable array (what sba_problem_upload does through hipMemcpyAsync);
//   (b) the same from pinned memory (the DMA ceiling);
//   (c) a pipeline: T host threads copy chunk k+1 into one of two pinned staging buffers while the DMA engine moves chunk k.
// build: hipcc -O2 -o h2d_probe tools/h2d_probe.cpp -lpthread     run: ./h2d_probe [MiB=640] [threads=8] [chunk MiB=16]
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); std::exit(1); } } while (0)
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

static void par_copy(char* dst, const char* src, size_t n, int threads) {
  if (threads <= 1) { std::memcpy(dst, src, n); return; }
  std::vector<std::thread> pool;
  const size_t per = (n + threads - 1) / threads;
  for (int t = 0; t < threads; ++t) {
    const size_t lo = std::min(n, per * t), hi = std::min(n, per * (t + 1));
    if (hi > lo) pool.emplace_back([=] { std::memcpy(dst + lo, src + lo, hi - lo); });
  }
  for (auto& th : pool) th.join();
}

int main(int argc, char** argv) {
  const size_t mib = argc > 1 ? std::strtoull(argv[1], nullptr, 10) : 640;
  const int threads = argc > 2 ? std::atoi(argv[2]) : 8;
  const size_t chunk = (argc > 3 ? std::strtoull(argv[3], nullptr, 10) : 16) << 20;
  const size_t bytes = mib << 20;
  std::vector<char> host(bytes);
  for (size_t i = 0; i < bytes; i += 4096) host[i] = static_cast<char>(i >> 12);     // touch every page
  char* dev; CHECK(hipMalloc(reinterpret_cast<void**>(&dev), bytes));
  char* pinned; CHECK(hipHostMalloc(reinterpret_cast<void**>(&pinned), bytes, hipHostMallocDefault));
  std::memcpy(pinned, host.data(), bytes);
  hipStream_t s; CHECK(hipStreamCreate(&s));
  for (int rep = 0; rep < 3; ++rep) {
    double t0 = now();
    CHECK(hipMemcpyAsync(dev, host.data(), bytes, hipMemcpyHostToDevice, s)); CHECK(hipStreamSynchronize(s));
    double t1 = now();
    CHECK(hipMemcpyAsync(dev, pinned, bytes, hipMemcpyHostToDevice, s)); CHECK(hipStreamSynchronize(s));
    double t2 = now();
    std::printf("rep %d: pageable %.1f GB/s (%.2f ms), pinned %.1f GB/s (%.2f ms)\n", rep, bytes / (t1 - t0) / 1e9, (t1 - t0) * 1e3,
                bytes / (t2 - t1) / 1e9, (t2 - t1) * 1e3);
  }
  char* stage[2]; hipEvent_t done[2];
  for (int k = 0; k < 2; ++k) { CHECK(hipHostMalloc(reinterpret_cast<void**>(&stage[k]), chunk, hipHostMallocDefault)); CHECK(hipEventCreate(&done[k])); }
  for (int T : {1, 2, 4, threads, 2 * threads}) {
    for (int rep = 0; rep < 2; ++rep) {
      double t0 = now();
      int k = 0;
      for (size_t off = 0; off < bytes; off += chunk, k ^= 1) {
        const size_t m = std::min(chunk, bytes - off);
        if (off >= 2 * chunk) CHECK(hipEventSynchronize(done[k]));            // the DMA out of this buffer has finished
        par_copy(stage[k], host.data() + off, m, T);
        CHECK(hipMemcpyAsync(dev + off, stage[k], m, hipMemcpyHostToDevice, s));
        CHECK(hipEventRecord(done[k], s));
      }
      CHECK(hipStreamSynchronize(s));
      double t1 = now();
      if (rep == 1) std::printf("pipeline, %2d copy threads, %zu MiB chunks: %.1f GB/s (%.2f ms)\n", T, chunk >> 20, bytes / (t1 - t0) / 1e9, (t1 - t0) * 1e3);
    }
  }
  return 0;
}
