// Measurement tool (not part of the library): how should the 3 source bytes of a gathered 8UC3 pixel be loaded?
// Same access pattern as gather_kernel (csrc/sba_maps.hip): equi2cube table of a 3840x1920 -> S=600 strip, F frames.
//   variant 0: three byte loads            variant 1: one unaligned dword load
//   variant 2: two aligned dword loads + v_alignbyte     variant 3: as 1 but one frame at a time
// usage: gather_probe [frames=64] [iters=5]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <algorithm>
#include <cstring>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ void table_kernel(int S, int H, int W, int* table) {
  const size_t o = static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (o >= static_cast<size_t>(S) * 6 * S) return;
  const int i = o / (6 * S), c = o % (6 * S), face = c / S, j = c - face * S;
  const double s = S, a = (s - 2.0 * j) / s, b = (s - 2.0 * i) / s, an = -a, bn = -b;
  double x, y, z;
  switch (face) { case 0: x = a; y = 1; z = b; break; case 1: x = -1; y = a; z = b; break; case 2: x = an; y = -1; z = b; break;
                  case 3: x = 1; y = an; z = b; break; case 4: x = b; y = a; z = 1; break; default: x = bn; y = a; z = -1; }
  const double nrm = sqrt(x * x + y * y + z * z), kPi = 3.14159265358979323846;
  const double th = acos(z / nrm); double ph = atan2(y / nrm, x / nrm); if (ph < 0) ph += 2 * kPi;
  int row = min(max(int(H * th / kPi), 0), H - 1), col = min(max(int(W * ph / (2 * kPi)), 0), W - 1);
  table[o] = row * W + col;
}

template <int V> __device__ __forceinline__ uint32_t load_px(const uint8_t* __restrict__ src, int p, int last) {
  const size_t o = static_cast<size_t>(p) * 3;
  if (V == 0) return src[o] | (uint32_t(src[o + 1]) << 8) | (uint32_t(src[o + 2]) << 16);
  if (V == 1 || V == 3) {
    uint32_t v;
    if (p == last) { __builtin_memcpy(&v, src + o - 1, 4); return v >> 8; }
    __builtin_memcpy(&v, src + o, 4);
    return v & 0xffffffu;
  }
  // two aligned dwords, funnel-shifted: bytes [o, o+3) live in dwords (o & ~3) and +4
  const size_t a = o & ~size_t(3);
  const uint32_t lo = *reinterpret_cast<const uint32_t*>(src + a);
  const uint32_t hi = (p == last) ? 0u : *reinterpret_cast<const uint32_t*>(src + a + 4);
  return __builtin_amdgcn_alignbyte(hi, lo, static_cast<uint32_t>(o & 3)) & 0xffffffu;
}

template <int V>
__global__ __launch_bounds__(256) void gather(const int* __restrict__ table, size_t out_pixels, const uint8_t* __restrict__ src,
                                              size_t src_stride, int src_pixels, uint8_t* __restrict__ out, size_t out_stride,
                                              int batch, int fpb) {
  const size_t g = static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (g * 4 >= out_pixels) return;
  const int4 q = reinterpret_cast<const int4*>(table)[g];
  const int p[4] = {q.x, q.y, q.z, q.w};
  const int last = src_pixels - 1;
  const size_t o = g * 12;
  const int f0 = blockIdx.y * fpb, f1 = min(batch, f0 + fpb);
  auto emit = [&](uint8_t* dst, const uint32_t* v) {
    uint32_t* o32 = reinterpret_cast<uint32_t*>(dst + o);
    o32[0] = v[0] | (v[1] << 24); o32[1] = (v[1] >> 8) | (v[2] << 16); o32[2] = (v[2] >> 16) | (v[3] << 8);
  };
  int f = f0;
  if (V != 3)
    for (; f + 1 < f1; f += 2) {
      const uint8_t* sa = src + static_cast<size_t>(f) * src_stride; const uint8_t* sb = sa + src_stride;
      uint32_t va[4], vb[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) { va[k] = load_px<V>(sa, p[k], last); vb[k] = load_px<V>(sb, p[k], last); }
      emit(out + static_cast<size_t>(f) * out_stride, va); emit(out + static_cast<size_t>(f + 1) * out_stride, vb);
    }
  for (; f < f1; ++f) {
    const uint8_t* sa = src + static_cast<size_t>(f) * src_stride;
    uint32_t va[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) va[k] = load_px<V>(sa, p[k], last);
    emit(out + static_cast<size_t>(f) * out_stride, va);
  }
}

// variant 4: one output pixel per lane (a wave instruction spans 64 consecutive output pixels: few source lines), byte stores
__global__ __launch_bounds__(256) void gather_pix1(const int* __restrict__ table, size_t out_pixels, const uint8_t* __restrict__ src,
                                                   size_t src_stride, uint8_t* __restrict__ out, size_t out_stride, int batch, int fpb) {
  const size_t g = static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (g >= out_pixels) return;
  const size_t p = static_cast<size_t>(table[g]) * 3, o = g * 3;
  const int f0 = blockIdx.y * fpb, f1 = min(batch, f0 + fpb);
  for (int f = f0; f < f1; ++f) {
    const uint8_t* sa = src + static_cast<size_t>(f) * src_stride;
    uint8_t* d = out + static_cast<size_t>(f) * out_stride + o;
    const uint8_t b0 = sa[p], b1 = sa[p + 1], b2 = sa[p + 2];
    d[0] = b0; d[1] = b1; d[2] = b2;
  }
}
// variants 6..8: one pixel per lane; 6 = two frames in flight; 7 = the four lanes of a quad pack their 12 bytes into three
// dword stores (DPP quad shift); 8 = both
template <bool TWO, bool PACK>
__global__ __launch_bounds__(256) void gather_pix1x(const int* __restrict__ table, size_t out_pixels, const uint8_t* __restrict__ src,
                                                    size_t src_stride, uint8_t* __restrict__ out, size_t out_stride, int batch, int fpb) {
  const size_t g = static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  const bool valid = g < out_pixels;
  const size_t p = valid ? static_cast<size_t>(table[g]) * 3 : 0, o = g * 3;
  const int j = threadIdx.x & 3;
  const int f0 = blockIdx.y * fpb, f1 = min(batch, f0 + fpb);
  auto load = [&](const uint8_t* sa) { return uint32_t(sa[p]) | (uint32_t(sa[p + 1]) << 8) | (uint32_t(sa[p + 2]) << 16); };
  auto store = [&](uint8_t* base, uint32_t v) {
    if (PACK) {
      // lanes 0..2 of each quad store one dword: bytes of pixels j and j + 1
      const uint32_t nx = __builtin_amdgcn_update_dpp(0u, v, 0xF9 /* quad_perm [1,2,3,3] */, 0xf, 0xf, false);
      const uint32_t d = (v >> (8 * j)) | (nx << (24 - 8 * j));
      if (valid && j < 3) reinterpret_cast<uint32_t*>(base + (g & ~size_t(3)) * 3)[j] = d;   // out_pixels % 4 == 0
    } else if (valid) {
      base[o] = uint8_t(v); base[o + 1] = uint8_t(v >> 8); base[o + 2] = uint8_t(v >> 16);
    }
  };
  int f = f0;
  if (TWO)
    for (; f + 1 < f1; f += 2) {
      const uint8_t* sa = src + static_cast<size_t>(f) * src_stride;
      const uint32_t va = load(sa), vb = load(sa + src_stride);
      store(out + static_cast<size_t>(f) * out_stride, va); store(out + static_cast<size_t>(f + 1) * out_stride, vb);
    }
  for (; f < f1; ++f) store(out + static_cast<size_t>(f) * out_stride, load(src + static_cast<size_t>(f) * src_stride));
}
// variant 9: entries sorted by SOURCE index (lanes of a wave read neighbouring source pixels: few lines per load
// instruction even on the polar faces), scattered 3-byte stores to out[dst[k]]
__global__ __launch_bounds__(256) void gather_sorted(const int2* __restrict__ pairs, size_t out_pixels, const uint8_t* __restrict__ src,
                                                     size_t src_stride, uint8_t* __restrict__ out, size_t out_stride, int batch, int fpb) {
  const size_t g = static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (g >= out_pixels) return;
  const int2 e = pairs[g];
  const size_t p = static_cast<size_t>(e.x) * 3, o = static_cast<size_t>(e.y) * 3;
  const int f0 = blockIdx.y * fpb, f1 = min(batch, f0 + fpb);
  for (int f = f0; f < f1; ++f) {
    const uint8_t* sa = src + static_cast<size_t>(f) * src_stride;
    uint8_t* d = out + static_cast<size_t>(f) * out_stride + o;
    const uint8_t b0 = sa[p], b1 = sa[p + 1], b2 = sa[p + 2];
    d[0] = b0; d[1] = b1; d[2] = b2;
  }
}
// variant 5: a wave owns 256 consecutive output pixels; round k gathers pixels 64 k + lane (narrow span per instruction),
// the 768 bytes are transposed through LDS and stored as one dwordx3 per lane
__global__ __launch_bounds__(256) void gather_lds(const int* __restrict__ table, size_t out_pixels, const uint8_t* __restrict__ src,
                                                  size_t src_stride, uint8_t* __restrict__ out, size_t out_stride, int batch, int fpb) {
  __shared__ uint8_t stage[4][2][768 + 16];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const size_t wbase = (static_cast<size_t>(blockIdx.x) * 4 + wave) * 256;      // first output pixel of this wave
  if (wbase >= out_pixels) return;
  size_t p[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) p[k] = wbase + 64 * k + lane < out_pixels ? static_cast<size_t>(table[wbase + 64 * k + lane]) * 3 : 0;
  const size_t valid_bytes = (min(out_pixels - wbase, static_cast<size_t>(256))) * 3;
  const int f0 = blockIdx.y * fpb, f1 = min(batch, f0 + fpb);
  for (int f = f0; f < f1; ++f) {
    const uint8_t* sa = src + static_cast<size_t>(f) * src_stride;
    uint8_t* st = stage[wave][f & 1];
    uint8_t v[12];
#pragma unroll
    for (int k = 0; k < 4; ++k) { v[3 * k] = sa[p[k]]; v[3 * k + 1] = sa[p[k] + 1]; v[3 * k + 2] = sa[p[k] + 2]; }
#pragma unroll
    for (int k = 0; k < 4; ++k) { st[(64 * k + lane) * 3] = v[3 * k]; st[(64 * k + lane) * 3 + 1] = v[3 * k + 1]; st[(64 * k + lane) * 3 + 2] = v[3 * k + 2]; }
    __builtin_amdgcn_wave_barrier();
    const uint32_t* s32 = reinterpret_cast<const uint32_t*>(st) + lane * 3;
    const uint32_t a = s32[0], b = s32[1], c = s32[2];
    uint32_t* o32 = reinterpret_cast<uint32_t*>(out + static_cast<size_t>(f) * out_stride + wbase * 3) + lane * 3;
    if (static_cast<size_t>(lane) * 12 + 12 <= valid_bytes) { o32[0] = a; o32[1] = b; o32[2] = c; }
  }
}

int main(int argc, char** argv) {
  const int F = argc > 1 ? std::atoi(argv[1]) : 64, iters = argc > 2 ? std::atoi(argv[2]) : 5;
  const int H = 1920, W = 3840, S = 600;
  const size_t outpx = size_t(S) * 6 * S, srcb = size_t(H) * W * 3, outb = outpx * 3;
  int* table; uint8_t *src, *out, *ref;
  CK(hipMalloc(&table, outpx * 4)); CK(hipMalloc(&src, srcb * F + 16)); CK(hipMalloc(&out, outb * F)); CK(hipMalloc(&ref, outb * F));
  std::vector<uint8_t> h(srcb);
  for (size_t i = 0; i < srcb; ++i) h[i] = uint8_t((i * 2654435761u) >> 13);
  for (int f = 0; f < F; ++f) { h[0] = uint8_t(f); CK(hipMemcpy(src + srcb * f, h.data(), srcb, hipMemcpyHostToDevice)); }
  hipLaunchKernelGGL(table_kernel, dim3((outpx + 255) / 256), dim3(256), 0, 0, S, H, W, table);
  CK(hipDeviceSynchronize());
  // sorted (source, destination) pairs for variant 9
  int2* pairs;
  {
    std::vector<int> ht(outpx);
    CK(hipMemcpy(ht.data(), table, outpx * 4, hipMemcpyDeviceToHost));
    std::vector<int2> hp(outpx);
    for (size_t i = 0; i < outpx; ++i) hp[i] = make_int2(ht[i], static_cast<int>(i));
    std::sort(hp.begin(), hp.end(), [](const int2& a, const int2& b) { return a.x != b.x ? a.x < b.x : a.y < b.y; });
    CK(hipMalloc(&pairs, outpx * sizeof(int2)));
    CK(hipMemcpy(pairs, hp.data(), outpx * sizeof(int2), hipMemcpyHostToDevice));
  }
  const unsigned gx = (outpx / 4 + 255) / 256;
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int fpb : {1, 2, 4, 8}) {
    if (fpb > F) break;
    const unsigned gy = (F + fpb - 1) / fpb;
    for (int v = 7; v < 10; ++v) {
      auto launch = [&](uint8_t* dst) {
        switch (v) {
          case 0: hipLaunchKernelGGL(gather<0>, dim3(gx, gy), dim3(256), 0, 0, table, outpx, src, srcb, H * W, dst, outb, F, fpb); break;
          case 1: hipLaunchKernelGGL(gather<1>, dim3(gx, gy), dim3(256), 0, 0, table, outpx, src, srcb, H * W, dst, outb, F, fpb); break;
          case 2: hipLaunchKernelGGL(gather<2>, dim3(gx, gy), dim3(256), 0, 0, table, outpx, src, srcb, H * W, dst, outb, F, fpb); break;
          case 3: hipLaunchKernelGGL(gather<3>, dim3(gx, gy), dim3(256), 0, 0, table, outpx, src, srcb, H * W, dst, outb, F, fpb); break;
          case 4: hipLaunchKernelGGL(gather_pix1, dim3((outpx + 255) / 256, gy), dim3(256), 0, 0, table, outpx, src, srcb, dst, outb, F, fpb); break;
          case 6: hipLaunchKernelGGL((gather_pix1x<true, false>), dim3((outpx + 255) / 256, gy), dim3(256), 0, 0, table, outpx, src, srcb, dst, outb, F, fpb); break;
          case 7: hipLaunchKernelGGL((gather_pix1x<false, true>), dim3((outpx + 255) / 256, gy), dim3(256), 0, 0, table, outpx, src, srcb, dst, outb, F, fpb); break;
          case 9: hipLaunchKernelGGL(gather_sorted, dim3((outpx + 255) / 256, gy), dim3(256), 0, 0, pairs, outpx, src, srcb, dst, outb, F, fpb); break;
          case 8: hipLaunchKernelGGL((gather_pix1x<true, true>), dim3((outpx + 255) / 256, gy), dim3(256), 0, 0, table, outpx, src, srcb, dst, outb, F, fpb); break;
          default: hipLaunchKernelGGL(gather_lds, dim3((outpx / 256 + 3) / 4, gy), dim3(256), 0, 0, table, outpx, src, srcb, dst, outb, F, fpb);
        }
      };
      if (v == 7 && fpb == 1) hipLaunchKernelGGL(gather<0>, dim3(gx, gy), dim3(256), 0, 0, table, outpx, src, srcb, H * W, ref, outb, F, fpb);
      launch(out);
      CK(hipDeviceSynchronize());
      CK(hipEventRecord(e0, 0));
      for (int it = 0; it < iters; ++it) launch(out);
      CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= iters;
      std::vector<uint8_t> a(outb), b(outb);
      CK(hipMemcpy(a.data(), out + outb * (F - 1), outb, hipMemcpyDeviceToHost)); CK(hipMemcpy(b.data(), ref + outb * (F - 1), outb, hipMemcpyDeviceToHost));
      std::printf("fpb %2d variant %d: %.3f ms / %d frames = %.0f frames/s, %.0f GB/s algorithmic, %s\n", fpb, v, ms, F, F / (ms * 1e-3),
                  double(F) * outpx * 6 / (ms * 1e-3) / 1e9, std::memcmp(a.data(), b.data(), outb) == 0 ? "same" : "DIFFERENT");
    }
  }
  return 0;
}
