// Device pass of the 8-point initial guess (reference spherical_bundle_adjuster.cpp:53-68: one row
// kron(left_i, right_i) of the N x 9 matrix A per match).  Instead of materialising A for 80 random subsets and
// running 80 SVDs of (N/4) x 9 matrices, one sweep accumulates A^T A (45 sums) separately for 64 interleaved
// groups of matches, group(i) = (i / 4) % 64: a lane always handles 4 consecutive matches and keeps its own 45
// accumulators, so lane id == group id and no cross-lane reduction is needed at all.  One-wave blocks; the rows of
// all blocks are folded per (group, entry) by epipolar_fold_kernel in a fixed order.  Reads the same planes as the
// sweep kernel (48 B per match), once per problem -- not on the per-iteration path.
#include "sba_device.hpp"

namespace sba {
namespace {

constexpr int kMom = 45;

__device__ __forceinline__ void add_match(double x, double y, double z, double u, double v, double w,
                                          double* __restrict__ acc) {
  // row of A: left (x) right, .cpp:59-67
  const double a[9] = {x * u, x * v, x * w, y * u, y * v, y * w, z * u, z * v, z * w};
  int k = 0;
#pragma unroll
  for (int i = 0; i < 9; ++i)
#pragma unroll
    for (int j = i; j < 9; ++j) {
      acc[k] = __builtin_fma(a[i], a[j], acc[k]);
      ++k;
    }
}

template <typename ST>
__device__ __forceinline__ void load4(const void* plane, size_t quad, double out[4]);
template <>
__device__ __forceinline__ void load4<double>(const void* plane, size_t quad, double out[4]) {
  typedef float f4 __attribute__((ext_vector_type(4)));
  const f4 ra = __builtin_nontemporal_load(reinterpret_cast<const f4*>(plane) + 2 * quad);
  const f4 rb = __builtin_nontemporal_load(reinterpret_cast<const f4*>(plane) + 2 * quad + 1);
  const double2 a = *reinterpret_cast<const double2*>(&ra), b = *reinterpret_cast<const double2*>(&rb);
  out[0] = a.x; out[1] = a.y; out[2] = b.x; out[3] = b.y;
}
template <>
__device__ __forceinline__ void load4<float>(const void* plane, size_t quad, double out[4]) {
  const float4 a = reinterpret_cast<const float4*>(plane)[quad];
  out[0] = a.x; out[1] = a.y; out[2] = a.z; out[3] = a.w;
}

// partials[block][entry][lane]  (entry-major so that the 64 lanes store 512 contiguous bytes per entry)
template <typename ST>
__global__ __launch_bounds__(64) void epipolar_moments_kernel(Planes pl, unsigned long long n,
                                                              double* __restrict__ partials) {
  double acc[kMom];
#pragma unroll
  for (int k = 0; k < kMom; ++k) acc[k] = 0.0;
  const size_t nquad = (n + 3) / 4;
  const size_t stride = static_cast<size_t>(gridDim.x) * 64;
  // next quad's loads are issued before the current one is consumed (register double buffer, like the sweep kernel)
  size_t q = static_cast<size_t>(blockIdx.x) * 64 + threadIdx.x;
  double cur[6][4], nxt[6][4];
  if (q < nquad) {
#pragma unroll
    for (int k = 0; k < 3; ++k) { load4<ST>(pl.x1[k], q, cur[k]); load4<ST>(pl.x2[k], q, cur[3 + k]); }
  }
  while (q < nquad) {
    const size_t qn = q + stride;
    if (qn < nquad) {
#pragma unroll
      for (int k = 0; k < 3; ++k) { load4<ST>(pl.x1[k], qn, nxt[k]); load4<ST>(pl.x2[k], qn, nxt[3 + k]); }
    }
#pragma unroll
    for (int h = 0; h < 4; ++h)
      if (4 * q + h < n)   // the planes are zero-padded, so this only guards the count-exactness of a ragged tail
        add_match(cur[0][h], cur[1][h], cur[2][h], cur[3][h], cur[4][h], cur[5][h], acc);
#pragma unroll
    for (int k = 0; k < 6; ++k)
#pragma unroll
      for (int h = 0; h < 4; ++h) cur[k][h] = nxt[k][h];
    q = qn;
  }
  double* row = partials + static_cast<size_t>(blockIdx.x) * kMom * 64;
#pragma unroll
  for (int k = 0; k < kMom; ++k) row[k * 64 + threadIdx.x] = acc[k];
}

// groups[lane][entry] = sum over blocks of partials[block][entry][lane].  One 1024-thread block per entry: wave s
// folds blocks s, s+16, s+32, ... (four independent accumulators), wave 0 then adds the 16 segment sums in segment
// order -- a fixed order for a given grid, and 16 x 4 loads in flight per (entry, lane) instead of 4.
__global__ __launch_bounds__(1024) void epipolar_fold_kernel(const double* __restrict__ partials, int nblocks,
                                                             double* __restrict__ groups) {
  __shared__ double seg_sum[16][64];
  const int entry = blockIdx.x, lane = threadIdx.x & 63, seg = threadIdx.x >> 6;
  double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
  const size_t bs = static_cast<size_t>(kMom) * 64;
  const double* src = partials + static_cast<size_t>(entry) * 64 + lane;
  int b = seg;
  for (; b + 48 < nblocks; b += 64) {
    s0 += src[static_cast<size_t>(b) * bs];
    s1 += src[static_cast<size_t>(b + 16) * bs];
    s2 += src[static_cast<size_t>(b + 32) * bs];
    s3 += src[static_cast<size_t>(b + 48) * bs];
  }
  for (; b < nblocks; b += 16) s0 += src[static_cast<size_t>(b) * bs];
  seg_sum[seg][lane] = (s0 + s1) + (s2 + s3);
  __syncthreads();
  if (seg == 0) {
    double tot = seg_sum[0][lane];
#pragma unroll
    for (int k = 1; k < 16; ++k) tot += seg_sum[k][lane];
    groups[lane * kMom + entry] = tot;
  }
}

}  // namespace

// groups_dev: [64][45] doubles; partials: [grid][45][64] doubles scratch.
hipError_t launch_epipolar_moments(int store, const Planes& pl, size_t n, double* partials, int grid,
                                   double* groups_dev, hipStream_t stream) {
  if (grid > 0) {
    if (store == 0)
      hipLaunchKernelGGL((epipolar_moments_kernel<double>), dim3(grid), dim3(64), 0, stream, pl,
                         static_cast<unsigned long long>(n), partials);
    else
      hipLaunchKernelGGL((epipolar_moments_kernel<float>), dim3(grid), dim3(64), 0, stream, pl,
                         static_cast<unsigned long long>(n), partials);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
  }
  hipLaunchKernelGGL(epipolar_fold_kernel, dim3(kMom), dim3(1024), 0, stream, partials, grid, groups_dev);
  return hipGetLastError();
}

}  // namespace sba
