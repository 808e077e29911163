// Kernels either side of the sweep: AoS -> plane re-layout at upload time, pixel -> unit sphere (reference
// spherical_bundle_adjuster.cpp:271-298), and their launchers.  (The image / key-point maps live in sba_maps.hip.)
#include "sba_device.hpp"

namespace sba {
namespace {

// ---- layout conversion at upload time (once per problem, not per LM iteration) ---------------
// tile_elems == 0: element i -> first + i.  Otherwise (interleaved batch layout, sba_device.hpp: PairDesc): tiles of
// tile_elems consecutive elements sit tile_stride_elems apart.
__device__ __forceinline__ size_t tiled_index(size_t first, size_t i, size_t tile_elems, size_t tile_stride_elems) {
  return tile_elems == 0 ? first + i : first + (i / tile_elems) * tile_stride_elems + i % tile_elems;
}
template <typename ST>
__global__ void aos_to_planes_kernel(const double* __restrict__ aos, size_t n, size_t first,
                                     ST* __restrict__ px, ST* __restrict__ py, ST* __restrict__ pz, size_t tile_elems,
                                     size_t tile_stride_elems) {
  const size_t i = static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const size_t o = tiled_index(first, i, tile_elems, tile_stride_elems);
  px[o] = static_cast<ST>(aos[3 * i + 0]);
  py[o] = static_cast<ST>(aos[3 * i + 1]);
  pz[o] = static_cast<ST>(aos[3 * i + 2]);
}
__global__ void d12_to_planes_kernel(const double* __restrict__ d12, size_t n, size_t first,
                                     double* __restrict__ d1, double* __restrict__ d2, size_t tile_elems, size_t tile_stride_elems) {
  const size_t i = static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const double2 d = reinterpret_cast<const double2*>(d12)[i];
  const size_t o = tiled_index(first, i, tile_elems, tile_stride_elems);
  d1[o] = d.x;
  d2[o] = d.y;
}
// The same re-layout for a whole batch in one launch: rows [first_row, first_row + m) of the caller's CONCATENATED arrays
// (pair g = rows offsets[g] .. offsets[g + 1]) sit in `stage`; every row finds its pair by bisection of the offsets (a few KB,
// L2-resident) and goes to its place in the pair's tiles (PairDesc).  One launch per staged chunk instead of one per pair.
__device__ __forceinline__ size_t batch_row_index(size_t row, const unsigned long long* __restrict__ offsets, int num_pairs,
                                                  const PairDesc* __restrict__ desc, size_t ppt) {
  int lo = 0, hi = num_pairs;                 // offsets[lo] <= row < offsets[hi]; empty pairs repeat an offset: the last one wins
  while (hi - lo > 1) {
    const int mid = (lo + hi) >> 1;
    if (offsets[mid] <= row) lo = mid; else hi = mid;
  }
  const PairDesc d = desc[lo];
  return tiled_index(d.first_vec * ppt, row - offsets[lo], kPairTile * ppt, d.tile_stride * ppt);
}
template <typename ST>
__global__ void batch_aos_to_planes_kernel(const double* __restrict__ aos, size_t m, size_t first_row,
                                           const unsigned long long* __restrict__ offsets, int num_pairs,
                                           const PairDesc* __restrict__ desc, size_t ppt, ST* __restrict__ px,
                                           ST* __restrict__ py, ST* __restrict__ pz) {
  const size_t i = static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i >= m) return;
  const size_t o = batch_row_index(first_row + i, offsets, num_pairs, desc, ppt);
  px[o] = static_cast<ST>(aos[3 * i + 0]);
  py[o] = static_cast<ST>(aos[3 * i + 1]);
  pz[o] = static_cast<ST>(aos[3 * i + 2]);
}
__global__ void batch_d12_to_planes_kernel(const double* __restrict__ d12, size_t m, size_t first_row,
                                           const unsigned long long* __restrict__ offsets, int num_pairs,
                                           const PairDesc* __restrict__ desc, size_t ppt, double* __restrict__ d1,
                                           double* __restrict__ d2) {
  const size_t i = static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i >= m) return;
  const double2 d = reinterpret_cast<const double2*>(d12)[i];
  const size_t o = batch_row_index(first_row + i, offsets, num_pairs, desc, ppt);
  d1[o] = d.x;
  d2[o] = d.y;
}
__global__ void planes_to_d12_kernel(const double* __restrict__ d1, const double* __restrict__ d2,
                                     size_t n, double* __restrict__ d12) {
  const size_t i = static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i >= n) return;
  reinterpret_cast<double2*>(d12)[i] = make_double2(d1[i], d2[i]);
}

// ---- pixel -> unit sphere (reference spherical_bundle_adjuster.cpp:271-298) ---------------------
//   lon = 2 pi (pt.x / W), colat = pi (pt.y / H);  v = (sin colat cos lon, sin colat sin lon, cos colat)
// pt.x / pt.y are the first two floats of each `stride_bytes`-byte key-point record.
__global__ void keypoints_to_sphere_kernel(const uint8_t* __restrict__ kp, size_t n, size_t stride_bytes,
                                           double im_w, double im_h, double* __restrict__ out_xyz) {
  const size_t i = static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float* rec = reinterpret_cast<const float*>(kp + i * stride_bytes);
  const double px = static_cast<double>(rec[0]), py = static_cast<double>(rec[1]);
  const double kPi = 3.14159265358979323846;
  const double lon = 2 * kPi * (px / im_w);
  const double colat = kPi * (py / im_h);
  const double sc = sin(colat), cc = cos(colat);
  out_xyz[3 * i + 0] = sc * cos(lon);
  out_xyz[3 * i + 1] = sc * sin(lon);
  out_xyz[3 * i + 2] = cc;
}

// Same map, fused with the upload: key-point records of BOTH images -> the six coordinate planes of a problem
// (no host-side cv::Point3d arrays in between).
template <typename ST>
__global__ void keypoints_to_planes_kernel(const uint8_t* __restrict__ kp_left, const uint8_t* __restrict__ kp_right,
                                           size_t n, size_t stride_bytes, double im_w, double im_h,
                                           ST* __restrict__ x1x, ST* __restrict__ x1y, ST* __restrict__ x1z,
                                           ST* __restrict__ x2x, ST* __restrict__ x2y, ST* __restrict__ x2z) {
  const size_t i = static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const double kPi = 3.14159265358979323846;
  const float* l = reinterpret_cast<const float*>(kp_left + i * stride_bytes);
  const float* r = reinterpret_cast<const float*>(kp_right + i * stride_bytes);
  const double lon1 = 2 * kPi * (static_cast<double>(l[0]) / im_w), col1 = kPi * (static_cast<double>(l[1]) / im_h);
  const double lon2 = 2 * kPi * (static_cast<double>(r[0]) / im_w), col2 = kPi * (static_cast<double>(r[1]) / im_h);
  const double s1 = sin(col1), s2 = sin(col2);
  x1x[i] = static_cast<ST>(s1 * cos(lon1)); x1y[i] = static_cast<ST>(s1 * sin(lon1)); x1z[i] = static_cast<ST>(cos(col1));
  x2x[i] = static_cast<ST>(s2 * cos(lon2)); x2y[i] = static_cast<ST>(s2 * sin(lon2)); x2z[i] = static_cast<ST>(cos(col2));
}

}  // namespace

hipError_t launch_aos_to_planes(const double* aos, size_t n, size_t first, void* px, void* py,
                                void* pz, int store, hipStream_t stream, size_t tile_elems, size_t tile_stride_elems) {
  if (n == 0) return hipSuccess;
  const unsigned grid = static_cast<unsigned>((n + 255) / 256);
  if (store == 0)
    hipLaunchKernelGGL((aos_to_planes_kernel<double>), dim3(grid), dim3(256), 0, stream, aos, n, first,
                       static_cast<double*>(px), static_cast<double*>(py), static_cast<double*>(pz), tile_elems, tile_stride_elems);
  else
    hipLaunchKernelGGL((aos_to_planes_kernel<float>), dim3(grid), dim3(256), 0, stream, aos, n, first,
                       static_cast<float*>(px), static_cast<float*>(py), static_cast<float*>(pz), tile_elems, tile_stride_elems);
  return hipGetLastError();
}

hipError_t launch_d12_to_planes(const double* d12, size_t n, size_t first, double* d1, double* d2,
                                hipStream_t stream, size_t tile_elems, size_t tile_stride_elems) {
  if (n == 0) return hipSuccess;
  const unsigned grid = static_cast<unsigned>((n + 255) / 256);
  hipLaunchKernelGGL(d12_to_planes_kernel, dim3(grid), dim3(256), 0, stream, d12, n, first, d1, d2, tile_elems, tile_stride_elems);
  return hipGetLastError();
}

hipError_t launch_batch_aos_to_planes(const double* aos, size_t m, size_t first_row, const unsigned long long* offsets_dev,
                                      int num_pairs, const PairDesc* desc, size_t ppt, void* px, void* py, void* pz, int store,
                                      hipStream_t stream) {
  if (m == 0 || num_pairs <= 0) return hipSuccess;
  const unsigned grid = static_cast<unsigned>((m + 255) / 256);
  if (store == 0)
    hipLaunchKernelGGL((batch_aos_to_planes_kernel<double>), dim3(grid), dim3(256), 0, stream, aos, m, first_row, offsets_dev,
                       num_pairs, desc, ppt, static_cast<double*>(px), static_cast<double*>(py), static_cast<double*>(pz));
  else
    hipLaunchKernelGGL((batch_aos_to_planes_kernel<float>), dim3(grid), dim3(256), 0, stream, aos, m, first_row, offsets_dev,
                       num_pairs, desc, ppt, static_cast<float*>(px), static_cast<float*>(py), static_cast<float*>(pz));
  return hipGetLastError();
}

hipError_t launch_batch_d12_to_planes(const double* d12, size_t m, size_t first_row, const unsigned long long* offsets_dev,
                                      int num_pairs, const PairDesc* desc, size_t ppt, double* d1, double* d2,
                                      hipStream_t stream) {
  if (m == 0 || num_pairs <= 0) return hipSuccess;
  const unsigned grid = static_cast<unsigned>((m + 255) / 256);
  hipLaunchKernelGGL(batch_d12_to_planes_kernel, dim3(grid), dim3(256), 0, stream, d12, m, first_row, offsets_dev, num_pairs, desc,
                     ppt, d1, d2);
  return hipGetLastError();
}

hipError_t launch_planes_to_d12(const double* d1, const double* d2, size_t n, double* d12,
                                hipStream_t stream) {
  if (n == 0) return hipSuccess;
  const unsigned grid = static_cast<unsigned>((n + 255) / 256);
  hipLaunchKernelGGL(planes_to_d12_kernel, dim3(grid), dim3(256), 0, stream, d1, d2, n, d12);
  return hipGetLastError();
}

hipError_t launch_keypoints_to_sphere(const uint8_t* kp, size_t n, size_t stride_bytes, double im_w,
                                      double im_h, double* out_xyz, hipStream_t stream) {
  if (n == 0) return hipSuccess;
  const unsigned grid = static_cast<unsigned>((n + 255) / 256);
  hipLaunchKernelGGL(keypoints_to_sphere_kernel, dim3(grid), dim3(256), 0, stream, kp, n, stride_bytes,
                     im_w, im_h, out_xyz);
  return hipGetLastError();
}

hipError_t launch_keypoints_to_planes(const uint8_t* kp_left, const uint8_t* kp_right, size_t n, size_t stride_bytes,
                                      double im_w, double im_h, void* const planes[6], int store, hipStream_t stream) {
  if (n == 0) return hipSuccess;
  const unsigned grid = static_cast<unsigned>((n + 255) / 256);
  if (store == 0)
    hipLaunchKernelGGL((keypoints_to_planes_kernel<double>), dim3(grid), dim3(256), 0, stream, kp_left, kp_right, n,
                       stride_bytes, im_w, im_h, static_cast<double*>(planes[0]), static_cast<double*>(planes[1]),
                       static_cast<double*>(planes[2]), static_cast<double*>(planes[3]), static_cast<double*>(planes[4]),
                       static_cast<double*>(planes[5]));
  else
    hipLaunchKernelGGL((keypoints_to_planes_kernel<float>), dim3(grid), dim3(256), 0, stream, kp_left, kp_right, n,
                       stride_bytes, im_w, im_h, static_cast<float*>(planes[0]), static_cast<float*>(planes[1]),
                       static_cast<float*>(planes[2]), static_cast<float*>(planes[3]), static_cast<float*>(planes[4]),
                       static_cast<float*>(planes[5]));
  return hipGetLastError();
}

}  // namespace sba
