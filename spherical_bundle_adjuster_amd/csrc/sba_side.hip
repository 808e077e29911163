// Kernels either side of the sweep: AoS -> plane re-layout at upload time, pixel -> unit sphere (reference
// spherical_bundle_adjuster.cpp:271-298), ERP -> cubemap strip (equi2cube.cpp:12-302), and their launchers.
#include "sba_device.hpp"

namespace sba {
namespace {

// ---- layout conversion at upload time (once per problem, not per LM iteration) ---------------
template <typename ST>
__global__ void aos_to_planes_kernel(const double* __restrict__ aos, size_t n, size_t first,
                                     ST* __restrict__ px, ST* __restrict__ py, ST* __restrict__ pz) {
  const size_t i = static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i >= n) return;
  px[first + i] = static_cast<ST>(aos[3 * i + 0]);
  py[first + i] = static_cast<ST>(aos[3 * i + 1]);
  pz[first + i] = static_cast<ST>(aos[3 * i + 2]);
}
__global__ void d12_to_planes_kernel(const double* __restrict__ d12, size_t n, size_t first,
                                     double* __restrict__ d1, double* __restrict__ d2) {
  const size_t i = static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const double2 d = reinterpret_cast<const double2*>(d12)[i];
  d1[first + i] = d.x;
  d2[first + i] = d.y;
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

// ---- ERP -> cubemap strip (reference equi2cube.cpp:12-302) ----------------------------------------
// Output strip is S x 6S, faces left,front,right,back,top,bottom (equi2cube.cpp:292-298).  Per
// output pixel (i = row, j = column inside the face) the face-specific direction
// (equi2cube.cpp:28-30, 73-75, 118-120, 163-165, 208-210, 253-255) is normalised and mapped to a
// source pixel with truncation (equi2cube.cpp:40-50).  Each lane produces PIX consecutive output
// pixels so that stores are whole dwords; the gather side is byte loads (poor locality at the
// poles is inherent to the mapping).
__device__ __forceinline__ int erp_source_index(int face, int i, int j, int S, int im_h, int im_w) {
  const double s = static_cast<double>(S);
  const double a = (s - 2.0 * j) / s;   // (cube_size - 2 j) / cube_size
  const double b = (s - 2.0 * i) / s;   // (cube_size - 2 i) / cube_size
  const double an = (2.0 * j - s) / s;  // (2 j - cube_size) / cube_size
  const double bn = (2.0 * i - s) / s;
  double x, y, z;
  switch (face) {
    case 0: x = a;    y = 1.0;  z = b;    break;  // left   (.cpp:118-120)
    case 1: x = -1.0; y = a;    z = b;    break;  // front  (.cpp:73-75)
    case 2: x = an;   y = -1.0; z = b;    break;  // right  (.cpp:163-165)
    case 3: x = 1.0;  y = an;   z = b;    break;  // back   (.cpp:28-30)
    case 4: x = b;    y = a;    z = 1.0;  break;  // top    (.cpp:208-210)
    default: x = bn;  y = a;    z = -1.0; break;  // bottom (.cpp:253-255)
  }
  const double kPi = 3.14159265358979323846;
  const double nrm = sqrt(x * x + y * y + z * z);
  const double ux = x / nrm, uy = y / nrm, uz = z / nrm;
  const double theta = acos(uz);
  double phi = atan2(uy, ux);
  if (phi < 0) phi += kPi * 2;
  int row = static_cast<int>(im_h * theta / kPi);
  int col = static_cast<int>(im_w * phi / (2 * kPi));
  // The reference does not clamp (equi2cube.cpp:47-50); only the exact pole could leave the image.
  row = min(max(row, 0), im_h - 1);
  col = min(max(col, 0), im_w - 1);
  return row * im_w + col;
}

template <int PIX>
__global__ __launch_bounds__(256) void equi2cube_kernel(const uint8_t* __restrict__ erp, int im_h,
                                                        int im_w, int S, uint8_t* __restrict__ out,
                                                        size_t erp_stride, size_t out_stride, int batch,
                                                        int frames_per_block) {
  const int groups_per_row = (6 * S) / PIX;
  const size_t g = static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (g >= static_cast<size_t>(groups_per_row) * S) return;
  const int i = static_cast<int>(g / groups_per_row);
  const int c0 = static_cast<int>(g % groups_per_row) * PIX;   // strip column of the first pixel
  // The mapping depends only on (S, H, W): the f64 sqrt/acos/atan2 work is done once per output pixel and
  // reused for every frame of this block's slice of the batch; per frame only the gather and the store remain.
  size_t si[PIX];
#pragma unroll
  for (int k = 0; k < PIX; ++k) {
    const int c = c0 + k;
    const int face = c / S, j = c - face * S;
    si[k] = static_cast<size_t>(erp_source_index(face, i, j, S, im_h, im_w)) * 3;
  }
  const size_t o = (static_cast<size_t>(i) * 6 * S + c0) * 3;
  const int f0 = blockIdx.y * frames_per_block;
  const int f1 = min(batch, f0 + frames_per_block);
  for (int f = f0; f < f1; ++f) {
    const uint8_t* src = erp + static_cast<size_t>(f) * erp_stride;
    uint8_t* dst = out + static_cast<size_t>(f) * out_stride;
    uint8_t px[3 * PIX];
#pragma unroll
    for (int k = 0; k < PIX; ++k) {
      px[3 * k + 0] = src[si[k] + 0];
      px[3 * k + 1] = src[si[k] + 1];
      px[3 * k + 2] = src[si[k] + 2];
    }
    if (PIX == 4) {
      uint32_t* o32 = reinterpret_cast<uint32_t*>(dst + o);   // 12-byte group, 4-byte aligned
#pragma unroll
      for (int w = 0; w < 3; ++w)
        o32[w] = static_cast<uint32_t>(px[4 * w]) | (static_cast<uint32_t>(px[4 * w + 1]) << 8) |
                 (static_cast<uint32_t>(px[4 * w + 2]) << 16) | (static_cast<uint32_t>(px[4 * w + 3]) << 24);
    } else {
#pragma unroll
      for (int b = 0; b < 3 * PIX; ++b) dst[o + b] = px[b];
    }
  }
}

}  // namespace

hipError_t launch_aos_to_planes(const double* aos, size_t n, size_t first, void* px, void* py,
                                void* pz, int store, hipStream_t stream) {
  if (n == 0) return hipSuccess;
  const unsigned grid = static_cast<unsigned>((n + 255) / 256);
  if (store == 0)
    hipLaunchKernelGGL((aos_to_planes_kernel<double>), dim3(grid), dim3(256), 0, stream, aos, n, first,
                       static_cast<double*>(px), static_cast<double*>(py), static_cast<double*>(pz));
  else
    hipLaunchKernelGGL((aos_to_planes_kernel<float>), dim3(grid), dim3(256), 0, stream, aos, n, first,
                       static_cast<float*>(px), static_cast<float*>(py), static_cast<float*>(pz));
  return hipGetLastError();
}

hipError_t launch_d12_to_planes(const double* d12, size_t n, size_t first, double* d1, double* d2,
                                hipStream_t stream) {
  if (n == 0) return hipSuccess;
  const unsigned grid = static_cast<unsigned>((n + 255) / 256);
  hipLaunchKernelGGL(d12_to_planes_kernel, dim3(grid), dim3(256), 0, stream, d12, n, first, d1, d2);
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

hipError_t launch_equi2cube(const uint8_t* erp, int im_h, int im_w, int cube, int batch, uint8_t* out,
                            hipStream_t stream) {
  if (cube <= 0 || batch <= 0) return hipSuccess;
  const size_t erp_stride = static_cast<size_t>(im_h) * im_w * 3;
  const size_t out_stride = static_cast<size_t>(cube) * 6 * cube * 3;
  const bool wide = (6 * cube) % 4 == 0 && cube % 4 == 0;
  const size_t groups = wide ? static_cast<size_t>(6 * cube / 4) * cube : static_cast<size_t>(6 * cube) * cube;
  const unsigned gx = static_cast<unsigned>((groups + 255) / 256);
  // frames per block: amortise the index computation over the batch, but keep >= ~2048 blocks in flight
  int fpb = 1;
  while (fpb < 16 && fpb * 2 <= batch && static_cast<size_t>(gx) * ((batch + 2 * fpb - 1) / (2 * fpb)) >= 2048) fpb *= 2;
  const unsigned gy = static_cast<unsigned>((batch + fpb - 1) / fpb);
  if (wide)
    hipLaunchKernelGGL((equi2cube_kernel<4>), dim3(gx, gy), dim3(256), 0, stream, erp, im_h, im_w, cube, out,
                       erp_stride, out_stride, batch, fpb);
  else
    hipLaunchKernelGGL((equi2cube_kernel<1>), dim3(gx, gy), dim3(256), 0, stream, erp, im_h, im_w, cube, out,
                       erp_stride, out_stride, batch, fpb);
  return hipGetLastError();
}

}  // namespace sba
