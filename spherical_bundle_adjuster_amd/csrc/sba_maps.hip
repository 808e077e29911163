// Key-point / image coordinate maps of the reference's matchers: the pieces
// of spherical_surf and equi2cube_surf that turn matcher output into the ERP key-points the BA path consumes.
//   rotate_keypoints      spherical_surf::rotate_keypoint   (spherical_surf.cpp:110-123) via rotate_pixel (:48-74)
//   crop_rotated_image    spherical_surf::crop_rotated_image (spherical_surf.cpp:76-108)
//   cube2equi_keypoints   equi2cube_surf::cube2equi_pixel    (equi2cube_surf.cpp:19-76)
// Integer pixel indices come from the same truncations as the reference (Vec2i), in f64 like the reference.
#include <cstring>

#include "sba_internal.hpp"

namespace sba {
namespace {

constexpr double kPi = 3.14159265358979323846;

struct Rot3 { double m[9]; };

// rotate_pixel (spherical_surf.cpp:48-74)
__device__ __forceinline__ void rotate_pixel(int row, int col, const Rot3& R, int width, int height, int* out_row,
                                             int* out_col) {
  const double r0 = kPi * row / height, r1 = 2 * kPi * col / width;
  const double s0 = sin(r0);
  const double v0 = s0 * cos(r1), v1 = s0 * sin(r1), v2 = cos(r0);
  const double w0 = R.m[0] * v0 + R.m[1] * v1 + R.m[2] * v2;
  const double w1 = R.m[3] * v0 + R.m[4] * v1 + R.m[5] * v2;
  const double w2 = R.m[6] * v0 + R.m[7] * v1 + R.m[8] * v2;
  const double a = acos(w2);
  double b = atan2(w1, w0);
  if (b < 0) b += kPi * 2;
  *out_row = static_cast<int>(height * a / kPi);
  *out_col = static_cast<int>(width * b / (2 * kPi));
}

__global__ void rotate_keypoints_kernel(uint8_t* __restrict__ kp, size_t n, size_t stride, Rot3 R, int width,
                                        int height) {
  const size_t i = static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float* rec = reinterpret_cast<float*>(kp + i * stride);
  const float px = rec[0], py = rec[1];
  const int offset_i = static_cast<int>(py + static_cast<float>(height * 3 / 8));   // .cpp:116
  int r, c;
  rotate_pixel(offset_i, static_cast<int>(px), R, width, height, &r, &c);
  rec[0] = static_cast<float>(c);
  rec[1] = static_cast<float>(r);
}

__global__ void crop_rotated_kernel(const uint8_t* __restrict__ im, int im_h, int im_w, Rot3 R,
                                    uint8_t* __restrict__ out) {
  const size_t g = static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  const int rows = im_h / 4;
  if (g >= static_cast<size_t>(rows) * im_w) return;
  const int i = static_cast<int>(g / im_w), j = static_cast<int>(g % im_w);
  int r, c;
  rotate_pixel(i + im_h * 3 / 8, j, R, im_w, im_h, &r, &c);               // inverse warping, .cpp:91-96
  uint8_t b0 = 0, b1 = 0, b2 = 0;                                          // outside: 0 (reference: uninitialised)
  if (r >= 0 && c >= 0 && r < im_h && c < im_w) {                           // .cpp:100
    const uint8_t* s = im + (static_cast<size_t>(r) * im_w + c) * 3;
    b0 = s[0]; b1 = s[1]; b2 = s[2];
  }
  uint8_t* d = out + g * 3;
  d[0] = b0; d[1] = b1; d[2] = b2;
}

__global__ void cube2equi_kernel(uint8_t* __restrict__ kp, size_t n, size_t stride, int S, int im_w, int im_h) {
  const size_t i = static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float* rec = reinterpret_cast<float*>(kp + i * stride);
  const float cx = rec[0], cy = rec[1];
  const double s = S;
  double x = 0, y = 0, z = 0;
  if (cx < S) { x = (s - 2.0 * cx) / s; y = 1.0; z = (s - 2.0 * cy) / s; }                                       // left
  else if (cx >= S && cx < 2 * S) { x = -1.0; y = (s - 2.0 * (cx - S)) / s; z = (s - 2.0 * cy) / s; }           // front
  else if (cx >= 2 * S && cx < 3 * S) { x = (2.0 * (cx - 2 * S) - s) / s; y = -1.0; z = (s - 2.0 * cy) / s; }   // right
  else if (cx >= 3 * S && cx < 4 * S) { x = 1.0; y = (2.0 * (cx - 3 * S) - s) / s; z = (s - 2.0 * cy) / s; }    // back
  else if (cx >= 4 * S && cx < 5 * S) { x = (s - 2.0 * cy) / s; y = (s - 2.0 * (cx - 4 * S)) / s; z = 1.0; }    // top
  else if (cx >= 5 * S) { x = (2.0 * cy - s) / s; y = (s - 2.0 * (cx - 5 * S)) / s; z = -1.0; }                 // bottom
  const double nrm = sqrt(x * x + y * y + z * z);
  const double a = acos(z / nrm);
  double b = atan2(y / nrm, x / nrm);
  if (b < 0) b += kPi * 2;
  rec[0] = static_cast<float>(im_w * b / (2 * kPi));
  rec[1] = static_cast<float>(im_h * a / kPi);
}

// eular2rot(Vec3f(0, RAD(pitch), 0)) (spherical_surf.cpp:18-45): float angle, float cos/sin (std::cos(float)),
// R_x = R_z = I so R = R_y.
Rot3 pitch_rotation(float pitch_deg) {
  const float th = static_cast<float>(kPi * pitch_deg / 180.0);
  const double c = cosf(th), s = sinf(th);
  return Rot3{{c, 0, s, 0, 1, 0, -s, 0, c}};
}

int require_device(int device) {
  int count = 0;
  const hipError_t e = hipGetDeviceCount(&count);
  if (e != hipSuccess || count <= 0)
    return set_error(SBA_ERR_NO_DEVICE, "no HIP device available (%s); this library has no CPU path",
                     e != hipSuccess ? hipGetErrorString(e) : "device count is 0");
  if (device < 0 || device >= count) return set_error(SBA_ERR_INVALID_ARG, "device %d out of range [0,%d)", device, count);
  SBA_TRY_HIP(hipSetDevice(device));
  return SBA_OK;
}

// Round trip of `bytes` of key-point records through the device around `launch`.
template <typename Launch>
int with_device_copy(int device, void* host, size_t bytes, Launch&& launch) {
  int rc = require_device(device);
  if (rc) return rc;
  if (bytes == 0) return SBA_OK;
  DeviceBuffer dev;
  SBA_TRY_HIP(dev.alloc(bytes));
  SBA_TRY_HIP(hipMemcpy(dev.ptr, host, bytes, hipMemcpyHostToDevice));
  launch(dev.as<uint8_t>());
  SBA_TRY_HIP(hipGetLastError());
  SBA_TRY_HIP(hipMemcpy(host, dev.ptr, bytes, hipMemcpyDeviceToHost));
  return SBA_OK;
}

}  // namespace
}  // namespace sba

extern "C" {

int sba_rotate_keypoints(int device, void* keypoints, size_t n, size_t stride_bytes, float pitch_deg, int im_width,
                         int im_height) {
  if (n > 0 && !keypoints) return sba::set_error(SBA_ERR_INVALID_ARG, "null key-point array");
  if (stride_bytes < 8 || stride_bytes % 4 != 0) return sba::set_error(SBA_ERR_INVALID_ARG, "stride_bytes must be a multiple of 4 and >= 8");
  if (im_width <= 0 || im_height <= 0) return sba::set_error(SBA_ERR_INVALID_ARG, "bad image size");
  const sba::Rot3 R = sba::pitch_rotation(pitch_deg);
  return sba::with_device_copy(device, keypoints, n * stride_bytes, [&](uint8_t* dev) {
    hipLaunchKernelGGL(sba::rotate_keypoints_kernel, dim3(static_cast<unsigned>((n + 255) / 256)), dim3(256), 0, nullptr,
                       dev, n, stride_bytes, R, im_width, im_height);
  });
}

int sba_cube2equi_keypoints(int device, void* keypoints, size_t n, size_t stride_bytes, int cube_size, int im_width,
                            int im_height) {
  if (n > 0 && !keypoints) return sba::set_error(SBA_ERR_INVALID_ARG, "null key-point array");
  if (stride_bytes < 8 || stride_bytes % 4 != 0) return sba::set_error(SBA_ERR_INVALID_ARG, "stride_bytes must be a multiple of 4 and >= 8");
  if (im_width <= 0 || im_height <= 0 || cube_size <= 0) return sba::set_error(SBA_ERR_INVALID_ARG, "bad image / cube size");
  return sba::with_device_copy(device, keypoints, n * stride_bytes, [&](uint8_t* dev) {
    hipLaunchKernelGGL(sba::cube2equi_kernel, dim3(static_cast<unsigned>((n + 255) / 256)), dim3(256), 0, nullptr, dev,
                       n, stride_bytes, cube_size, im_width, im_height);
  });
}

int sba_crop_rotated_image(int device, const uint8_t* erp, int im_height, int im_width, float pitch_deg,
                           uint8_t* out) {
  if (!erp || !out) return sba::set_error(SBA_ERR_INVALID_ARG, "null image pointer");
  if (im_height < 4 || im_width <= 0) return sba::set_error(SBA_ERR_INVALID_ARG, "bad image size");
  int rc = sba::require_device(device);
  if (rc) return rc;
  const size_t in_bytes = static_cast<size_t>(im_height) * im_width * 3;
  const size_t px = static_cast<size_t>(im_height / 4) * im_width;
  sba::DeviceBuffer in_dev, out_dev;
  SBA_TRY_HIP(in_dev.alloc(in_bytes));
  SBA_TRY_HIP(out_dev.alloc(px * 3));
  SBA_TRY_HIP(hipMemcpy(in_dev.ptr, erp, in_bytes, hipMemcpyHostToDevice));
  const sba::Rot3 R = sba::pitch_rotation(pitch_deg);
  hipLaunchKernelGGL(sba::crop_rotated_kernel, dim3(static_cast<unsigned>((px + 255) / 256)), dim3(256), 0, nullptr,
                     in_dev.as<uint8_t>(), im_height, im_width, R, out_dev.as<uint8_t>());
  SBA_TRY_HIP(hipGetLastError());
  SBA_TRY_HIP(hipMemcpy(out, out_dev.ptr, px * 3, hipMemcpyDeviceToHost));
  return SBA_OK;
}

}  // extern "C"
