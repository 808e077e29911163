// HIP kernels (gfx950 / CDNA4, wave64) of the spherical bundle-adjustment hot path.
//
// sweep_kernel<MODE, DEPTH, ST> is the fused replacement of one Ceres residual+Jacobian
// evaluation over all residual blocks (reference spherical_bundle_adjuster.cpp:843-868,
// :891-919, :947-976 evaluated through AutoDiffCostFunction + HuberLoss(1.0), :887/:943/:1000):
//
//   per correspondence i (one "evaluation"):
//       e  = t + d2 x2_i + d1 Rn x1_i                        residual            (.cpp:897-916)
//       A  = d1 [Gn_0 x1_i | Gn_1 x1_i | Gn_2 x1_i]          d e / d rot  (what Jet<double,3> carries)
//       s  = e.e ;  w = rho'(s) ;  rho(s)                    block-wise Huber (Ceres corrector with
//                                                            rho'' <= 0: scale e and J by sqrt(w))
//       acc += w A^T A, w A^T, w, w A^T e, w e, rho/2        23 running sums (+ outlier count)
//   wave:   DPP butterfly over the 64 lanes
//   block:  4 waves through LDS -> partials[block][24]
//   grid:   finalize_kernel folds partials in a fixed order -> pack[24]  (deterministic)
//
// Memory: the six coordinate planes are read exactly once, 16 B per lane per load instruction
// (1 KiB contiguous per wave instruction).  No reuse, no MFMA: the kernel is HBM-bound
// (48 B per evaluation with f64 planes and uniform depths).  The wave-uniform R|t state
// (SweepParams) arrives as a by-value kernel argument and is staged once per block in LDS.
#include "sba_device.hpp"

#ifndef SBA_PARAMS_IN_LDS
#define SBA_PARAMS_IN_LDS 0
#endif

namespace sba {
namespace {

constexpr int MODE_ROT = 0, MODE_TRAN = 1, MODE_RT = 2;
constexpr int DEPTH_UNIFORM = 0, DEPTH_PER_MATCH = 1;

// ---- accumulator <-> pack slot maps -------------------------------------------------------
template <int MODE> struct AccMap;
template <> struct AccMap<MODE_ROT> {   // haa[6] ga[3] cost nout
  static constexpr int N = 11;
  __host__ __device__ static constexpr int slot(int k) {
    return k < 6 ? k : (k < 9 ? 16 + (k - 6) : (k == 9 ? 22 : 23));
  }
};
template <> struct AccMap<MODE_TRAN> {  // sw gt[3] cost nout
  static constexpr int N = 6;
  __host__ __device__ static constexpr int slot(int k) {
    return k == 0 ? 15 : (k < 4 ? 19 + (k - 1) : (k == 4 ? 22 : 23));
  }
};
template <> struct AccMap<MODE_RT> {    // the full pack
  static constexpr int N = 24;
  __host__ __device__ static constexpr int slot(int k) { return k; }
};

// ---- wave64 sum via DPP; the total ends up in lane 63 --------------------------------------
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double dpp_f64(double v) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, ROW_MASK, 0xf, false);
  hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, ROW_MASK, 0xf, false);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double wave_sum_to_lane63(double v) {
  v += dpp_f64<0xB1, 0xf>(v);   // quad_perm [1,0,3,2]
  v += dpp_f64<0x4E, 0xf>(v);   // quad_perm [2,3,0,1]
  v += dpp_f64<0x141, 0xf>(v);  // row_half_mirror
  v += dpp_f64<0x140, 0xf>(v);  // row_mirror            -> every lane: its 16-lane row sum
  v += dpp_f64<0x142, 0xa>(v);  // row_bcast15 into rows 1,3
  v += dpp_f64<0x143, 0xc>(v);  // row_bcast31 into rows 2,3 -> lane 63: wave sum
  return v;
}

// ---- vector loads: 16 bytes per lane ---------------------------------------------------------
template <typename ST> struct Vec;
template <> struct Vec<double> {
  static constexpr int PPT = 2;
  double v[2];
  __device__ __forceinline__ void load(const void* plane, size_t vec_index) {
    const double2 q = reinterpret_cast<const double2*>(plane)[vec_index];
    v[0] = q.x; v[1] = q.y;
  }
};
template <> struct Vec<float> {
  static constexpr int PPT = 4;
  double v[4];
  __device__ __forceinline__ void load(const void* plane, size_t vec_index) {
    const float4 q = reinterpret_cast<const float4*>(plane)[vec_index];
    v[0] = q.x; v[1] = q.y; v[2] = q.z; v[3] = q.w;
  }
};

// ---- Huber: w = rho'(s), r = rho(s) ---------------------------------------------------------
// Outlier region needs 1/sqrt(s): v_rsq_f64 seed + two Newton steps (full f64 accuracy to a
// couple of ulp) instead of the library sqrt + divide (~40 instructions).
__device__ __forceinline__ void huber(double s, double delta, double delta2, double& w, double& rho,
                                      double& is_out) {
  double y = __builtin_amdgcn_rsq(s);
  const double hs = 0.5 * s;
  y = y * __builtin_fma(-hs * y, y, 1.5);
  y = y * __builtin_fma(-hs * y, y, 1.5);
  const bool out = s > delta2;
  const double sq = s * y;                                     // sqrt(s)
  w = out ? delta * y : 1.0;
  rho = out ? __builtin_fma(2.0 * delta, sq, -delta2) : s;
  is_out = out ? 1.0 : 0.0;
}

// ---- one correspondence ------------------------------------------------------------------------
// P points at the LDS copy of SweepParams (wave-uniform address -> broadcast reads).
template <int MODE, int DEPTH>
__device__ __forceinline__ void accumulate(const SweepParams* __restrict__ P, bool use_loss,
                                           double x, double y, double z, double u, double v,
                                           double q, double d1, double d2, bool valid,
                                           double* __restrict__ acc) {
  // r = Rn x1
  double r0 = P->Rn[0] * x + P->Rn[1] * y + P->Rn[2] * z;
  double r1 = P->Rn[3] * x + P->Rn[4] * y + P->Rn[5] * z;
  double r2 = P->Rn[6] * x + P->Rn[7] * y + P->Rn[8] * z;
  double e0, e1, e2;
  if (DEPTH == DEPTH_PER_MATCH) {
    e0 = __builtin_fma(d1, r0, __builtin_fma(d2, u, P->t[0]));
    e1 = __builtin_fma(d1, r1, __builtin_fma(d2, v, P->t[1]));
    e2 = __builtin_fma(d1, r2, __builtin_fma(d2, q, P->t[2]));
  } else {
    e0 = r0 + __builtin_fma(P->d2, u, P->t[0]);
    e1 = r1 + __builtin_fma(P->d2, v, P->t[1]);
    e2 = r2 + __builtin_fma(P->d2, q, P->t[2]);
  }
  const double s = e0 * e0 + e1 * e1 + e2 * e2;
  double w = 1.0, rho = s, is_out = 0.0;
  if (use_loss) huber(s, P->delta, P->delta2, w, rho, is_out);
  if (!valid) { w = 0.0; rho = 0.0; is_out = 0.0; }

  if (MODE == MODE_TRAN) {
    acc[0] += w;
    acc[1] = __builtin_fma(w, e0, acc[1]);
    acc[2] = __builtin_fma(w, e1, acc[2]);
    acc[3] = __builtin_fma(w, e2, acc[3]);
    acc[4] = __builtin_fma(0.5, rho, acc[4]);
    acc[5] += is_out;
    return;
  }

  // A[r][j] = (Gn_j x1)[r]
  double A[3][3];
#pragma unroll
  for (int j = 0; j < 3; ++j) {
#pragma unroll
    for (int r = 0; r < 3; ++r) {
      double a = P->Gn[9 * j + 3 * r + 0] * x + P->Gn[9 * j + 3 * r + 1] * y +
                 P->Gn[9 * j + 3 * r + 2] * z;
      if (DEPTH == DEPTH_PER_MATCH) a *= d1;
      A[r][j] = a;
    }
  }
  double wA[3][3];
#pragma unroll
  for (int r = 0; r < 3; ++r)
#pragma unroll
    for (int j = 0; j < 3; ++j) wA[r][j] = w * A[r][j];

  // sum w A^T A (upper), slots 0..5 : 00 01 02 11 12 22
  int k = 0;
#pragma unroll
  for (int a = 0; a < 3; ++a)
#pragma unroll
    for (int b = a; b < 3; ++b) {
      acc[k] = __builtin_fma(wA[0][a], A[0][b],
               __builtin_fma(wA[1][a], A[1][b], __builtin_fma(wA[2][a], A[2][b], acc[k])));
      ++k;
    }
  if (MODE == MODE_ROT) {
#pragma unroll
    for (int a = 0; a < 3; ++a)
      acc[6 + a] = __builtin_fma(wA[0][a], e0,
                   __builtin_fma(wA[1][a], e1, __builtin_fma(wA[2][a], e2, acc[6 + a])));
    acc[9] = __builtin_fma(0.5, rho, acc[9]);
    acc[10] += is_out;
  } else {  // MODE_RT: full pack layout
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
      for (int c = 0; c < 3; ++c) acc[6 + 3 * a + c] += wA[c][a];   // sum w A^T
    acc[15] += w;
#pragma unroll
    for (int a = 0; a < 3; ++a)
      acc[16 + a] = __builtin_fma(wA[0][a], e0,
                    __builtin_fma(wA[1][a], e1, __builtin_fma(wA[2][a], e2, acc[16 + a])));
    acc[19] = __builtin_fma(w, e0, acc[19]);
    acc[20] = __builtin_fma(w, e1, acc[20]);
    acc[21] = __builtin_fma(w, e2, acc[21]);
    acc[22] = __builtin_fma(0.5, rho, acc[22]);
    acc[23] += is_out;
  }
}

// One 16-byte vector of correspondences per lane (2 with f64 planes, 4 with f32 planes).
template <int MODE, int DEPTH, typename ST, bool CHECK>
__device__ __forceinline__ void process_vec(const Planes& pl, const SweepParams* __restrict__ P,
                                            bool use_loss, size_t p, size_t n,
                                            double* __restrict__ acc) {
  constexpr int PPT = Vec<ST>::PPT;
  Vec<ST> ax, ay, az, bx, by, bz;
  ax.load(pl.x1[0], p); ay.load(pl.x1[1], p); az.load(pl.x1[2], p);
  bx.load(pl.x2[0], p); by.load(pl.x2[1], p); bz.load(pl.x2[2], p);
  double d1v[PPT], d2v[PPT];
  if (DEPTH == DEPTH_PER_MATCH) {
#pragma unroll
    for (int h = 0; h < PPT / 2; ++h) {
      const double2 a = reinterpret_cast<const double2*>(pl.d1)[p * (PPT / 2) + h];
      const double2 b = reinterpret_cast<const double2*>(pl.d2)[p * (PPT / 2) + h];
      d1v[2 * h] = a.x; d1v[2 * h + 1] = a.y;
      d2v[2 * h] = b.x; d2v[2 * h + 1] = b.y;
    }
  } else {
#pragma unroll
    for (int h = 0; h < PPT; ++h) { d1v[h] = 1.0; d2v[h] = 0.0; }
  }
  const size_t first = p * PPT;
#pragma unroll
  for (int h = 0; h < PPT; ++h)
    accumulate<MODE, DEPTH>(P, use_loss, ax.v[h], ay.v[h], az.v[h], bx.v[h], by.v[h], bz.v[h],
                            d1v[h], d2v[h], CHECK ? (first + h < n) : true, acc);
}

template <int MODE, int DEPTH, typename ST>
__global__ __launch_bounds__(kBlock) void sweep_kernel(Planes pl, SweepParams prm,
                                                      double* __restrict__ partials) {
  constexpr int NACC = AccMap<MODE>::N;
  constexpr int PPT = Vec<ST>::PPT;
  // One LDS object: [0, 48) the staged R|t state (SBA_PARAMS_IN_LDS), then the cross-wave scratch.
  __shared__ double lds[48 + (kBlock / 64) * 24];
  double* wave_out = lds + 48;
  const int tid = threadIdx.x;
#if SBA_PARAMS_IN_LDS
  const SweepParams* P = reinterpret_cast<const SweepParams*>(lds);
  if (tid < 43) lds[tid] = reinterpret_cast<const double*>(&prm)[tid];
  __syncthreads();
#else
  const SweepParams* P = &prm;   // kernarg segment: scalar loads, operands stay in SGPRs
#endif

  const bool use_loss = P->delta > 0.0;
  const size_t n = prm.n;
  const size_t stride = static_cast<size_t>(gridDim.x) * kBlock;

  double acc[NACC];
#pragma unroll
  for (int k = 0; k < NACC; ++k) acc[k] = 0.0;

  // Full vectors: every lane valid, no masking in the hot loop.
  const size_t nfull = n / PPT;
#pragma unroll 2
  for (size_t p = static_cast<size_t>(blockIdx.x) * kBlock + tid; p < nfull; p += stride)
    process_vec<MODE, DEPTH, ST, false>(pl, P, use_loss, p, n, acc);
  // Ragged tail (n % PPT != 0): one lane of the grid handles the last, partly valid vector
  // (the planes are zero-padded to a whole vector at upload).
  if (nfull * PPT != n && blockIdx.x == gridDim.x - 1 && tid == kBlock - 1)
    process_vec<MODE, DEPTH, ST, true>(pl, P, use_loss, nfull, n, acc);

  // wave -> lane 63 -> LDS -> block partial (pack layout, unused slots zero)
  const int lane = tid & 63, wave = tid >> 6;
#pragma unroll
  for (int k = 0; k < NACC; ++k) {
    const double s = wave_sum_to_lane63(acc[k]);
    if (lane == 63) wave_out[wave * 24 + AccMap<MODE>::slot(k)] = s;
  }
  // slots this mode does not produce
  if (NACC < 24 && tid < 24) {
    bool used = false;
#pragma unroll
    for (int k = 0; k < NACC; ++k) used |= (AccMap<MODE>::slot(k) == tid);
    if (!used) {
#pragma unroll
      for (int wv = 0; wv < kBlock / 64; ++wv) wave_out[wv * 24 + tid] = 0.0;
    }
  }
  __syncthreads();
  if (tid < 24) {
    double s = wave_out[tid];
#pragma unroll
    for (int wv = 1; wv < kBlock / 64; ++wv) s += wave_out[wv * 24 + tid];
    partials[static_cast<size_t>(blockIdx.x) * 24 + tid] = s;
  }
}

// Fold partials[nblocks][24] in a fixed order: 8 strided serial chains per slot, then a serial
// sum of the 8 chains.  One block; independent of timing, so results are run-to-run identical.
__global__ __launch_bounds__(256) void finalize_kernel(const double* __restrict__ partials,
                                                       int nblocks, double* __restrict__ pack_out) {
  __shared__ double part[8][32];
  const int slot = threadIdx.x & 31, grp = threadIdx.x >> 5;
  double s = 0.0;
  if (slot < 24)
    for (int b = grp; b < nblocks; b += 8) s += partials[static_cast<size_t>(b) * 24 + slot];
  part[grp][slot] = s;
  __syncthreads();
  if (threadIdx.x < 24) {
    double tot = part[0][threadIdx.x];
#pragma unroll
    for (int g = 1; g < 8; ++g) tot += part[g][threadIdx.x];
    pack_out[threadIdx.x] = tot;
  }
}

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
                                                        size_t erp_stride, size_t out_stride) {
  const int groups_per_row = (6 * S) / PIX;
  const size_t g = static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (g >= static_cast<size_t>(groups_per_row) * S) return;
  const uint8_t* src = erp + static_cast<size_t>(blockIdx.y) * erp_stride;
  uint8_t* dst = out + static_cast<size_t>(blockIdx.y) * out_stride;
  const int i = static_cast<int>(g / groups_per_row);
  const int c0 = static_cast<int>(g % groups_per_row) * PIX;   // strip column of the first pixel
  uint8_t px[3 * PIX];
#pragma unroll
  for (int k = 0; k < PIX; ++k) {
    const int c = c0 + k;
    const int face = c / S, j = c - face * S;
    const size_t si = static_cast<size_t>(erp_source_index(face, i, j, S, im_h, im_w)) * 3;
    px[3 * k + 0] = src[si + 0];
    px[3 * k + 1] = src[si + 1];
    px[3 * k + 2] = src[si + 2];
  }
  const size_t o = (static_cast<size_t>(i) * 6 * S + c0) * 3;
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

template <int MODE, int DEPTH>
hipError_t launch_sweep_store(int store, const Planes& pl, const SweepParams& prm, double* partials,
                              int grid, hipStream_t stream) {
  if (store == 0)
    hipLaunchKernelGGL((sweep_kernel<MODE, DEPTH, double>), dim3(grid), dim3(kBlock), 0, stream, pl,
                       prm, partials);
  else
    hipLaunchKernelGGL((sweep_kernel<MODE, DEPTH, float>), dim3(grid), dim3(kBlock), 0, stream, pl,
                       prm, partials);
  return hipGetLastError();
}

}  // namespace

int points_per_lane(int store) { return store == 0 ? 2 : 4; }

hipError_t launch_sweep(int mode, int depth, int store, const Planes& pl, const SweepParams& prm,
                        double* partials, int grid, hipStream_t stream) {
  if (grid <= 0) return hipSuccess;
  switch (mode * 2 + depth) {
    case 0: return launch_sweep_store<MODE_ROT, DEPTH_UNIFORM>(store, pl, prm, partials, grid, stream);
    case 1: return launch_sweep_store<MODE_ROT, DEPTH_PER_MATCH>(store, pl, prm, partials, grid, stream);
    case 2: return launch_sweep_store<MODE_TRAN, DEPTH_UNIFORM>(store, pl, prm, partials, grid, stream);
    case 3: return launch_sweep_store<MODE_TRAN, DEPTH_PER_MATCH>(store, pl, prm, partials, grid, stream);
    case 4: return launch_sweep_store<MODE_RT, DEPTH_UNIFORM>(store, pl, prm, partials, grid, stream);
    case 5: return launch_sweep_store<MODE_RT, DEPTH_PER_MATCH>(store, pl, prm, partials, grid, stream);
  }
  return hipErrorInvalidValue;
}

hipError_t launch_finalize(const double* partials, int nblocks, double* pack_out, hipStream_t stream) {
  hipLaunchKernelGGL(finalize_kernel, dim3(1), dim3(256), 0, stream, partials, nblocks, pack_out);
  return hipGetLastError();
}

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

hipError_t launch_equi2cube(const uint8_t* erp, int im_h, int im_w, int cube, int batch, uint8_t* out,
                            hipStream_t stream) {
  if (cube <= 0 || batch <= 0) return hipSuccess;
  const size_t erp_stride = static_cast<size_t>(im_h) * im_w * 3;
  const size_t out_stride = static_cast<size_t>(cube) * 6 * cube * 3;
  if ((6 * cube) % 4 == 0 && cube % 4 == 0) {
    const size_t groups = static_cast<size_t>(6 * cube / 4) * cube;
    hipLaunchKernelGGL((equi2cube_kernel<4>), dim3(static_cast<unsigned>((groups + 255) / 256), batch),
                       dim3(256), 0, stream, erp, im_h, im_w, cube, out, erp_stride, out_stride);
  } else {
    const size_t groups = static_cast<size_t>(6 * cube) * cube;
    hipLaunchKernelGGL((equi2cube_kernel<1>), dim3(static_cast<unsigned>((groups + 255) / 256), batch),
                       dim3(256), 0, stream, erp, im_h, im_w, cube, out, erp_stride, out_stride);
  }
  return hipGetLastError();
}

}  // namespace sba
