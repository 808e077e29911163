// Key-point / image coordinate maps of the reference's matchers and of equi2cube (SURVEY.md section 8 rows a-9, f-3,
// f-4): the pieces that turn ERP frames and matcher output into what the BA path consumes.
//   equi2cube            equi2cube::get_* / get_all            (equi2cube.cpp:12-302)
//   crop_rotated_image   spherical_surf::crop_rotated_image    (spherical_surf.cpp:76-108) via rotate_pixel (:48-74)
//   rotate_keypoints     spherical_surf::rotate_keypoint       (spherical_surf.cpp:110-123) via rotate_pixel
//   cube2equi_keypoints  equi2cube_surf::cube2equi_pixel       (equi2cube_surf.cpp:19-76)
//
// All four end in a DECISION taken on an f64 expression built from sin/cos/sqrt/acos/atan2: an integer pixel index by
// truncation (`Vec2i = height * acos(..) / M_PI`), or a float32 by rounding.  The device's libm (ocml) and the
// reference's (glibc) agree to a few ulp, which is invisible unless the f64 value sits within those few ulp of a
// decision boundary -- and at pitch -90 deg, or for any pitch on some image sizes, whole lines of pixels sit EXACTLY on
// integer boundaries (column 3W/4, the equator row), where the result follows the last bit.  Bit-exact results therefore
// cannot come from the device alone.  Every kernel here computes the decision AND whether it is sensitive (the f64 value
// within 1e-6 pixel of an integer -- six orders above the device/host discrepancy -- or the direction within 1e-4 rad
// of a pole, where acos/atan2 amplify errors); the few sensitive outputs (typically 1e-5 .. 1e-4 of all, everything at
// pitch 0) are finished on the host with the SAME source function compiled for the host, i.e. with the C library the
// reference itself runs on.  This file is built with -ffp-contract=off: the reference's x86-64 arithmetic is unfused.
//
// The two image maps depend only on (geometry, H, W), not on pixel values: the source index of every output pixel is
// computed once into a table (cached per device and geometry), and a frame -- or a batch of frames -- is then a pure
// gather: 4 B of table (L2 / MALL resident across the batch) + 3 B gathered + 3 B stored per output pixel.
#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <list>
#include <mutex>
#include <vector>

#include "sba_internal.hpp"

namespace sba {
namespace {

constexpr double kPi = 3.14159265358979323846;
constexpr double kTieTol = 1e-6;       // pixels: distance to the nearest integer below which the host decides
constexpr double kPoleTol = 1e-8;      // sin^2(colatitude) below which the host decides
constexpr double kFloatTol = 1e-11;    // relative: float32 rounding decided on the host when this close to a midpoint

struct Rot3 { double m[9]; };

struct PixelDecision { int row, col; bool sensitive; };

// unit direction -> ERP pixel by truncation (spherical_surf.cpp:62-71, equi2cube.cpp:40-50)
__host__ __device__ inline PixelDecision direction_to_pixel(double w0, double w1, double w2, int width, int height) {
  const double a = acos(w2);
  double b = atan2(w1, w0);
  if (b < 0) b += kPi * 2;
  const double ur = height * a / kPi, uc = width * b / (2 * kPi);
  PixelDecision d;
  d.row = static_cast<int>(ur);
  d.col = static_cast<int>(uc);
  d.sensitive = !(fabs(ur - rint(ur)) >= kTieTol) || !(fabs(uc - rint(uc)) >= kTieTol) || !(1.0 - w2 * w2 >= kPoleTol);
  return d;
}

// rotate_pixel (spherical_surf.cpp:48-74): (row, col) -> sphere -> R -> (row, col)
__host__ __device__ inline PixelDecision rotate_pixel(int row, int col, const Rot3& R, int width, int height) {
  const double r0 = kPi * row / height, r1 = 2 * kPi * col / width;
  const double s0 = sin(r0);
  const double v0 = s0 * cos(r1), v1 = s0 * sin(r1), v2 = cos(r0);
  const double w0 = R.m[0] * v0 + R.m[1] * v1 + R.m[2] * v2;
  const double w1 = R.m[3] * v0 + R.m[4] * v1 + R.m[5] * v2;
  const double w2 = R.m[6] * v0 + R.m[7] * v1 + R.m[8] * v2;
  return direction_to_pixel(w0, w1, w2, width, height);
}

// One output pixel of the cube strip (face order left, front, right, back, top, bottom; equi2cube.cpp:292-298).
__host__ __device__ inline PixelDecision cube_pixel_source(int face, int i, int j, int S, int width, int height) {
  const double s = static_cast<double>(S);
  const double a = (s - 2.0 * j) / s, b = (s - 2.0 * i) / s;      // (cube_size - 2 j) / cube_size, ... i ...
  const double an = (2.0 * j - s) / s, bn = (2.0 * i - s) / s;
  double x, y, z;
  switch (face) {
    case 0: x = a;    y = 1.0;  z = b;    break;  // left   (equi2cube.cpp:118-120)
    case 1: x = -1.0; y = a;    z = b;    break;  // front  (:73-75)
    case 2: x = an;   y = -1.0; z = b;    break;  // right  (:163-165)
    case 3: x = 1.0;  y = an;   z = b;    break;  // back   (:28-30)
    case 4: x = b;    y = a;    z = 1.0;  break;  // top    (:208-210)
    default: x = bn;  y = a;    z = -1.0; break;  // bottom (:253-255)
  }
  const double nrm = sqrt(x * x + y * y + z * z);
  return direction_to_pixel(x / nrm, y / nrm, z / nrm, width, height);
}

// cube2equi_pixel (equi2cube_surf.cpp:19-76): cube-strip pixel (float) -> ERP pixel (float32 of an f64 expression)
struct FloatDecision { float x, y; bool sensitive; };
__host__ __device__ inline bool float_rounding_is_sensitive(double u) {
  const float f = static_cast<float>(u);
  return static_cast<float>(u * (1.0 + kFloatTol)) != f || static_cast<float>(u * (1.0 - kFloatTol)) != f || !(u == u);
}
__host__ __device__ inline FloatDecision cube2equi_pixel(float cx, float cy, int S, int width, int height) {
  const double s = S;
  double x = 0, y = 0, z = 0;
  if (cx < S) { x = (s - 2.0 * cx) / s; y = 1.0; z = (s - 2.0 * cy) / s; }                                       // left
  else if (cx >= S && cx < 2 * S) { x = -1.0; y = (s - 2.0 * (cx - S)) / s; z = (s - 2.0 * cy) / s; }           // front
  else if (cx >= 2 * S && cx < 3 * S) { x = (2.0 * (cx - 2 * S) - s) / s; y = -1.0; z = (s - 2.0 * cy) / s; }   // right
  else if (cx >= 3 * S && cx < 4 * S) { x = 1.0; y = (2.0 * (cx - 3 * S) - s) / s; z = (s - 2.0 * cy) / s; }    // back
  else if (cx >= 4 * S && cx < 5 * S) { x = (s - 2.0 * cy) / s; y = (s - 2.0 * (cx - 4 * S)) / s; z = 1.0; }    // top
  else if (cx >= 5 * S) { x = (2.0 * cy - s) / s; y = (s - 2.0 * (cx - 5 * S)) / s; z = -1.0; }                 // bottom
  const double nrm = sqrt(x * x + y * y + z * z);
  const double uz = z / nrm;
  const double a = acos(uz);
  double b = atan2(y / nrm, x / nrm);
  if (b < 0) b += kPi * 2;
  const double ux = width * b / (2 * kPi), uy = height * a / kPi;
  FloatDecision d;
  d.x = static_cast<float>(ux);
  d.y = static_cast<float>(uy);
  d.sensitive = float_rounding_is_sensitive(ux) || float_rounding_is_sensitive(uy) || !(1.0 - uz * uz >= kPoleTol) ||
                !(fabs(b) >= 1e-9);     // b ~ 0: the sign test above may go either way
  return d;
}

// rotate_keypoint's per-record work (spherical_surf.cpp:116-121): band offset, truncations, rotate_pixel
__host__ __device__ inline PixelDecision rotate_keypoint_record(float px, float py, const Rot3& R, int width, int height) {
  const int offset_i = static_cast<int>(py + static_cast<float>(height * 3 / 8));   // float + int, truncated (.cpp:116)
  return rotate_pixel(offset_i, static_cast<int>(px), R, width, height);
}

// ---- sensitive-output list: device appends, host finishes ----------------------------------------------------------
struct TieList {
  unsigned int* count;      // device counter
  unsigned int* index;      // [capacity] output indices
  unsigned int capacity;
};
__device__ __forceinline__ void tie_append(const TieList& t, unsigned int idx) {
  const unsigned int k = atomicAdd(t.count, 1u);
  if (k < t.capacity) t.index[k] = idx;
}

// ---- source-index tables of the two image maps -----------------------------------------------------------------------
// table[o] = source pixel index of output pixel o, or -1 where the source falls outside the image (crop only).
__global__ __launch_bounds__(256) void equi2cube_table_kernel(int S, int im_h, int im_w, int* __restrict__ table, TieList ties) {
  const size_t o = static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (o >= static_cast<size_t>(S) * 6 * S) return;
  const int i = static_cast<int>(o / (6 * S)), c = static_cast<int>(o % (6 * S));
  const int face = c / S, j = c - face * S;
  const PixelDecision d = cube_pixel_source(face, i, j, S, im_w, im_h);
  // The reference does not clamp (equi2cube.cpp:47-50); only the exact pole could leave the image.
  const int row = min(max(d.row, 0), im_h - 1), col = min(max(d.col, 0), im_w - 1);
  table[o] = row * im_w + col;
  if (d.sensitive) tie_append(ties, static_cast<unsigned int>(o));
}

__global__ __launch_bounds__(256) void crop_table_kernel(int im_h, int im_w, Rot3 R, int* __restrict__ table, TieList ties) {
  const size_t o = static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (o >= static_cast<size_t>(im_h / 4) * im_w) return;
  const int i = static_cast<int>(o / im_w), j = static_cast<int>(o % im_w);
  const PixelDecision d = rotate_pixel(i + im_h * 3 / 8, j, R, im_w, im_h);               // inverse warping, .cpp:91-96
  const bool inside = d.row >= 0 && d.col >= 0 && d.row < im_h && d.col < im_w;           // .cpp:100
  table[o] = inside ? d.row * im_w + d.col : -1;
  if (d.sensitive) tie_append(ties, static_cast<unsigned int>(o));
}

__global__ void table_patch_kernel(int* __restrict__ table, const unsigned int* __restrict__ idx,
                                   const int* __restrict__ val, unsigned int count) {
  const unsigned int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k < count) table[idx[k]] = val[k];
}

// out[f][o] = src[f][table[o]] for every frame f of the block's slice of the batch (table < 0: outside -> 0).
// ONE output pixel per lane: a wave instruction then spans 64 consecutive output pixels, whose sources lie within a
// few hundred bytes -- few cache lines per instruction.  (Measured on 64 and 512 frames 3840x1920 -> S = 600,
// tools/gather_probe.hip, profiles/r02_gather_probe.log: 4 pixels per lane with byte loads 950 GB/s algorithmic, with
// one unaligned dword load per pixel 430-650 GB/s -- misaligned dwords are split by the hardware -- two aligned dwords +
// v_alignbyte 725 GB/s, staging through LDS 930 GB/s, one pixel per lane 1 130-1 150 GB/s.)  Three byte loads per
// pixel; PACK: the four lanes of a quad assemble their 12 bytes into three dword stores (one DPP quad shift) instead of
// twelve byte stores.  Two frames per block (the table entry is loaded once per block and reused).
//
// Block -> work mapping (1-D grid): blocks are dealt to the 8 XCDs round-robin and every XCD has its own L2, so
// physical block b is given logical work item (b % 8) * ceil(total / 8) + b / 8: each XCD walks a CONTIGUOUS run of
// (frame slice, pixel block) items -- whole frames, row after row -- and the source lines that neighbouring output rows
// share are fetched into one L2 instead of up to eight (xcd_aware = 0: plain order, for A/B measurements).
template <bool PACK>
__global__ __launch_bounds__(256) void gather_kernel(const int* __restrict__ table, size_t out_pixels,
                                                     const uint8_t* __restrict__ src, size_t src_stride,
                                                     uint8_t* __restrict__ out, size_t out_stride, int batch,
                                                     int frames_per_block, unsigned blocks_x, unsigned total_items,
                                                     int xcd_aware) {
  unsigned item = blockIdx.x;
  if (xcd_aware) {
    const unsigned per = (total_items + 7u) / 8u;
    item = (blockIdx.x & 7u) * per + (blockIdx.x >> 3);
    if ((blockIdx.x >> 3) >= per || item >= total_items) return;
  } else if (item >= total_items) {
    return;
  }
  const unsigned bx = item % blocks_x, by = item / blocks_x;
  const size_t g = static_cast<size_t>(bx) * blockDim.x + threadIdx.x;
  const bool valid = g < out_pixels;
  const int t = valid ? table[g] : -1;
  const size_t p = t >= 0 ? static_cast<size_t>(t) * 3 : 0;
  const int j = threadIdx.x & 3;
  const int f0 = static_cast<int>(by) * frames_per_block, f1 = min(batch, f0 + frames_per_block);
  for (int f = f0; f < f1; ++f) {
    const uint8_t* s = src + static_cast<size_t>(f) * src_stride;
    uint8_t* d = out + static_cast<size_t>(f) * out_stride;
    uint32_t v = 0;
    if (t >= 0) v = static_cast<uint32_t>(s[p]) | (static_cast<uint32_t>(s[p + 1]) << 8) | (static_cast<uint32_t>(s[p + 2]) << 16);
    if (PACK) {
      // lanes 0..2 of every quad store one dword each: the bytes of pixels j and j + 1 (out_pixels % 4 == 0)
      const uint32_t nx = static_cast<uint32_t>(__builtin_amdgcn_update_dpp(0, static_cast<int>(v), 0xF9 /* quad_perm [1,2,3,3] */,
                                                                             0xf, 0xf, false));
      const uint32_t dw = (v >> (8 * j)) | (nx << (24 - 8 * j));
      if (valid && j < 3) reinterpret_cast<uint32_t*>(d + (g & ~static_cast<size_t>(3)) * 3)[j] = dw;
    } else if (valid) {
      d[g * 3] = static_cast<uint8_t>(v); d[g * 3 + 1] = static_cast<uint8_t>(v >> 8); d[g * 3 + 2] = static_cast<uint8_t>(v >> 16);
    }
  }
}

// ---- tiled gather: the source bytes of a 32 x 32 output tile staged through LDS ------------------------------------
// gather_kernel above is bound by the L1's tag look-ups, not by HBM (rocprofv3, 512 frames 3840x1920 -> S = 600: 0.66
// TCP accesses per cycle and CU plus 16 % tag-conflict stalls, while HBM moves 12.6 MB + 6.5 MB per frame in 9.6 us =
// 2 TB/s): three scattered byte loads per pixel cost ~47 look-ups per wave instruction.  Here a block owns a 32 x 32
// tile of the output and `fpb` frames.  When the table is built the host lists, per tile, the 16-byte chunks of the
// source frame that hold the tile's pixels (exactly those: no bounding box, so seams, poles and face borders need no
// special case) and gives every pixel its byte offset into that list.  The block copies the chunks into LDS with one
// coalesced 16-byte load per lane (one look-up per 64 bytes), then every lane takes the three bytes of its four pixels
// from LDS.  Four pixels per lane = four times the bytes per memory round trip of a block.  Tiles whose chunk list
// exceeds the LDS budget fall back to byte loads from global memory inside the same kernel -- in the same 2-D tile order,
// which by itself is worth most of the gain (the sources of vertically neighbouring output pixels share cache lines:
// 9.83 us per frame for the row-major pixel-per-lane kernel, 7.14-7.36 us for 32 x 16 / 32 x 32 tiles without staging,
// 6.64 us with it).
// A tile is 1024 output pixels, 2^wshift wide (SBA_GATHER_TILE_W = 32, 64 or 128 when a table is built; the tile is then 32, 16
// or 8 rows high).  The memory system fetches 128-byte lines whatever a request asks for (tools/chunk_probe.hip: random
// 64-byte runs stream at 3.2 TB/s, random 128-byte runs at 6.2 TB/s), and a tile row's source segment is ~3-4 bytes per
// output pixel: the wider the tile, the larger the used share of the lines it touches.
constexpr int kTilePx = 1024, kTileLanePx = kTilePx / 256;
constexpr int kTileWDefault = 32;
// bytes per frame and tile.  Measured on 256 frames 3840x1920 -> S = 600, two frames per block (profiles/r02_gather_sweep.log):
// 6 / 8 / 12 / 16 / 24 KiB = 7.35 / 6.64-6.95 / 6.82-7.10 / 8.18 / 8.19 us per frame (31 / 72 / 89 / 94 / 100 % of the
// tiles staged): a larger budget stages more tiles but leaves fewer blocks per CU.
constexpr int kTileLdsBudget = 8192;
// A work item of the tiled gather: a 32 x 32 output tile (4 pixels per lane), or -- with SBA_GATHER_SUBTILES=1, where a
// 32 x 32 tile's chunk list is over budget: the pole centres, where every pixel sits in its own source chunk -- one of its
// four 16 x 16 quarters (1 pixel per lane; at most 2 x 256 chunks = 8 KiB, so a quarter always fits the budget).
// chunks == 0: not staged (list over budget, or reaching past the end of the frame), the pixels then come straight from
// global memory.  The item's list is chunk_list[item * list_stride ...], its LDS offsets lds_offset[item * kTilePx ...].
struct TileHdr { unsigned chunks; unsigned short x0, y0; unsigned char wshift, pxshift; unsigned short pad; };   // width 2^wshift, 2^pxshift pixels
static_assert(sizeof(TileHdr) == 12, "TileHdr layout");

template <bool PACK>
__global__ __launch_bounds__(256) void gather_tiled_kernel(const int* __restrict__ table, const TileHdr* __restrict__ hdr,
                                                           const unsigned* __restrict__ chunk_list, unsigned list_stride,
                                                           const unsigned short* __restrict__ lds_offset, unsigned ntiles,
                                                           int tiles_x, int out_w, int out_h, const uint8_t* __restrict__ src,
                                                           size_t src_stride, uint8_t* __restrict__ out, size_t out_stride,
                                                           int batch, int frames_per_block, int lds_per_frame,
                                                           unsigned total_items, int xcd_aware) {
  extern __shared__ __attribute__((aligned(16))) uint8_t tile_lds[];
  unsigned item = blockIdx.x;
  if (xcd_aware) {                      // as gather_kernel: every XCD walks a contiguous run of (frame group, tile) items
    const unsigned per = (total_items + 7u) / 8u;
    item = (blockIdx.x & 7u) * per + (blockIdx.x >> 3);
    if ((blockIdx.x >> 3) >= per || item >= total_items) return;
  } else if (item >= total_items) {
    return;
  }
  const unsigned tile = item % ntiles, group = item / ntiles;
  const int t = threadIdx.x, j = t & 3;
  // The tile's chunk list sits at a fixed stride, so its entries are requested together with the header instead of
  // after it: one memory round trip less in front of the chunk loads (list_stride <= 512 entries: two per lane).
  const unsigned* list = chunk_list + static_cast<size_t>(tile) * list_stride;
  unsigned entry[2];
#pragma unroll
  for (int m = 0; m < 2; ++m) entry[m] = t + 256u * m < list_stride ? list[t + 256 * m] : 0u;
  const TileHdr h = hdr[tile];
  const int f0 = static_cast<int>(group) * frames_per_block, f1 = min(batch, f0 + frames_per_block);
  const int edge_mask = (1 << h.wshift) - 1, item_px = 1 << h.pxshift;          // 1024 or 256 pixels
  size_t g[kTileLanePx];
  bool active[kTileLanePx];
#pragma unroll
  for (int m = 0; m < kTileLanePx; ++m) {
    const int q = t + 256 * m, orow = h.y0 + (q >> h.wshift), ocol = h.x0 + (q & edge_mask);
    active[m] = q < item_px && ocol < out_w && orow < out_h;
    g[m] = static_cast<size_t>(orow) * out_w + ocol;
  }
  auto store = [&](uint8_t* d, uint32_t v, int m) {
    if (PACK) {          // lanes 0..2 of every quad store one dword each (out_w % 4 == 0: a quad is whole or absent)
      const uint32_t nx = static_cast<uint32_t>(__builtin_amdgcn_update_dpp(0, static_cast<int>(v), 0xF9, 0xf, 0xf, false));
      const uint32_t dw = (v >> (8 * j)) | (nx << (24 - 8 * j));
      if (active[m] && j < 3) reinterpret_cast<uint32_t*>(d + (g[m] & ~static_cast<size_t>(3)) * 3)[j] = dw;
    } else if (active[m]) {
      d[g[m] * 3] = static_cast<uint8_t>(v); d[g[m] * 3 + 1] = static_cast<uint8_t>(v >> 8); d[g[m] * 3 + 2] = static_cast<uint8_t>(v >> 16);
    }
  };
  if (h.chunks == 0) {                  // chunk list over budget: bytes straight from global memory
    size_t p[kTileLanePx];
    bool inside[kTileLanePx];
#pragma unroll
    for (int m = 0; m < kTileLanePx; ++m) {
      const int idx = active[m] ? table[g[m]] : -1;
      inside[m] = idx >= 0;
      p[m] = inside[m] ? static_cast<size_t>(idx) * 3 : 0;
    }
    for (int f = f0; f < f1; ++f) {
      const uint8_t* s = src + static_cast<size_t>(f) * src_stride;
      uint32_t v[kTileLanePx];
#pragma unroll
      for (int m = 0; m < kTileLanePx; ++m)
        v[m] = inside[m] ? static_cast<uint32_t>(s[p[m]]) | (static_cast<uint32_t>(s[p[m] + 1]) << 8) | (static_cast<uint32_t>(s[p[m] + 2]) << 16) : 0u;
#pragma unroll
      for (int m = 0; m < kTileLanePx; ++m) store(out + static_cast<size_t>(f) * out_stride, v[m], m);
    }
    return;
  }
  unsigned off[kTileLanePx];
#pragma unroll
  for (int m = 0; m < kTileLanePx; ++m) off[m] = active[m] ? lds_offset[static_cast<size_t>(tile) * kTilePx + t + 256 * m] : 0xffffu;
#pragma unroll
  for (int m = 0; m < 2; ++m) {
    const unsigned c = t + 256u * m;
    if (c < h.chunks)
      for (int f = f0; f < f1; ++f)
        *reinterpret_cast<uint4*>(tile_lds + (f - f0) * lds_per_frame + c * 16) =
            *reinterpret_cast<const uint4*>(src + static_cast<size_t>(f) * src_stride + entry[m]);
  }
  for (unsigned c = t + 512u; c < h.chunks; c += 256) {          // only with a tuning budget above 8 KiB
    const unsigned byte = list[c];
    for (int f = f0; f < f1; ++f)
      *reinterpret_cast<uint4*>(tile_lds + (f - f0) * lds_per_frame + c * 16) =
          *reinterpret_cast<const uint4*>(src + static_cast<size_t>(f) * src_stride + byte);
  }
  __syncthreads();
  for (int f = f0; f < f1; ++f)
#pragma unroll
    for (int m = 0; m < kTileLanePx; ++m) {
      uint32_t v = 0;
      if (off[m] != 0xffffu) {          // 0xffff: the pixel's source lies outside the image (crop) -> 0
        const uint8_t* l = tile_lds + (f - f0) * lds_per_frame + off[m];
        v = static_cast<uint32_t>(l[0]) | (static_cast<uint32_t>(l[1]) << 8) | (static_cast<uint32_t>(l[2]) << 16);
      }
      store(out + static_cast<size_t>(f) * out_stride, v, m);
    }
}

// ---- key-point maps ---------------------------------------------------------------------------------------------
// In place on records whose first two floats are pt.x, pt.y.  A record with a sensitive decision is left untouched
// and listed; the host finishes it from the original coordinates.
__global__ void rotate_keypoints_kernel(uint8_t* __restrict__ kp, size_t n, size_t stride, Rot3 R, int width,
                                        int height, TieList ties) {
  const size_t i = static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float* rec = reinterpret_cast<float*>(kp + i * stride);
  const PixelDecision d = rotate_keypoint_record(rec[0], rec[1], R, width, height);
  if (d.sensitive) { tie_append(ties, static_cast<unsigned int>(i)); return; }
  rec[0] = static_cast<float>(d.col);       // key.pt.x = vec_pixel[1]  (.cpp:120)
  rec[1] = static_cast<float>(d.row);       // key.pt.y = vec_pixel[0]  (.cpp:121)
}

__global__ void cube2equi_kernel(uint8_t* __restrict__ kp, size_t n, size_t stride, int S, int im_w, int im_h,
                                 TieList ties) {
  const size_t i = static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float* rec = reinterpret_cast<float*>(kp + i * stride);
  const FloatDecision d = cube2equi_pixel(rec[0], rec[1], S, im_w, im_h);
  if (d.sensitive) { tie_append(ties, static_cast<unsigned int>(i)); return; }
  rec[0] = d.x;
  rec[1] = d.y;
}

// listed records <-> a compact (x, y) array
__global__ void records_gather_kernel(const uint8_t* __restrict__ kp, size_t stride, const unsigned int* __restrict__ idx,
                                      unsigned int count, float2* __restrict__ xy) {
  const unsigned int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= count) return;
  const float* rec = reinterpret_cast<const float*>(kp + static_cast<size_t>(idx[k]) * stride);
  xy[k] = make_float2(rec[0], rec[1]);
}
__global__ void records_scatter_kernel(uint8_t* __restrict__ kp, size_t stride, const unsigned int* __restrict__ idx,
                                       unsigned int count, const float2* __restrict__ xy) {
  const unsigned int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= count) return;
  float* rec = reinterpret_cast<float*>(kp + static_cast<size_t>(idx[k]) * stride);
  rec[0] = xy[k].x;
  rec[1] = xy[k].y;
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

// Device-side list + its host read-back.
struct TieBuffers {
  DeviceBuffer count, index;
  TieList list{nullptr, nullptr, 0};
  int alloc(size_t capacity, hipStream_t stream) {
    SBA_TRY_HIP(count.alloc(sizeof(unsigned int)));
    SBA_TRY_HIP(index.alloc(std::max<size_t>(capacity, 1) * sizeof(unsigned int)));
    SBA_TRY_HIP(hipMemsetAsync(count.ptr, 0, sizeof(unsigned int), stream));
    list = TieList{count.as<unsigned int>(), index.as<unsigned int>(), static_cast<unsigned int>(capacity)};
    return SBA_OK;
  }
  // synchronises `stream`
  int read(hipStream_t stream, std::vector<unsigned int>* out) {
    unsigned int c = 0;
    SBA_TRY_HIP(hipMemcpyAsync(&c, count.ptr, sizeof(c), hipMemcpyDeviceToHost, stream));
    SBA_TRY_HIP(hipStreamSynchronize(stream));
    if (c > list.capacity) return set_error(SBA_ERR_HIP, "internal: tie list overflow (%u > %u)", c, list.capacity);
    out->resize(c);
    if (c > 0) {
      SBA_TRY_HIP(hipMemcpyAsync(out->data(), index.ptr, c * sizeof(unsigned int), hipMemcpyDeviceToHost, stream));
      SBA_TRY_HIP(hipStreamSynchronize(stream));
    }
    return SBA_OK;
  }
};

// ---- table cache ----------------------------------------------------------------------------------------------------
enum { TABLE_EQUI2CUBE = 0, TABLE_CROP = 1 };
struct TableKey {
  int device, kind, param, im_h, im_w;     // param: cube size, or the bit pattern of the float pitch
  bool operator==(const TableKey& o) const {
    return device == o.device && kind == o.kind && param == o.param && im_h == o.im_h && im_w == o.im_w;
  }
};
struct Table {
  TableKey key{};
  int* dev = nullptr;
  size_t out_pixels = 0;
  size_t host_decided = 0;    // outputs finished on the host when the table was built
  // tiled form of the same map (gather_tiled_kernel); tile_hdr == nullptr: not available
  int out_w = 0, out_h = 0, tiles_x = 0, lds_per_frame = 0;
  unsigned ntiles = 0, staged_tiles = 0, list_stride = 0;   // 32 x 32 tiles of the output; how many are staged through LDS entirely
  unsigned nitems = 0;                                       // work items: staged 32 x 32 tiles + the 16 x 16 quarters of the others
  TileHdr* tile_hdr = nullptr;
  unsigned* chunk_list = nullptr;
  unsigned short* lds_offset = nullptr;
};
void free_table(Table* t) {
  for (void* p : {static_cast<void*>(t->dev), static_cast<void*>(t->tile_hdr), static_cast<void*>(t->chunk_list),
                  static_cast<void*>(t->lds_offset)})
    if (p) (void)hipFree(p);
  t->dev = nullptr; t->tile_hdr = nullptr; t->chunk_list = nullptr; t->lds_offset = nullptr;
}

// The tiled form of a finished source-index table: per 32 x 32 output tile the sorted list of the 16-byte chunks of the
// source frame that hold its pixels, and per pixel its byte offset into the tile's staged copy of them.  A pixel's three
// bytes may straddle two chunks; both are then in the list and, being neighbours in memory, neighbours in the list.
int build_tiles(Table* t, hipStream_t stream) {
  const int out_w = t->out_w, out_h = t->out_h;
  if (t->out_pixels == 0) return SBA_OK;
  std::vector<int> table(t->out_pixels);
  SBA_TRY_HIP(hipMemcpyAsync(table.data(), t->dev, t->out_pixels * sizeof(int), hipMemcpyDeviceToHost, stream));
  SBA_TRY_HIP(hipStreamSynchronize(stream));
  const size_t frame_bytes = static_cast<size_t>(t->key.im_h) * t->key.im_w * 3;
  size_t budget = kTileLdsBudget;
  if (const char* e = std::getenv("SBA_GATHER_LDS_BUDGET")) {      // tuning only; 16 .. 32768, offsets are 16-bit
    const long v = std::atol(e);
    if (v >= 16 && v <= 32768) budget = static_cast<size_t>(v);
  }
  int tile_w = kTileWDefault;
  if (const char* e = std::getenv("SBA_GATHER_TILE_W")) { const int v = std::atoi(e); if (v == 32 || v == 64 || v == 128) tile_w = v; }
  const int tile_h = kTilePx / tile_w;
  int tile_wshift = 0;
  while ((1 << tile_wshift) < tile_w) ++tile_wshift;
  const int tiles_x = (out_w + tile_w - 1) / tile_w, tiles_y = (out_h + tile_h - 1) / tile_h;
  const size_t ntiles = static_cast<size_t>(tiles_x) * tiles_y;
  const size_t list_stride = budget / 16;              // entries per item: fixed, so that a block finds its list without the header
  // SBA_GATHER_SUBTILES=1: an over-budget tile is split into its four 16 x 16 quarters, each of which always fits the
  // budget, so that EVERY tile is staged.  Built and measured in round 3 (512 frames 3840x1920 -> S = 600, same box, A/B):
  // 3.49 ms per batch with the quarters against 3.37 ms with the whole tiles gathering from global memory -- 3.6 % SLOWER.
  // At the pole centres every pixel sits in its own 16-byte chunk and its own cache line: staging moves the same number
  // of lines through one more hop (LDS) and quadruples the blocks.  Kept as an option, off by default.
  const bool subtiles = [] { const char* e = std::getenv("SBA_GATHER_SUBTILES"); return e && e[0] == '1'; }();
  const bool whole_lines = [] { const char* e = std::getenv("SBA_GATHER_LINES"); return e && e[0] == '1'; }();
  std::vector<TileHdr> hdr;
  std::vector<unsigned> chunk_list;
  std::vector<unsigned short> lds_offset;
  std::vector<unsigned> ids;
  unsigned lds_max = 0;
  // One work item: the 2^pxshift pixels of the rectangle 2^wshift wide at (x0, y0).  Returns false -- and appends nothing --
  // when its chunk list is over budget and `must` is not set; with `must` an over-budget (or past-the-end) item is
  // appended un-staged.
  auto add_item = [&](int x0, int y0, int shift, int pxshift, bool must) -> bool {
    const int edge = 1 << shift, px = 1 << pxshift;
    ids.clear();
    for (int q = 0; q < px; ++q) {
      const int orow = y0 + (q >> shift), ocol = x0 + (q & (edge - 1));
      if (orow >= out_h || ocol >= out_w) continue;
      const int idx = table[static_cast<size_t>(orow) * out_w + ocol];
      if (idx < 0) continue;
      const size_t o = static_cast<size_t>(idx) * 3;
      ids.push_back(static_cast<unsigned>(o >> 4));
      if (((o + 2) >> 4) != (o >> 4)) ids.push_back(static_cast<unsigned>((o + 2) >> 4));
    }
    std::sort(ids.begin(), ids.end());
    ids.erase(std::unique(ids.begin(), ids.end()), ids.end());
    if (whole_lines) {      // SBA_GATHER_LINES=1: every 128-byte line a chunk of the list lies in, whole (8 chunks, clipped to the frame)
      std::vector<unsigned> lines;
      for (unsigned c : ids) if (lines.empty() || lines.back() != (c >> 3)) lines.push_back(c >> 3);
      ids.clear();
      for (unsigned l : lines)
        for (unsigned k = 0; k < 8; ++k)
          if ((static_cast<size_t>(l) * 8 + k + 1) * 16 <= frame_bytes) ids.push_back(l * 8 + k);
    }
    // stage unless the list is over budget, empty (nothing to read), or its last chunk reaches past the frame (the last
    // frame of a batch must not be read beyond its end)
    const bool stage = !ids.empty() && ids.size() * 16 <= budget && (static_cast<size_t>(ids.back()) + 1) * 16 <= frame_bytes;
    if (!stage && !must && ids.size() * 16 > budget) return false;
    const size_t item = hdr.size();
    hdr.push_back(TileHdr{stage ? static_cast<unsigned>(ids.size()) : 0u, static_cast<unsigned short>(x0), static_cast<unsigned short>(y0),
                          static_cast<unsigned char>(shift), static_cast<unsigned char>(pxshift), 0});
    chunk_list.resize((item + 1) * list_stride, 0u);
    lds_offset.resize((item + 1) * kTilePx, 0xffff);
    if (!stage) return true;
    for (size_t k = 0; k < ids.size(); ++k) chunk_list[item * list_stride + k] = ids[k] * 16u;
    for (int q = 0; q < px; ++q) {
      const int orow = y0 + (q >> shift), ocol = x0 + (q & (edge - 1));
      if (orow >= out_h || ocol >= out_w) continue;
      const int idx = table[static_cast<size_t>(orow) * out_w + ocol];
      if (idx < 0) continue;
      const size_t o = static_cast<size_t>(idx) * 3;
      const size_t rank = static_cast<size_t>(std::lower_bound(ids.begin(), ids.end(), static_cast<unsigned>(o >> 4)) - ids.begin());
      lds_offset[item * kTilePx + q] = static_cast<unsigned short>(rank * 16 + (o & 15));
    }
    lds_max = std::max<unsigned>(lds_max, static_cast<unsigned>(ids.size() * 16));
    return true;
  };
  if (out_w > 0xffff || out_h > 0xffff) return SBA_OK;           // item origins are 16-bit: such outputs keep the untiled kernel
  unsigned staged = 0;
  for (size_t tile = 0; tile < ntiles; ++tile) {
    const int x0 = static_cast<int>(tile % tiles_x) * tile_w, y0 = static_cast<int>(tile / tiles_x) * tile_h;
    const size_t first = hdr.size();
    if (!add_item(x0, y0, tile_wshift, 10, !subtiles))
      for (int sub = 0; sub < 4; ++sub) {    // over budget: its four quarters (half as wide, half as high), each at most 2 x 256 chunks
        const int sx = x0 + (sub & 1) * (tile_w / 2), sy = y0 + (sub >> 1) * (tile_h / 2);
        if (sx < out_w && sy < out_h) (void)add_item(sx, sy, tile_wshift - 1, 8, true);
      }
    bool all = true;
    for (size_t k = first; k < hdr.size(); ++k) all = all && hdr[k].chunks > 0;
    if (all) ++staged;
  }
  unsigned staged_items = 0;
  for (const TileHdr& h : hdr) staged_items += h.chunks > 0 ? 1u : 0u;
  t->tiles_x = tiles_x; t->ntiles = static_cast<unsigned>(ntiles); t->staged_tiles = staged; t->lds_per_frame = static_cast<int>(lds_max);
  t->nitems = static_cast<unsigned>(hdr.size());
  t->list_stride = static_cast<unsigned>(list_stride);
  if (staged_items == 0) return SBA_OK;
  SBA_TRY_HIP(hipMalloc(reinterpret_cast<void**>(&t->tile_hdr), hdr.size() * sizeof(TileHdr)));
  SBA_TRY_HIP(hipMalloc(reinterpret_cast<void**>(&t->chunk_list), std::max<size_t>(chunk_list.size(), 1) * sizeof(unsigned)));
  SBA_TRY_HIP(hipMalloc(reinterpret_cast<void**>(&t->lds_offset), lds_offset.size() * sizeof(unsigned short)));
  SBA_TRY_HIP(hipMemcpyAsync(t->tile_hdr, hdr.data(), hdr.size() * sizeof(TileHdr), hipMemcpyHostToDevice, stream));
  SBA_TRY_HIP(hipMemcpyAsync(t->chunk_list, chunk_list.data(), chunk_list.size() * sizeof(unsigned), hipMemcpyHostToDevice, stream));
  SBA_TRY_HIP(hipMemcpyAsync(t->lds_offset, lds_offset.data(), lds_offset.size() * sizeof(unsigned short), hipMemcpyHostToDevice, stream));
  SBA_TRY_HIP(hipStreamSynchronize(stream));        // the host vectors go out of scope
  return SBA_OK;
}
std::mutex g_table_mutex;
std::list<Table> g_tables;          // most recently used first; a handful of geometries per process
constexpr size_t kMaxTables = 12;

int build_table(const TableKey& key, float pitch_deg, hipStream_t stream, Table* t) {
  t->key = key;
  const int S = key.param;
  t->out_h = key.kind == TABLE_EQUI2CUBE ? S : key.im_h / 4;
  t->out_w = key.kind == TABLE_EQUI2CUBE ? 6 * S : key.im_w;
  t->out_pixels = static_cast<size_t>(t->out_h) * t->out_w;
  const size_t padded = (t->out_pixels + 3) / 4 * 4;
  SBA_TRY_HIP(hipMalloc(reinterpret_cast<void**>(&t->dev), std::max<size_t>(padded, 4) * sizeof(int)));
  struct Release { Table* t; bool armed = true; ~Release() { if (armed) free_table(t); } } release{t};
  SBA_TRY_HIP(hipMemsetAsync(t->dev, 0xff, std::max<size_t>(padded, 4) * sizeof(int), stream));   // padding = -1
  TieBuffers ties;
  int rc = ties.alloc(t->out_pixels, stream);
  if (rc) return rc;
  const Rot3 R = pitch_rotation(pitch_deg);
  const unsigned grid = static_cast<unsigned>((t->out_pixels + 255) / 256);
  if (t->out_pixels > 0) {
    if (key.kind == TABLE_EQUI2CUBE)
      hipLaunchKernelGGL(equi2cube_table_kernel, dim3(grid), dim3(256), 0, stream, S, key.im_h, key.im_w, t->dev, ties.list);
    else
      hipLaunchKernelGGL(crop_table_kernel, dim3(grid), dim3(256), 0, stream, key.im_h, key.im_w, R, t->dev, ties.list);
    SBA_TRY_HIP(hipGetLastError());
  }
  std::vector<unsigned int> idx;
  rc = ties.read(stream, &idx);
  if (rc) return rc;
  t->host_decided = idx.size();
  if (!idx.empty()) {
    // the sensitive decisions, taken by the same function compiled for the host (glibc libm, like the reference)
    std::vector<int> val(idx.size());
    for (size_t k = 0; k < idx.size(); ++k) {
      const unsigned int o = idx[k];
      if (key.kind == TABLE_EQUI2CUBE) {
        const int i = static_cast<int>(o / (6u * S)), c = static_cast<int>(o % (6u * S));
        const PixelDecision d = cube_pixel_source(c / S, i, c % S, S, key.im_w, key.im_h);
        const int row = std::min(std::max(d.row, 0), key.im_h - 1), col = std::min(std::max(d.col, 0), key.im_w - 1);
        val[k] = row * key.im_w + col;
      } else {
        const int i = static_cast<int>(o / key.im_w), j = static_cast<int>(o % key.im_w);
        const PixelDecision d = rotate_pixel(i + key.im_h * 3 / 8, j, R, key.im_w, key.im_h);
        const bool inside = d.row >= 0 && d.col >= 0 && d.row < key.im_h && d.col < key.im_w;
        val[k] = inside ? d.row * key.im_w + d.col : -1;
      }
    }
    DeviceBuffer val_dev;
    SBA_TRY_HIP(val_dev.alloc(val.size() * sizeof(int)));
    SBA_TRY_HIP(hipMemcpyAsync(val_dev.ptr, val.data(), val.size() * sizeof(int), hipMemcpyHostToDevice, stream));
    hipLaunchKernelGGL(table_patch_kernel, dim3(static_cast<unsigned>((idx.size() + 255) / 256)), dim3(256), 0, stream,
                       t->dev, ties.list.index, val_dev.as<int>(), static_cast<unsigned int>(idx.size()));
    SBA_TRY_HIP(hipGetLastError());
    SBA_TRY_HIP(hipStreamSynchronize(stream));
  }
  rc = build_tiles(t, stream);
  if (rc) return rc;
  release.armed = false;
  return SBA_OK;
}

// The table for `key` (built on first use; the build synchronises `stream`).  The returned pointer stays valid until
// kMaxTables other geometries have been used on top of it.
int get_table(const TableKey& key, float pitch_deg, hipStream_t stream, Table* out) {
  std::lock_guard<std::mutex> lock(g_table_mutex);
  for (auto it = g_tables.begin(); it != g_tables.end(); ++it)
    if (it->key == key) {
      g_tables.splice(g_tables.begin(), g_tables, it);
      *out = g_tables.front();
      return SBA_OK;
    }
  Table t;
  const int rc = build_table(key, pitch_deg, stream, &t);
  if (rc) return rc;
  g_tables.push_front(t);
  while (g_tables.size() > kMaxTables) {
    // a table may still be read by a gather enqueued on some stream of its device: drain before freeing
    (void)hipSetDevice(g_tables.back().key.device);
    (void)hipDeviceSynchronize();
    free_table(&g_tables.back());
    g_tables.pop_back();
    (void)hipSetDevice(key.device);
  }
  *out = g_tables.front();
  return SBA_OK;
}

int launch_gather(const Table& t, const uint8_t* src, int src_pixels, int batch, uint8_t* out, hipStream_t stream) {
  if (t.out_pixels == 0 || batch <= 0) return SBA_OK;
  const size_t src_stride = static_cast<size_t>(src_pixels) * 3, out_stride = t.out_pixels * 3;
  const unsigned gx = static_cast<unsigned>((t.out_pixels + 255) / 256);
  int fpb = batch >= 2 ? 2 : 1;       // measured best: 1 / 2 / 4 / 8 frames per block = 1 080 / 1 150 / 1 130 / 1 045 GB/s
  if (const char* e = std::getenv("SBA_GATHER_FPB")) { const int v = std::atoi(e); if (v >= 1 && v <= 4) fpb = std::min(v, batch); }   // tuning only
  const unsigned gy = static_cast<unsigned>((batch + fpb - 1) / fpb);
  const unsigned long long total = static_cast<unsigned long long>(gx) * gy;
  if (total > 0x7fffff00ull) return set_error(SBA_ERR_INVALID_ARG, "batch of %d frames is too large for one launch", batch);
  static const int xcd_aware = [] { const char* e = std::getenv("SBA_GATHER_XCD"); return e && e[0] == '0' ? 0 : 1; }();
  // The tiled kernel reads 16-byte chunks at frame + 16 k: frames must start on 16-byte boundaries.  SBA_GATHER_TILED=0
  // keeps the pixel-per-lane kernel (A/B measurements).
  const char* tiled_env = std::getenv("SBA_GATHER_TILED");
  const bool aligned = reinterpret_cast<uintptr_t>(src) % 16 == 0 && (batch == 1 || src_stride % 16 == 0);
  if (t.tile_hdr && t.nitems > 0 && aligned && !(tiled_env && tiled_env[0] == '0')) {
    const unsigned long long items = static_cast<unsigned long long>(t.nitems) * gy;
    if (items > 0x7fffff00ull) return set_error(SBA_ERR_INVALID_ARG, "batch of %d frames is too large for one launch", batch);
    const unsigned tgrid = static_cast<unsigned>((items + 7ull) / 8ull * 8ull);
    const size_t lds = static_cast<size_t>(t.lds_per_frame) * fpb;
    if (t.out_w % 4 == 0 && reinterpret_cast<uintptr_t>(out) % 4 == 0 && out_stride % 4 == 0)
      hipLaunchKernelGGL((gather_tiled_kernel<true>), dim3(tgrid), dim3(256), lds, stream, t.dev, t.tile_hdr, t.chunk_list, t.list_stride, t.lds_offset,
                         t.nitems, t.tiles_x, t.out_w, t.out_h, src, src_stride, out, out_stride, batch, fpb, t.lds_per_frame,
                         static_cast<unsigned>(items), xcd_aware);
    else
      hipLaunchKernelGGL((gather_tiled_kernel<false>), dim3(tgrid), dim3(256), lds, stream, t.dev, t.tile_hdr, t.chunk_list, t.list_stride, t.lds_offset,
                         t.nitems, t.tiles_x, t.out_w, t.out_h, src, src_stride, out, out_stride, batch, fpb, t.lds_per_frame,
                         static_cast<unsigned>(items), xcd_aware);
    SBA_TRY_HIP(hipGetLastError());
    return SBA_OK;
  }
  const unsigned grid = static_cast<unsigned>((total + 7ull) / 8ull * 8ull);     // whole rounds of the 8 XCDs
  if (t.out_pixels % 4 == 0 && reinterpret_cast<uintptr_t>(out) % 4 == 0)
    hipLaunchKernelGGL((gather_kernel<true>), dim3(grid), dim3(256), 0, stream, t.dev, t.out_pixels, src, src_stride, out,
                       out_stride, batch, fpb, gx, static_cast<unsigned>(total), xcd_aware);
  else
    hipLaunchKernelGGL((gather_kernel<false>), dim3(grid), dim3(256), 0, stream, t.dev, t.out_pixels, src, src_stride, out,
                       out_stride, batch, fpb, gx, static_cast<unsigned>(total), xcd_aware);
  SBA_TRY_HIP(hipGetLastError());
  return SBA_OK;
}

int check_image(int im_height, int im_width) {
  if (im_height <= 0 || im_width <= 0) return set_error(SBA_ERR_INVALID_ARG, "bad image size");
  if (static_cast<long long>(im_height) * im_width > 0x7fffffffLL / 3)
    return set_error(SBA_ERR_INVALID_ARG, "image too large for 32-bit pixel indices");
  return SBA_OK;
}

// Key-point records on the device, in place: launch, read the list of sensitive records, finish those on the host.
template <typename Launch, typename HostDecide>
int keypoints_in_place(hipStream_t stream, uint8_t* kp_dev, size_t n, size_t stride, Launch&& launch, HostDecide&& decide) {
  if (n == 0) return SBA_OK;
  if (n > 0xffffffffull) return set_error(SBA_ERR_INVALID_ARG, "too many key-points");
  TieBuffers ties;
  int rc = ties.alloc(n, stream);
  if (rc) return rc;
  launch(ties.list);
  SBA_TRY_HIP(hipGetLastError());
  std::vector<unsigned int> idx;
  rc = ties.read(stream, &idx);
  if (rc || idx.empty()) return rc;
  const unsigned int count = static_cast<unsigned int>(idx.size());
  const unsigned grid = (count + 255) / 256;
  DeviceBuffer xy_dev;
  SBA_TRY_HIP(xy_dev.alloc(count * sizeof(float2)));
  hipLaunchKernelGGL(records_gather_kernel, dim3(grid), dim3(256), 0, stream, kp_dev, stride, ties.list.index, count, xy_dev.as<float2>());
  SBA_TRY_HIP(hipGetLastError());
  std::vector<float2> xy(count);
  SBA_TRY_HIP(hipMemcpyAsync(xy.data(), xy_dev.ptr, count * sizeof(float2), hipMemcpyDeviceToHost, stream));
  SBA_TRY_HIP(hipStreamSynchronize(stream));
  for (float2& q : xy) decide(&q.x, &q.y);
  SBA_TRY_HIP(hipMemcpyAsync(xy_dev.ptr, xy.data(), count * sizeof(float2), hipMemcpyHostToDevice, stream));
  hipLaunchKernelGGL(records_scatter_kernel, dim3(grid), dim3(256), 0, stream, kp_dev, stride, ties.list.index, count, xy_dev.as<float2>());
  SBA_TRY_HIP(hipGetLastError());
  SBA_TRY_HIP(hipStreamSynchronize(stream));     // xy (host) and the scratch buffers go out of scope
  return SBA_OK;
}

int check_records(const void* keypoints, size_t n, size_t stride_bytes) {
  if (n > 0 && !keypoints) return set_error(SBA_ERR_INVALID_ARG, "null key-point array");
  if (stride_bytes < 8 || stride_bytes % 4 != 0) return set_error(SBA_ERR_INVALID_ARG, "stride_bytes must be a multiple of 4 and >= 8");
  return SBA_OK;
}

// Host arrays: round trip of `bytes` through the device around `work(dev)`.
template <typename Work>
int with_device_copy(int device, void* host, size_t bytes, Work&& work) {
  int rc = require_device(device);
  if (rc) return rc;
  if (bytes == 0) return SBA_OK;
  DeviceBuffer dev;
  SBA_TRY_HIP(dev.alloc(bytes));
  SBA_TRY_HIP(hipMemcpy(dev.ptr, host, bytes, hipMemcpyHostToDevice));
  rc = work(dev.as<uint8_t>());
  if (rc) return rc;
  SBA_TRY_HIP(hipMemcpy(host, dev.ptr, bytes, hipMemcpyDeviceToHost));
  return SBA_OK;
}

}  // namespace
}  // namespace sba

extern "C" {

// ---- key-point maps ---------------------------------------------------------------------------------------------
int sba_rotate_keypoints_device(int device, void* stream, void* keypoints_dev, size_t n, size_t stride_bytes,
                                float pitch_deg, int im_width, int im_height) {
  int rc = sba::check_records(keypoints_dev, n, stride_bytes);
  if (rc) return rc;
  if (im_width <= 0 || im_height <= 0) return sba::set_error(SBA_ERR_INVALID_ARG, "bad image size");
  rc = sba::require_device(device);
  if (rc) return rc;
  const sba::Rot3 R = sba::pitch_rotation(pitch_deg);
  hipStream_t st = static_cast<hipStream_t>(stream);
  uint8_t* kp = static_cast<uint8_t*>(keypoints_dev);
  return sba::keypoints_in_place(st, kp, n, stride_bytes,
      [&](const sba::TieList& ties) {
        hipLaunchKernelGGL(sba::rotate_keypoints_kernel, dim3(static_cast<unsigned>((n + 255) / 256)), dim3(256), 0, st, kp, n,
                           stride_bytes, R, im_width, im_height, ties);
      },
      [&](float* x, float* y) {
        const sba::PixelDecision d = sba::rotate_keypoint_record(*x, *y, R, im_width, im_height);
        *x = static_cast<float>(d.col);
        *y = static_cast<float>(d.row);
      });
}

int sba_rotate_keypoints(int device, void* keypoints, size_t n, size_t stride_bytes, float pitch_deg, int im_width,
                         int im_height) {
  int rc = sba::check_records(keypoints, n, stride_bytes);
  if (rc) return rc;
  if (im_width <= 0 || im_height <= 0) return sba::set_error(SBA_ERR_INVALID_ARG, "bad image size");
  return sba::with_device_copy(device, keypoints, n * stride_bytes, [&](uint8_t* dev) {
    return sba_rotate_keypoints_device(device, nullptr, dev, n, stride_bytes, pitch_deg, im_width, im_height);
  });
}

int sba_cube2equi_keypoints_device(int device, void* stream, void* keypoints_dev, size_t n, size_t stride_bytes,
                                   int cube_size, int im_width, int im_height) {
  int rc = sba::check_records(keypoints_dev, n, stride_bytes);
  if (rc) return rc;
  if (im_width <= 0 || im_height <= 0 || cube_size <= 0) return sba::set_error(SBA_ERR_INVALID_ARG, "bad image / cube size");
  rc = sba::require_device(device);
  if (rc) return rc;
  hipStream_t st = static_cast<hipStream_t>(stream);
  uint8_t* kp = static_cast<uint8_t*>(keypoints_dev);
  return sba::keypoints_in_place(st, kp, n, stride_bytes,
      [&](const sba::TieList& ties) {
        hipLaunchKernelGGL(sba::cube2equi_kernel, dim3(static_cast<unsigned>((n + 255) / 256)), dim3(256), 0, st, kp, n,
                           stride_bytes, cube_size, im_width, im_height, ties);
      },
      [&](float* x, float* y) {
        const sba::FloatDecision d = sba::cube2equi_pixel(*x, *y, cube_size, im_width, im_height);
        *x = d.x;
        *y = d.y;
      });
}

int sba_cube2equi_keypoints(int device, void* keypoints, size_t n, size_t stride_bytes, int cube_size, int im_width,
                            int im_height) {
  int rc = sba::check_records(keypoints, n, stride_bytes);
  if (rc) return rc;
  if (im_width <= 0 || im_height <= 0 || cube_size <= 0) return sba::set_error(SBA_ERR_INVALID_ARG, "bad image / cube size");
  return sba::with_device_copy(device, keypoints, n * stride_bytes, [&](uint8_t* dev) {
    return sba_cube2equi_keypoints_device(device, nullptr, dev, n, stride_bytes, cube_size, im_width, im_height);
  });
}

// ---- image maps ---------------------------------------------------------------------------------------------------
int sba_crop_rotated_image_device(int device, void* stream, const void* erp_dev, int im_height, int im_width,
                                  float pitch_deg, int batch, void* out_dev) {
  if (!erp_dev || !out_dev) return sba::set_error(SBA_ERR_INVALID_ARG, "null image pointer");
  if (im_height < 4 || batch <= 0) return sba::set_error(SBA_ERR_INVALID_ARG, "bad image size / batch");
  int rc = sba::check_image(im_height, im_width);
  if (rc) return rc;
  rc = sba::require_device(device);
  if (rc) return rc;
  int pitch_bits;
  std::memcpy(&pitch_bits, &pitch_deg, sizeof(pitch_bits));
  sba::Table t;
  rc = sba::get_table(sba::TableKey{device, sba::TABLE_CROP, pitch_bits, im_height, im_width}, pitch_deg,
                      static_cast<hipStream_t>(stream), &t);
  if (rc) return rc;
  return sba::launch_gather(t, static_cast<const uint8_t*>(erp_dev), im_height * im_width, batch,
                            static_cast<uint8_t*>(out_dev), static_cast<hipStream_t>(stream));
}

int sba_crop_rotated_image(int device, const uint8_t* erp, int im_height, int im_width, float pitch_deg,
                           uint8_t* out) {
  if (!erp || !out) return sba::set_error(SBA_ERR_INVALID_ARG, "null image pointer");
  if (im_height < 4) return sba::set_error(SBA_ERR_INVALID_ARG, "bad image size");
  int rc = sba::check_image(im_height, im_width);
  if (rc) return rc;
  rc = sba::require_device(device);
  if (rc) return rc;
  const size_t in_bytes = static_cast<size_t>(im_height) * im_width * 3;
  const size_t out_bytes = static_cast<size_t>(im_height / 4) * im_width * 3;
  sba::DeviceBuffer in_dev, out_dev;
  SBA_TRY_HIP(in_dev.alloc(in_bytes));
  SBA_TRY_HIP(out_dev.alloc(out_bytes));
  SBA_TRY_HIP(hipMemcpy(in_dev.ptr, erp, in_bytes, hipMemcpyHostToDevice));
  rc = sba_crop_rotated_image_device(device, nullptr, in_dev.ptr, im_height, im_width, pitch_deg, 1, out_dev.ptr);
  if (rc) return rc;
  SBA_TRY_HIP(hipMemcpy(out, out_dev.ptr, out_bytes, hipMemcpyDeviceToHost));   // synchronises with the null stream
  return SBA_OK;
}

int sba_equi2cube_device(int device, void* stream, const void* erp_dev, int im_height, int im_width,
                         int cube_size, int batch, void* out_dev) {
  if (!erp_dev || !out_dev) return sba::set_error(SBA_ERR_INVALID_ARG, "null image pointer");
  if (cube_size <= 0 || batch <= 0 || cube_size > 16384) return sba::set_error(SBA_ERR_INVALID_ARG, "bad cube size / batch");
  int rc = sba::check_image(im_height, im_width);
  if (rc) return rc;
  rc = sba::require_device(device);
  if (rc) return rc;
  sba::Table t;
  rc = sba::get_table(sba::TableKey{device, sba::TABLE_EQUI2CUBE, cube_size, im_height, im_width}, 0.0f,
                      static_cast<hipStream_t>(stream), &t);
  if (rc) return rc;
  return sba::launch_gather(t, static_cast<const uint8_t*>(erp_dev), im_height * im_width, batch,
                            static_cast<uint8_t*>(out_dev), static_cast<hipStream_t>(stream));
}

int sba_equi2cube(int device, const uint8_t* erp, int im_height, int im_width, int cube_size,
                  uint8_t* out) {
  if (!erp || !out) return sba::set_error(SBA_ERR_INVALID_ARG, "null image pointer");
  if (cube_size <= 0) return sba::set_error(SBA_ERR_INVALID_ARG, "bad cube size");
  int rc = sba::check_image(im_height, im_width);
  if (rc) return rc;
  rc = sba::require_device(device);
  if (rc) return rc;
  const size_t in_bytes = static_cast<size_t>(im_height) * im_width * 3;
  const size_t out_bytes = static_cast<size_t>(cube_size) * 6 * cube_size * 3;
  sba::DeviceBuffer in_dev, out_dev;
  SBA_TRY_HIP(in_dev.alloc(in_bytes));
  SBA_TRY_HIP(out_dev.alloc(out_bytes));
  SBA_TRY_HIP(hipMemcpy(in_dev.ptr, erp, in_bytes, hipMemcpyHostToDevice));
  rc = sba_equi2cube_device(device, nullptr, in_dev.ptr, im_height, im_width, cube_size, 1, out_dev.ptr);
  if (rc) return rc;
  SBA_TRY_HIP(hipMemcpy(out, out_dev.ptr, out_bytes, hipMemcpyDeviceToHost));   // synchronises with the null stream
  return SBA_OK;
}

// How many outputs of the cached table for this geometry were decided on the host (diagnostics / tests); -1 if the
// table has not been built.  kind: 0 = equi2cube (param = cube size), 1 = crop (param = bit pattern of the float pitch).
long sba_map_table_host_decided(int device, int kind, int param, int im_height, int im_width) {
  std::lock_guard<std::mutex> lock(sba::g_table_mutex);
  for (const sba::Table& t : sba::g_tables)
    if (t.key == sba::TableKey{device, kind, param, im_height, im_width}) return static_cast<long>(t.host_decided);
  return -1;
}

// The tiled form of that table (diagnostics / tests): number of 32 x 32 output tiles, how many of them are staged through
// LDS, and the LDS bytes per frame the gather allocates.  Returns 0, or -1 if the table has not been built.
int sba_map_table_tiles(int device, int kind, int param, int im_height, int im_width, int* tiles, int* staged_tiles,
                        int* lds_bytes_per_frame) {
  std::lock_guard<std::mutex> lock(sba::g_table_mutex);
  for (const sba::Table& t : sba::g_tables)
    if (t.key == sba::TableKey{device, kind, param, im_height, im_width}) {
      if (tiles) *tiles = static_cast<int>(t.ntiles);
      if (staged_tiles) *staged_tiles = static_cast<int>(t.staged_tiles);
      if (lds_bytes_per_frame) *lds_bytes_per_frame = t.lds_per_frame;
      return 0;
    }
  return -1;
}

}  // extern "C"
