// Device-side address maps of the match planes (included by .hip files only).
// A kernel's per-lane loop runs over LOGICAL pair-of-matches indices pr = 0, 1, ... of one problem; the map turns pr into
// the index of that pair of matches inside the planes: identity for a single problem, the batch's pair layout (PairDesc,
// sba_device.hpp: contiguous, or interleaved in 4 KiB tiles) for a pair of a batch.
#pragma once
#include "sba_device.hpp"

namespace sba {

struct IdentityMap { __device__ __forceinline__ size_t operator()(size_t pr) const { return pr; } };
template <typename ST> struct BatchPairMap;
template <> struct BatchPairMap<double> {      // f64 planes: a 16-byte vector holds one pair of matches
  PairDesc d;
  __device__ __forceinline__ size_t operator()(size_t pr) const { return pair_vector(d, pr); }
};
template <> struct BatchPairMap<float> {       // f32 coordinate planes: a vector holds two pairs; the f64 depth planes follow it
  PairDesc d;
  __device__ __forceinline__ size_t operator()(size_t pr) const { return 2 * pair_vector(d, pr >> 1) + (pr & 1); }
};

}  // namespace sba
