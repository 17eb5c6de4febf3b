// Tile binning for gfx950: builds, for every 16x16 tile, the front-to-back list
// of Gaussians whose tile rectangle covers it.
//
// The published pipeline sorts N (tile << 32 | depth) 64-bit keys, N = number of
// (Gaussian, tile) instances, ~15-20x the Gaussian count: 6+ radix passes over
// 12-byte pairs.  All of it is HBM traffic.  Here the same ordering is reached
// with far fewer bytes:
//   1. stable radix sort of the P Gaussians by their 32-bit depth key
//      (positive floats order like their bit patterns);
//   2. prefix sum of tiles_touched in that depth order;
//   3. every Gaussian, visited in depth order, emits (tile id, Gaussian index)
//      for the tiles of its rectangle, row-major;
//   4. stable radix sort of the instances by tile id alone (13 bits at 1080p:
//      two 8-bit passes over 8-byte pairs);
//   5. tile ranges from the boundaries of the sorted tile ids.
// Stability of 1 and 4 makes the result identical, entry for entry, to a stable
// sort of the 64-bit keys emitted in ascending Gaussian index: equal (tile,
// depth) pairs stay in ascending Gaussian index.
//
// The two sorts and the scan are rocPRIM device primitives; the emit and range
// kernels are written here.
#include <cstring>

#include <rocprim/rocprim.hpp>

#include "gsr_common.h"

namespace {

constexpr int kBlock = 256;

struct TilesOf {
  const uint32_t* tiles_touched;
  __device__ __forceinline__ uint32_t operator()(uint32_t g) const { return tiles_touched[g]; }
};

inline unsigned bits_for(unsigned n) {  // smallest b with 2^b >= n, at least 1
  unsigned b = 1;
  while ((1ull << b) < n) b++;
  return b;
}

// One lane per Gaussian in depth order; a lane with many tiles would serialise
// its wave, so rectangles of more than kCoop tiles are written by the whole
// wave together: the owner's parameters are broadcast with readlane and the 64
// lanes stride over the rectangle.
constexpr int kCoop = 32;

__global__ void __launch_bounds__(kBlock) emit_instances_kernel(int P, int gx, const uint32_t* __restrict__ records,
                                                                const uint32_t* __restrict__ order,
                                                                const uint32_t* __restrict__ offsets,
                                                                uint32_t* __restrict__ tile_ids,
                                                                uint32_t* __restrict__ gauss_ids) {
  const int j = blockIdx.x * kBlock + threadIdx.x;
  const int lane = threadIdx.x & 63;
  uint32_t g = 0, begin = 0, n = 0, rmin = 0, rmax = 0;
  if (j < P) {
    g = order[j];
    const uint32_t end = offsets[j];
    begin = j == 0 ? 0u : offsets[j - 1];
    n = end - begin;
    if (n) {
      const uint2 rr = *reinterpret_cast<const uint2*>(records + (size_t)g * GSR_REC_WORDS + GSR_REC_RECT_MIN);
      rmin = rr.x; rmax = rr.y;
    }
  }
  // small rectangles: each lane writes its own
  if (n && n <= kCoop) {
    const uint32_t x0 = rmin & 0xffffu, y0 = rmin >> 16, x1 = rmax & 0xffffu, y1 = rmax >> 16;
    uint32_t off = begin;
    for (uint32_t y = y0; y < y1; y++)
      for (uint32_t x = x0; x < x1; x++) {
        tile_ids[off] = y * gx + x;
        gauss_ids[off] = g;
        off++;
      }
  }
  // large rectangles: the wave writes them one after the other
  unsigned long long big = __ballot(n > kCoop);
  while (big) {
    const int src = __ffsll((long long)big) - 1;
    big &= big - 1;
    const uint32_t bg = __builtin_amdgcn_readlane(g, src);
    const uint32_t bb = __builtin_amdgcn_readlane(begin, src);
    const uint32_t bn = __builtin_amdgcn_readlane(n, src);
    const uint32_t bmin = __builtin_amdgcn_readlane(rmin, src);
    const uint32_t bmax = __builtin_amdgcn_readlane(rmax, src);
    const uint32_t x0 = bmin & 0xffffu, y0 = bmin >> 16, w = (bmax & 0xffffu) - x0;
    for (uint32_t t = lane; t < bn; t += 64) {
      const uint32_t y = y0 + t / w, x = x0 + t % w;
      tile_ids[bb + t] = y * gx + x;
      gauss_ids[bb + t] = bg;
    }
  }
}

__global__ void __launch_bounds__(kBlock) zero_u32_kernel(uint32_t* p, size_t n) {
  const size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x;
  if (i < n) p[i] = 0u;
}

// ranges[t] = [first, last+1) positions of tile t in the sorted instance list.
__global__ void __launch_bounds__(kBlock) tile_ranges_kernel(uint32_t N, const uint32_t* __restrict__ tile_ids_sorted,
                                                             uint32_t* __restrict__ ranges) {
  const uint32_t i = blockIdx.x * kBlock + threadIdx.x;
  if (i >= N) return;
  const uint32_t t = tile_ids_sorted[i];
  if (i == 0) {
    ranges[2 * t] = 0u;
  } else {
    const uint32_t prev = tile_ids_sorted[i - 1];
    if (t != prev) {
      ranges[2 * prev + 1] = i;
      ranges[2 * t] = i;
    }
  }
  if (i == N - 1) ranges[2 * t + 1] = N;
}

size_t depth_sort_temp_bytes(int P) {
  size_t bytes = 0;
  (void)rocprim::radix_sort_pairs(nullptr, bytes, (const uint32_t*)nullptr, (uint32_t*)nullptr,
                            rocprim::counting_iterator<uint32_t>(0), (uint32_t*)nullptr, (size_t)P, 0, 32);
  return bytes;
}

size_t scan_temp_bytes(int P) {
  size_t bytes = 0;
  auto in = rocprim::make_transform_iterator((const uint32_t*)nullptr, TilesOf{nullptr});
  (void)rocprim::inclusive_scan(nullptr, bytes, in, (uint32_t*)nullptr, (size_t)P, rocprim::plus<uint32_t>());
  return bytes;
}

size_t tile_sort_temp_bytes(uint32_t N, unsigned bits) {
  size_t bytes = 0;
  (void)rocprim::radix_sort_pairs(nullptr, bytes, (const uint32_t*)nullptr, (uint32_t*)nullptr,
                            (const uint32_t*)nullptr, (uint32_t*)nullptr, (size_t)N, 0, bits);
  return bytes;
}

}  // namespace

extern "C" size_t gsr_bin_prepare_bytes(int P) {
  if (P <= 0) return 256;
  const size_t keys = gsr::align_up((size_t)P * 4, 256);
  const size_t a = depth_sort_temp_bytes(P), b = scan_temp_bytes(P);
  return keys + gsr::align_up(a > b ? a : b, 256) + 256;
}

extern "C" int gsr_bin_prepare(int P, const uint32_t* depth_keys, const uint32_t* tiles_touched,
                               uint32_t* order, uint32_t* offsets, void* scratch, size_t scratch_bytes,
                               void* stream) {
  GSR_REQUIRE(P >= 0, "bin_prepare: bad P");
  if (P == 0) return GSR_OK;
  GSR_REQUIRE(depth_keys && tiles_touched && order && offsets && scratch, "bin_prepare: null pointer");
  if (scratch_bytes < gsr_bin_prepare_bytes(P) || ((uintptr_t)scratch & 255))
    return gsr::fail(GSR_E_SCRATCH, "bin_prepare: scratch too small (%zu < %zu) or not 256-byte aligned",
                     scratch_bytes, gsr_bin_prepare_bytes(P));
  hipStream_t st = (hipStream_t)stream;
  char* base = (char*)scratch;
  uint32_t* keys_sorted = (uint32_t*)base;
  void* temp = base + gsr::align_up((size_t)P * 4, 256);
  size_t temp_bytes = scratch_bytes - gsr::align_up((size_t)P * 4, 256);
  hipError_t e = rocprim::radix_sort_pairs(temp, temp_bytes, depth_keys, keys_sorted,
                                           rocprim::counting_iterator<uint32_t>(0), order, (size_t)P, 0, 32, st);
  if (e != hipSuccess) return gsr::fail(GSR_E_LAUNCH, "bin_prepare: depth sort: %s", hipGetErrorString(e));
  temp_bytes = scratch_bytes - gsr::align_up((size_t)P * 4, 256);
  auto in = rocprim::make_transform_iterator((const uint32_t*)order, TilesOf{tiles_touched});
  e = rocprim::inclusive_scan(temp, temp_bytes, in, offsets, (size_t)P, rocprim::plus<uint32_t>(), st);
  if (e != hipSuccess) return gsr::fail(GSR_E_LAUNCH, "bin_prepare: scan: %s", hipGetErrorString(e));
  return gsr::check_launch("bin_prepare");
}

extern "C" size_t gsr_bin_emit_bytes(int P, uint32_t N, int W, int H) {
  (void)P;
  if (N == 0 || W <= 0 || H <= 0) return 256;
  const unsigned tiles = (unsigned)gsr::div_up(W, GSR_TILE_X) * (unsigned)gsr::div_up(H, GSR_TILE_Y);
  const size_t arr = gsr::align_up((size_t)N * 4, 256);
  return 3 * arr + gsr::align_up(tile_sort_temp_bytes(N, bits_for(tiles)), 256) + 256;
}

extern "C" int gsr_bin_emit(int P, uint32_t N, int W, int H, const uint32_t* records, const uint32_t* order,
                            const uint32_t* offsets, uint32_t* point_list, uint32_t* ranges,
                            uint32_t* tile_ids_sorted, void* scratch, size_t scratch_bytes, void* stream) {
  GSR_REQUIRE(P >= 0 && W > 0 && H > 0, "bin_emit: bad sizes");
  GSR_REQUIRE(ranges, "bin_emit: null ranges");
  hipStream_t st = (hipStream_t)stream;
  const int gx = gsr::div_up(W, GSR_TILE_X), gy = gsr::div_up(H, GSR_TILE_Y);
  const size_t tiles = (size_t)gx * gy;
  hipLaunchKernelGGL(zero_u32_kernel, dim3((unsigned)gsr::div_up((int)(2 * tiles), kBlock)), dim3(kBlock), 0, st,
                     ranges, 2 * tiles);
  if (N == 0 || P == 0) return gsr::check_launch("bin_emit");
  GSR_REQUIRE(records && order && offsets && point_list && scratch, "bin_emit: null pointer");
  if (scratch_bytes < gsr_bin_emit_bytes(P, N, W, H) || ((uintptr_t)scratch & 255))
    return gsr::fail(GSR_E_SCRATCH, "bin_emit: scratch too small (%zu < %zu) or not 256-byte aligned",
                     scratch_bytes, gsr_bin_emit_bytes(P, N, W, H));
  char* base = (char*)scratch;
  const size_t arr = gsr::align_up((size_t)N * 4, 256);
  uint32_t* tile_ids = (uint32_t*)base;
  uint32_t* gauss_ids = (uint32_t*)(base + arr);
  uint32_t* tiles_sorted = tile_ids_sorted ? tile_ids_sorted : (uint32_t*)(base + 2 * arr);
  void* temp = base + 3 * arr;
  size_t temp_bytes = scratch_bytes - 3 * arr;

  hipLaunchKernelGGL(emit_instances_kernel, dim3(gsr::div_up(P, kBlock)), dim3(kBlock), 0, st, P, gx, records,
                     order, offsets, tile_ids, gauss_ids);
  int rc = gsr::check_launch("bin_emit: emit");
  if (rc) return rc;
  hipError_t e = rocprim::radix_sort_pairs(temp, temp_bytes, (const uint32_t*)tile_ids, tiles_sorted,
                                           (const uint32_t*)gauss_ids, point_list, (size_t)N, 0,
                                           bits_for((unsigned)tiles), st);
  if (e != hipSuccess) return gsr::fail(GSR_E_LAUNCH, "bin_emit: tile sort: %s", hipGetErrorString(e));
  hipLaunchKernelGGL(tile_ranges_kernel, dim3((N + kBlock - 1) / kBlock), dim3(kBlock), 0, st, N, tiles_sorted, ranges);
  return gsr::check_launch("bin_emit: ranges");
}
