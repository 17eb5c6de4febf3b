// Tile binning for gfx950: builds, for every 16x16 tile, the front-to-back list
// of Gaussians whose tile rectangle covers it — entry for entry the list that a
// stable sort of (tile << 32 | depth) keys, emitted per Gaussian in ascending
// index and row-major tile order, produces.  Ties in (tile, depth) keep
// ascending Gaussian index.
//
// Sorting the N (tile, Gaussian) instances themselves (N ~ 20x the Gaussian
// count) is pure HBM traffic: 6+ passes over 12-byte pairs for 64-bit keys, or
// 2 passes over 8-byte pairs for tile-only keys.  Here no per-tile instance is
// ever sorted.  The rectangle structure is used instead:
//   1. stable radix sort of the P Gaussians by their 32-bit depth key;
//   2. every Gaussian, visited in depth order, emits one entry per CELL of its
//      rectangle, a cell being one tile row x one strip of 16 tiles; an entry
//      carries the Gaussian index and the rectangle's x range (8 bytes);
//   3. stable radix sort of the cell entries by cell id (10 bits at 1080p):
//      NC ~ N/6 entries;
//   4. each cell list, cut into segments of 2048 entries, is expanded in order:
//      a counting sweep (wave ballots per tile of the strip), prefix sums over
//      segments and tiles, then a writing sweep that places every Gaussian index
//      at tile_start + earlier_segments + earlier_waves + earlier_lanes.
// The per-tile order is depth order because every step is stable; the tile
// ranges fall out of the prefix sum.  HBM traffic ~ 4 N (the list itself) plus
// ~60 NC, instead of ~36 N.
//
// The sorts and the scans are rocPRIM device primitives; the rest is written here.
#include <cstring>

#include <rocprim/rocprim.hpp>

#include "gsr_common.h"

namespace {

constexpr int kBlock = 256;
constexpr int kStrip = 16;        // tiles per cell
constexpr int kSegLen = 2048;     // cell-list entries per segment

inline unsigned bits_for(unsigned n) {  // smallest b with 2^b >= n, at least 1
  unsigned b = 1;
  while ((1ull << b) < n) b++;
  return b;
}

struct Grid {
  int gx, gy, cx;                 // tiles per row, tile rows, cells per row
  __host__ __device__ int cells() const { return cx * gy; }
  __host__ __device__ int tiles() const { return gx * gy; }
};

inline Grid make_grid(int W, int H) {
  Grid g;
  g.gx = gsr::div_up(W, GSR_TILE_X);
  g.gy = gsr::div_up(H, GSR_TILE_Y);
  g.cx = gsr::div_up(g.gx, kStrip);
  return g;
}

// Number of cells the rectangle of Gaussian g touches (0 when culled).
struct CellsOf {
  const uint32_t* records;
  __device__ __forceinline__ uint32_t operator()(uint32_t g) const {
    const uint2 rr = *reinterpret_cast<const uint2*>(records + (size_t)g * GSR_REC_WORDS + GSR_REC_RECT_MIN);
    const uint32_t x0 = rr.x & 0xffffu, y0 = rr.x >> 16, x1 = rr.y & 0xffffu, y1 = rr.y >> 16;
    if (x1 <= x0 || y1 <= y0) return 0u;
    return (y1 - y0) * (((x1 - 1) / kStrip) - (x0 / kStrip) + 1);
  }
};

// [0] = sum of tiles_touched (N), [1] = last inclusive cell offset (NC)
__global__ void totals_kernel(int P, const uint32_t* __restrict__ cell_offsets, uint32_t* __restrict__ totals) {
  if (threadIdx.x == 0 && blockIdx.x == 0) totals[1] = cell_offsets[P - 1];
}

__global__ void __launch_bounds__(kBlock) emit_cells_kernel(int P, Grid grid, const uint32_t* __restrict__ records,
                                                            const uint32_t* __restrict__ order,
                                                            const uint32_t* __restrict__ cell_offsets,
                                                            uint32_t* __restrict__ keys,
                                                            unsigned long long* __restrict__ vals) {
  const int j = blockIdx.x * kBlock + threadIdx.x;
  if (j >= P) return;
  const uint32_t end = cell_offsets[j], begin = j == 0 ? 0u : cell_offsets[j - 1];
  if (end == begin) return;
  const uint32_t g = order[j];
  const uint2 rr = *reinterpret_cast<const uint2*>(records + (size_t)g * GSR_REC_WORDS + GSR_REC_RECT_MIN);
  const uint32_t x0 = rr.x & 0xffffu, y0 = rr.x >> 16, x1 = rr.y & 0xffffu, y1 = rr.y >> 16;
  const uint32_t s0 = x0 / kStrip, s1 = (x1 - 1) / kStrip;
  const unsigned long long v = (unsigned long long)g | ((unsigned long long)x0 << 32) | ((unsigned long long)x1 << 48);
  uint32_t off = begin;
  for (uint32_t y = y0; y < y1; y++)
    for (uint32_t s = s0; s <= s1; s++) {
      keys[off] = y * grid.cx + s;
      vals[off] = v;
      off++;
    }
}

__global__ void __launch_bounds__(kBlock) zero_u32_kernel(uint32_t* p, size_t n) {
  const size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x;
  if (i < n) p[i] = 0u;
}

// [first, last+1) of every key value in a sorted key array; untouched (zero) for absent keys.
__global__ void __launch_bounds__(kBlock) key_ranges_kernel(uint32_t n, const uint32_t* __restrict__ sorted_keys,
                                                            uint32_t* __restrict__ ranges) {
  const uint32_t i = blockIdx.x * kBlock + threadIdx.x;
  if (i >= n) return;
  const uint32_t t = sorted_keys[i];
  if (i == 0) {
    ranges[2 * t] = 0u;
  } else {
    const uint32_t prev = sorted_keys[i - 1];
    if (t != prev) {
      ranges[2 * prev + 1] = i;
      ranges[2 * t] = i;
    }
  }
  if (i == n - 1) ranges[2 * t + 1] = n;
}

// Cuts every cell list into segments of kSegLen entries.  One block; cells are
// few (hundreds to a few thousand).  seg_first[c] = index of cell c's first
// segment, seg_first[cells] = number of segments; seg_desc[s] = (cell, start, len).
__global__ void __launch_bounds__(1024) build_segments_kernel(int cells, const uint32_t* __restrict__ cell_ranges,
                                                              uint32_t* __restrict__ seg_first,
                                                              uint32_t* __restrict__ seg_desc) {
  __shared__ uint32_t scan[1024];
  __shared__ uint32_t carry;
  if (threadIdx.x == 0) carry = 0u;
  __syncthreads();
  for (int base = 0; base < cells; base += 1024) {
    const int c = base + (int)threadIdx.x;
    uint32_t start = 0, len = 0;
    if (c < cells) { start = cell_ranges[2 * c]; len = cell_ranges[2 * c + 1] - start; }
    const uint32_t nseg = (len + kSegLen - 1) / kSegLen;
    scan[threadIdx.x] = nseg;
    __syncthreads();
    for (int o = 1; o < 1024; o <<= 1) {
      const uint32_t add = (int)threadIdx.x >= o ? scan[threadIdx.x - o] : 0u;
      __syncthreads();
      scan[threadIdx.x] += add;
      __syncthreads();
    }
    const uint32_t first = carry + scan[threadIdx.x] - nseg;
    if (c < cells) {
      seg_first[c] = first;
      for (uint32_t k = 0; k < nseg; k++) {
        const uint32_t s = first + k;
        seg_desc[3 * s] = (uint32_t)c;
        seg_desc[3 * s + 1] = start + k * kSegLen;
        seg_desc[3 * s + 2] = min((uint32_t)kSegLen, len - k * kSegLen);
      }
    }
    __syncthreads();
    if (threadIdx.x == 1023) carry += scan[1023];
    __syncthreads();
  }
  if (threadIdx.x == 0) seg_first[cells] = carry;
}

// 16-bit mask of the strip's tiles covered by an entry's x range.
__device__ __forceinline__ uint32_t strip_mask(unsigned long long v, uint32_t xb) {
  const uint32_t x0 = (uint32_t)(v >> 32) & 0xffffu, x1 = (uint32_t)(v >> 48);
  const uint32_t lo = max(x0, xb) - xb, hi = min(x1, xb + kStrip) - xb;   // lo < hi <= 16 by construction
  return ((1u << hi) - 1u) & ~((1u << lo) - 1u);
}

// Counting sweep: seg_counts[s][t] = entries of segment s covering tile t of its strip.
__global__ void __launch_bounds__(kBlock) segment_count_kernel(Grid grid, const uint32_t* __restrict__ seg_first,
                                                               const uint32_t* __restrict__ seg_desc,
                                                               const unsigned long long* __restrict__ vals,
                                                               uint32_t* __restrict__ seg_counts) {
  __shared__ uint32_t part[4][kStrip];
  const uint32_t s = blockIdx.x;
  if (s >= seg_first[grid.cells()]) return;
  const uint32_t cell = seg_desc[3 * s], start = seg_desc[3 * s + 1], len = seg_desc[3 * s + 2];
  const uint32_t xb = (cell % grid.cx) * kStrip;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  uint32_t cnt[kStrip];
#pragma unroll
  for (int t = 0; t < kStrip; t++) cnt[t] = 0u;
  for (uint32_t i = threadIdx.x; i < (len + kBlock - 1) / kBlock * kBlock; i += kBlock) {
    const uint32_t m = i < len ? strip_mask(vals[start + i], xb) : 0u;
#pragma unroll
    for (int t = 0; t < kStrip; t++) cnt[t] += (uint32_t)__popcll(__ballot((m >> t) & 1u));
  }
  if (lane == 0) {
#pragma unroll
    for (int t = 0; t < kStrip; t++) part[wave][t] = cnt[t];
  }
  __syncthreads();
  if (threadIdx.x < kStrip)
    seg_counts[(size_t)s * kStrip + threadIdx.x] =
        part[0][threadIdx.x] + part[1][threadIdx.x] + part[2][threadIdx.x] + part[3][threadIdx.x];
}

// Per cell: running sum over its segments -> seg_base (offset of a segment inside each tile's
// list) and the tile totals.  One wave per cell, lane t < 16 owns tile t of the strip.
__global__ void __launch_bounds__(64) segment_scan_kernel(Grid grid, const uint32_t* __restrict__ seg_first,
                                                          const uint32_t* __restrict__ seg_counts,
                                                          uint32_t* __restrict__ seg_base,
                                                          uint32_t* __restrict__ tile_counts) {
  const int cell = blockIdx.x, t = threadIdx.x;
  if (t >= kStrip) return;
  const uint32_t s0 = seg_first[cell], s1 = seg_first[cell + 1];
  uint32_t run = 0;
  for (uint32_t s = s0; s < s1; s++) {
    const uint32_t c = seg_counts[(size_t)s * kStrip + t];
    seg_base[(size_t)s * kStrip + t] = run;
    run += c;
  }
  const int x = (cell % grid.cx) * kStrip + t, y = cell / grid.cx;
  if (x < grid.gx) tile_counts[y * grid.gx + x] = run;
}

// ranges[t] = [start, start + count), (0,0) for empty tiles.
__global__ void __launch_bounds__(kBlock) ranges_kernel(int tiles, const uint32_t* __restrict__ tile_counts,
                                                        const uint32_t* __restrict__ tile_starts,
                                                        uint32_t* __restrict__ ranges) {
  const int t = blockIdx.x * kBlock + threadIdx.x;
  if (t >= tiles) return;
  const uint32_t c = tile_counts[t], s = tile_starts[t];
  ranges[2 * t] = c ? s : 0u;
  ranges[2 * t + 1] = c ? s + c : 0u;
}

// Writing sweep: every entry of the segment drops its Gaussian index into each tile it covers,
// at tile_start + seg_base + (entries of earlier chunks / waves / lanes covering that tile).
__global__ void __launch_bounds__(kBlock) segment_write_kernel(Grid grid, const uint32_t* __restrict__ seg_first,
                                                               const uint32_t* __restrict__ seg_desc,
                                                               const unsigned long long* __restrict__ vals,
                                                               const uint32_t* __restrict__ seg_base,
                                                               const uint32_t* __restrict__ tile_starts,
                                                               uint32_t* __restrict__ point_list) {
  __shared__ uint32_t cursor[kStrip];
  __shared__ uint32_t part[4][kStrip];
  const uint32_t s = blockIdx.x;
  if (s >= seg_first[grid.cells()]) return;
  const uint32_t cell = seg_desc[3 * s], start = seg_desc[3 * s + 1], len = seg_desc[3 * s + 2];
  const uint32_t xb = (cell % grid.cx) * kStrip, y = cell / grid.cx;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  if (threadIdx.x < kStrip) {
    const uint32_t x = xb + threadIdx.x;
    cursor[threadIdx.x] = x < (uint32_t)grid.gx ? tile_starts[y * grid.gx + x] + seg_base[(size_t)s * kStrip + threadIdx.x] : 0u;
  }
  __syncthreads();
  const unsigned long long lt = (1ull << lane) - 1ull;
  for (uint32_t base = 0; base < len; base += kBlock) {
    const uint32_t i = base + threadIdx.x;
    unsigned long long v = 0ull;
    uint32_t m = 0u;
    if (i < len) { v = vals[start + i]; m = strip_mask(v, xb); }
    unsigned long long bal[kStrip];
#pragma unroll
    for (int t = 0; t < kStrip; t++) {
      bal[t] = __ballot((m >> t) & 1u);
      if (lane == 0) part[wave][t] = (uint32_t)__popcll(bal[t]);
    }
    __syncthreads();
    const uint32_t g = (uint32_t)v;
#pragma unroll
    for (int t = 0; t < kStrip; t++) {
      if ((m >> t) & 1u) {
        uint32_t pos = cursor[t] + (uint32_t)__popcll(bal[t] & lt);
        for (int w = 0; w < wave; w++) pos += part[w][t];
        point_list[pos] = g;
      }
    }
    __syncthreads();
    if (threadIdx.x < kStrip)
      cursor[threadIdx.x] += part[0][threadIdx.x] + part[1][threadIdx.x] + part[2][threadIdx.x] + part[3][threadIdx.x];
    __syncthreads();
  }
}

// Optional: tile id of every sorted instance (from which the 64-bit sort keys are rebuilt).
__global__ void __launch_bounds__(kBlock) fill_tile_ids_kernel(const uint32_t* __restrict__ ranges,
                                                               uint32_t* __restrict__ tile_ids) {
  const uint32_t t = blockIdx.x, b = ranges[2 * t], e = ranges[2 * t + 1];
  for (uint32_t i = b + threadIdx.x; i < e; i += kBlock) tile_ids[i] = t;
}

// rocPRIM switches to a block sort + log2(P/block) merge passes below 2^20 items; for the
// ~1M depth keys of a large scene four Onesweep radix passes move far fewer bytes.
using DepthSortConfig =
    rocprim::radix_sort_config<rocprim::default_config, rocprim::default_config, rocprim::default_config, (1u << 16)>;

size_t depth_sort_temp_bytes(int P) {
  size_t bytes = 0;
  (void)rocprim::radix_sort_pairs<DepthSortConfig>(nullptr, bytes, (const uint32_t*)nullptr, (uint32_t*)nullptr,
                                  rocprim::counting_iterator<uint32_t>(0), (uint32_t*)nullptr, (size_t)P, 0, 32);
  return bytes;
}

size_t cell_scan_temp_bytes(int P) {
  size_t bytes = 0;
  auto in = rocprim::make_transform_iterator((const uint32_t*)nullptr, CellsOf{nullptr});
  (void)rocprim::inclusive_scan(nullptr, bytes, in, (uint32_t*)nullptr, (size_t)P, rocprim::plus<uint32_t>());
  return bytes;
}

size_t reduce_temp_bytes(int P) {
  size_t bytes = 0;
  (void)rocprim::reduce(nullptr, bytes, (const uint32_t*)nullptr, (uint32_t*)nullptr, 0u, (size_t)P,
                        rocprim::plus<uint32_t>());
  return bytes;
}

size_t cell_sort_temp_bytes(uint32_t NC, unsigned bits) {
  size_t bytes = 0;
  (void)rocprim::radix_sort_pairs(nullptr, bytes, (const uint32_t*)nullptr, (uint32_t*)nullptr,
                                  (const unsigned long long*)nullptr, (unsigned long long*)nullptr, (size_t)NC, 0, bits);
  return bytes;
}

size_t tile_scan_temp_bytes(int tiles) {
  size_t bytes = 0;
  (void)rocprim::exclusive_scan(nullptr, bytes, (const uint32_t*)nullptr, (uint32_t*)nullptr, 0u, (size_t)tiles,
                                rocprim::plus<uint32_t>());
  return bytes;
}

// Scratch layout of gsr_bin_emit, in 256-byte aligned pieces.
struct EmitLayout {
  size_t keys, vals, keys_sorted, vals_sorted, cell_ranges, seg_first, seg_desc, seg_counts, seg_base, tile_counts,
      tile_starts, temp, temp_bytes, total;
  uint32_t max_segments;
};

EmitLayout emit_layout(uint32_t NC, const Grid& g) {
  EmitLayout L;
  size_t o = 0;
  auto take = [&](size_t bytes) { const size_t at = o; o += gsr::align_up(bytes ? bytes : 1, 256); return at; };
  L.max_segments = NC / kSegLen + (uint32_t)g.cells() + 1;
  L.keys = take((size_t)NC * 4);
  L.vals = take((size_t)NC * 8);
  L.keys_sorted = take((size_t)NC * 4);
  L.vals_sorted = take((size_t)NC * 8);
  L.cell_ranges = take((size_t)g.cells() * 8);
  L.seg_first = take(((size_t)g.cells() + 1) * 4);
  L.seg_desc = take((size_t)L.max_segments * 12);
  L.seg_counts = take((size_t)L.max_segments * kStrip * 4);
  L.seg_base = take((size_t)L.max_segments * kStrip * 4);
  L.tile_counts = take((size_t)g.tiles() * 4);
  L.tile_starts = take((size_t)g.tiles() * 4);
  const size_t a = cell_sort_temp_bytes(NC, bits_for((unsigned)g.cells())), b = tile_scan_temp_bytes(g.tiles());
  L.temp_bytes = a > b ? a : b;
  L.temp = take(L.temp_bytes);
  L.total = o + 256;
  return L;
}

}  // namespace

extern "C" size_t gsr_bin_prepare_bytes(int P) {
  if (P <= 0) return 256;
  const size_t keys = gsr::align_up((size_t)P * 4, 256);
  size_t t = depth_sort_temp_bytes(P);
  const size_t b = cell_scan_temp_bytes(P), c = reduce_temp_bytes(P);
  if (b > t) t = b;
  if (c > t) t = c;
  return keys + gsr::align_up(t, 256) + 256;
}

extern "C" int gsr_bin_prepare(int P, const uint32_t* records, const uint32_t* depth_keys,
                               const uint32_t* tiles_touched, uint32_t* order, uint32_t* cell_offsets,
                               uint32_t* totals, void* scratch, size_t scratch_bytes, void* stream) {
  GSR_REQUIRE(P >= 0, "bin_prepare: bad P");
  if (P == 0) return GSR_OK;
  GSR_REQUIRE(records && depth_keys && tiles_touched && order && cell_offsets && totals && scratch,
              "bin_prepare: null pointer");
  if (scratch_bytes < gsr_bin_prepare_bytes(P) || ((uintptr_t)scratch & 255))
    return gsr::fail(GSR_E_SCRATCH, "bin_prepare: scratch too small (%zu < %zu) or not 256-byte aligned",
                     scratch_bytes, gsr_bin_prepare_bytes(P));
  hipStream_t st = (hipStream_t)stream;
  char* base = (char*)scratch;
  const size_t keys_bytes = gsr::align_up((size_t)P * 4, 256);
  uint32_t* keys_sorted = (uint32_t*)base;
  void* temp = base + keys_bytes;
  size_t temp_bytes = scratch_bytes - keys_bytes;
  hipError_t e = rocprim::radix_sort_pairs<DepthSortConfig>(temp, temp_bytes, depth_keys, keys_sorted,
                                           rocprim::counting_iterator<uint32_t>(0), order, (size_t)P, 0, 32, st);
  if (e != hipSuccess) return gsr::fail(GSR_E_LAUNCH, "bin_prepare: depth sort: %s", hipGetErrorString(e));
  temp_bytes = scratch_bytes - keys_bytes;
  auto in = rocprim::make_transform_iterator((const uint32_t*)order, CellsOf{records});
  e = rocprim::inclusive_scan(temp, temp_bytes, in, cell_offsets, (size_t)P, rocprim::plus<uint32_t>(), st);
  if (e != hipSuccess) return gsr::fail(GSR_E_LAUNCH, "bin_prepare: cell scan: %s", hipGetErrorString(e));
  temp_bytes = scratch_bytes - keys_bytes;
  e = rocprim::reduce(temp, temp_bytes, tiles_touched, totals, 0u, (size_t)P, rocprim::plus<uint32_t>(), st);
  if (e != hipSuccess) return gsr::fail(GSR_E_LAUNCH, "bin_prepare: reduce: %s", hipGetErrorString(e));
  hipLaunchKernelGGL(totals_kernel, dim3(1), dim3(64), 0, st, P, cell_offsets, totals);
  return gsr::check_launch("bin_prepare");
}

extern "C" size_t gsr_bin_emit_bytes(int P, uint32_t N, uint32_t NC, int W, int H) {
  (void)P;
  if (N == 0 || NC == 0 || W <= 0 || H <= 0) return 256;
  return emit_layout(NC, make_grid(W, H)).total;
}

extern "C" int gsr_bin_emit(int P, uint32_t N, uint32_t NC, int W, int H, const uint32_t* records,
                            const uint32_t* order, const uint32_t* cell_offsets, uint32_t* point_list,
                            uint32_t* ranges, uint32_t* tile_ids_sorted, void* scratch, size_t scratch_bytes,
                            void* stream) {
  GSR_REQUIRE(P >= 0 && W > 0 && H > 0, "bin_emit: bad sizes");
  GSR_REQUIRE(ranges, "bin_emit: null ranges");
  hipStream_t st = (hipStream_t)stream;
  const Grid g = make_grid(W, H);
  if (N == 0 || NC == 0 || P == 0) {
    hipLaunchKernelGGL(zero_u32_kernel, dim3((unsigned)gsr::div_up(2 * g.tiles(), kBlock)), dim3(kBlock), 0, st, ranges,
                       (size_t)2 * g.tiles());
    return gsr::check_launch("bin_emit");
  }
  GSR_REQUIRE(records && order && cell_offsets && point_list && scratch, "bin_emit: null pointer");
  GSR_REQUIRE(NC <= N, "bin_emit: NC=%u exceeds N=%u", NC, N);
  const EmitLayout L = emit_layout(NC, g);
  if (scratch_bytes < L.total || ((uintptr_t)scratch & 255))
    return gsr::fail(GSR_E_SCRATCH, "bin_emit: scratch too small (%zu < %zu) or not 256-byte aligned", scratch_bytes,
                     L.total);
  char* base = (char*)scratch;
  uint32_t* keys = (uint32_t*)(base + L.keys);
  unsigned long long* vals = (unsigned long long*)(base + L.vals);
  uint32_t* keys_sorted = (uint32_t*)(base + L.keys_sorted);
  unsigned long long* vals_sorted = (unsigned long long*)(base + L.vals_sorted);
  uint32_t* cell_ranges = (uint32_t*)(base + L.cell_ranges);
  uint32_t* seg_first = (uint32_t*)(base + L.seg_first);
  uint32_t* seg_desc = (uint32_t*)(base + L.seg_desc);
  uint32_t* seg_counts = (uint32_t*)(base + L.seg_counts);
  uint32_t* seg_base = (uint32_t*)(base + L.seg_base);
  uint32_t* tile_counts = (uint32_t*)(base + L.tile_counts);
  uint32_t* tile_starts = (uint32_t*)(base + L.tile_starts);
  void* temp = base + L.temp;

  hipLaunchKernelGGL(emit_cells_kernel, dim3(gsr::div_up(P, kBlock)), dim3(kBlock), 0, st, P, g, records, order,
                     cell_offsets, keys, vals);
  int rc = gsr::check_launch("bin_emit: emit cells");
  if (rc) return rc;
  size_t temp_bytes = L.temp_bytes;
  hipError_t e = rocprim::radix_sort_pairs(temp, temp_bytes, (const uint32_t*)keys, keys_sorted,
                                           (const unsigned long long*)vals, vals_sorted, (size_t)NC, 0,
                                           bits_for((unsigned)g.cells()), st);
  if (e != hipSuccess) return gsr::fail(GSR_E_LAUNCH, "bin_emit: cell sort: %s", hipGetErrorString(e));
  // zero cell ranges and tile counts (adjacent pieces are zeroed separately: they are not contiguous)
  hipLaunchKernelGGL(zero_u32_kernel, dim3((unsigned)gsr::div_up(2 * g.cells(), kBlock)), dim3(kBlock), 0, st,
                     cell_ranges, (size_t)2 * g.cells());
  hipLaunchKernelGGL(zero_u32_kernel, dim3((unsigned)gsr::div_up(g.tiles(), kBlock)), dim3(kBlock), 0, st, tile_counts,
                     (size_t)g.tiles());
  hipLaunchKernelGGL(key_ranges_kernel, dim3((NC + kBlock - 1) / kBlock), dim3(kBlock), 0, st, NC, keys_sorted,
                     cell_ranges);
  hipLaunchKernelGGL(build_segments_kernel, dim3(1), dim3(1024), 0, st, g.cells(), cell_ranges, seg_first, seg_desc);
  hipLaunchKernelGGL(segment_count_kernel, dim3(L.max_segments), dim3(kBlock), 0, st, g, seg_first, seg_desc,
                     vals_sorted, seg_counts);
  hipLaunchKernelGGL(segment_scan_kernel, dim3(g.cells()), dim3(64), 0, st, g, seg_first, seg_counts, seg_base,
                     tile_counts);
  rc = gsr::check_launch("bin_emit: segment kernels");
  if (rc) return rc;
  temp_bytes = L.temp_bytes;
  e = rocprim::exclusive_scan(temp, temp_bytes, (const uint32_t*)tile_counts, tile_starts, 0u, (size_t)g.tiles(),
                              rocprim::plus<uint32_t>(), st);
  if (e != hipSuccess) return gsr::fail(GSR_E_LAUNCH, "bin_emit: tile scan: %s", hipGetErrorString(e));
  hipLaunchKernelGGL(ranges_kernel, dim3(gsr::div_up(g.tiles(), kBlock)), dim3(kBlock), 0, st, g.tiles(), tile_counts,
                     tile_starts, ranges);
  hipLaunchKernelGGL(segment_write_kernel, dim3(L.max_segments), dim3(kBlock), 0, st, g, seg_first, seg_desc,
                     vals_sorted, seg_base, tile_starts, point_list);
  if (tile_ids_sorted)
    hipLaunchKernelGGL(fill_tile_ids_kernel, dim3(g.tiles()), dim3(kBlock), 0, st, ranges, tile_ids_sorted);
  return gsr::check_launch("bin_emit: write");
}
