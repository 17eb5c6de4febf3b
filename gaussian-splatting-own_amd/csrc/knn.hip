// Mean squared distance of every point to its three nearest neighbours, for gfx950.
//
// The reference initialises the Gaussians' scales from this quantity
// (scene/gaussian_model.py:134-135: distCUDA2 of the simple_knn extension, an
// un-vendored submodule whose directory is empty in the reference tree).  The
// result is defined by its use: for point i, the mean of the three smallest
// squared Euclidean distances to the OTHER points (fewer when the cloud has
// fewer than four points; 0 for a single point).
//
// Exact search, not an approximation: the points are sorted along a Morton
// curve (30-bit codes in the cloud's bounding box), cut into boxes of 1024
// consecutive points with their bounding boxes; a point starts from its
// neighbours along the curve and then visits every box that is closer than its
// current third-best distance.  One lane per point; the box index and the
// candidate index are wave-uniform, so boxes and candidates arrive through
// scalar loads and only the distance arithmetic is per lane.
//
// Compiled with -ffp-contract=off: a distance is ((dx*dx + dy*dy) + dz*dz) in
// fp32, rounded after every operation, so a CPU restatement gets the same bits.
#include <cstring>

#include <rocprim/rocprim.hpp>

#include "gsr_common.h"

namespace {

constexpr int kBlock = 256;
constexpr int kBox = 1024;

struct Bounds { float lo[3], hi[3]; };

__device__ __forceinline__ float wave_min(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fminf(v, __shfl_xor(v, o));
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
  return v;
}

// Per-block bounds of points[first .. first + count): out[block] (6 floats).
__global__ void __launch_bounds__(kBlock) bounds_kernel(int P, int per_block, const float* __restrict__ pts,
                                                        const uint32_t* __restrict__ index, float* __restrict__ out) {
  __shared__ float s[6][kBlock / 64];
  const int first = blockIdx.x * per_block, last = min(P, first + per_block);
  float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
  for (int i = first + (int)threadIdx.x; i < last; i += kBlock) {
    const size_t j = index ? index[i] : (uint32_t)i;
#pragma unroll
    for (int c = 0; c < 3; c++) {
      const float v = pts[3 * j + c];
      lo[c] = fminf(lo[c], v);
      hi[c] = fmaxf(hi[c], v);
    }
  }
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
#pragma unroll
  for (int c = 0; c < 3; c++) {
    const float a = wave_min(lo[c]), b = wave_max(hi[c]);
    if (lane == 0) { s[c][wave] = a; s[3 + c][wave] = b; }
  }
  __syncthreads();
  if (threadIdx.x < 6) {
    const int c = threadIdx.x;
    float v = s[c][0];
    for (int w = 1; w < kBlock / 64; w++) v = c < 3 ? fminf(v, s[c][w]) : fmaxf(v, s[c][w]);
    out[6 * blockIdx.x + c] = v;
  }
}

// Reduces `n` bounds records to one (record 0 of out).
__global__ void __launch_bounds__(kBlock) bounds_final_kernel(int n, const float* __restrict__ in, float* __restrict__ out) {
  __shared__ float s[6][kBlock / 64];
  float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
  for (int i = threadIdx.x; i < n; i += kBlock) {
#pragma unroll
    for (int c = 0; c < 3; c++) { lo[c] = fminf(lo[c], in[6 * i + c]); hi[c] = fmaxf(hi[c], in[6 * i + 3 + c]); }
  }
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
#pragma unroll
  for (int c = 0; c < 3; c++) {
    const float a = wave_min(lo[c]), b = wave_max(hi[c]);
    if (lane == 0) { s[c][wave] = a; s[3 + c][wave] = b; }
  }
  __syncthreads();
  if (threadIdx.x < 6) {
    const int c = threadIdx.x;
    float v = s[c][0];
    for (int w = 1; w < kBlock / 64; w++) v = c < 3 ? fminf(v, s[c][w]) : fmaxf(v, s[c][w]);
    out[c] = v;
  }
}

__device__ __forceinline__ uint32_t spread10(uint32_t x) {   // 10 bits -> every third bit
  x = (x | (x << 16)) & 0x030000FFu;
  x = (x | (x << 8)) & 0x0300F00Fu;
  x = (x | (x << 4)) & 0x030C30C3u;
  x = (x | (x << 2)) & 0x09249249u;
  return x;
}

__global__ void __launch_bounds__(kBlock) morton_kernel(int P, const float* __restrict__ pts,
                                                        const float* __restrict__ bounds, uint32_t* __restrict__ codes) {
  const int i = blockIdx.x * kBlock + threadIdx.x;
  if (i >= P) return;
  uint32_t q[3];
#pragma unroll
  for (int c = 0; c < 3; c++) {
    const float lo = bounds[c], ext = bounds[3 + c] - lo;
    const float t = ext > 0.0f ? (pts[3 * (size_t)i + c] - lo) / ext : 0.0f;
    q[c] = (uint32_t)fminf(1023.0f, fmaxf(0.0f, t * 1023.0f));
  }
  codes[i] = spread10(q[0]) | (spread10(q[1]) << 1) | (spread10(q[2]) << 2);
}

// points in curve order, as float4 (x, y, z, unused)
__global__ void __launch_bounds__(kBlock) gather_kernel(int P, const float* __restrict__ pts,
                                                        const uint32_t* __restrict__ order, float4* __restrict__ sorted) {
  const int i = blockIdx.x * kBlock + threadIdx.x;
  if (i >= P) return;
  const size_t j = order[i];
  sorted[i] = make_float4(pts[3 * j], pts[3 * j + 1], pts[3 * j + 2], 0.0f);
}

__device__ __forceinline__ float dist2(float ax, float ay, float az, float bx, float by, float bz) {
  const float dx = ax - bx, dy = ay - by, dz = az - bz;
  return (dx * dx + dy * dy) + dz * dz;
}

__device__ __forceinline__ void keep3(float d, float* best) {   // best[0] <= best[1] <= best[2]
  if (d < best[2]) {
    if (d < best[1]) {
      best[2] = best[1];
      if (d < best[0]) { best[1] = best[0]; best[0] = d; } else { best[1] = d; }
    } else {
      best[2] = d;
    }
  }
}

__global__ void __launch_bounds__(kBlock) knn3_kernel(int P, int boxes, const float4* __restrict__ sorted,
                                                      const uint32_t* __restrict__ order,
                                                      const float* __restrict__ box_bounds, float* __restrict__ out) {
  const int i = blockIdx.x * kBlock + threadIdx.x;
  const bool valid = i < P;
  const float4 me = valid ? sorted[i] : make_float4(0.f, 0.f, 0.f, 0.f);
  float best[3] = {INFINITY, INFINITY, INFINITY};
  if (valid) {
    // neighbours along the curve: a good first bound
    for (int j = max(0, i - 3); j <= min(P - 1, i + 3); j++) {
      if (j == i) continue;
      const float4 q = sorted[j];
      keep3(dist2(me.x, me.y, me.z, q.x, q.y, q.z), best);
    }
  }
  for (int b = 0; b < boxes; b++) {
    // b is uniform over the wave: the box record is one scalar load
    const float* bb = box_bounds + 6 * (size_t)b;
    const float cx = fmaxf(bb[0], fminf(me.x, bb[3])), cy = fmaxf(bb[1], fminf(me.y, bb[4])),
                cz = fmaxf(bb[2], fminf(me.z, bb[5]));
    // <=: a candidate exactly as far as the third best changes nothing, but a box at distance 0
    // from a point whose bests are still infinite must be visited
    const bool want = valid && dist2(me.x, me.y, me.z, cx, cy, cz) <= best[2];
    if (__ballot(want) == 0ull) continue;
    const int first = b * kBox, last = min(P, first + kBox);
    for (int j = first; j < last; j++) {
      const float4 q = sorted[j];                      // uniform address
      if (want && j != i && (j < i - 3 || j > i + 3)) keep3(dist2(me.x, me.y, me.z, q.x, q.y, q.z), best);
    }
  }
  if (valid) {
    float sum = 0.0f;
    int n = 0;
#pragma unroll
    for (int k = 0; k < 3; k++)
      if (best[k] < INFINITY) { sum = sum + best[k]; n++; }
    out[order[i]] = n ? sum / (float)n : 0.0f;
  }
}

size_t sort_temp_bytes(int P) {
  size_t bytes = 0;
  (void)rocprim::radix_sort_pairs(nullptr, bytes, (const uint32_t*)nullptr, (uint32_t*)nullptr,
                                  rocprim::counting_iterator<uint32_t>(0), (uint32_t*)nullptr, (size_t)P, 0, 30);
  return bytes;
}

struct Layout { size_t partial, bounds, codes, codes_sorted, order, sorted, box_bounds, temp, temp_bytes, total; int blocks, boxes; };

Layout layout(int P) {
  Layout L;
  size_t o = 0;
  auto take = [&](size_t bytes) { const size_t at = o; o += gsr::align_up(bytes ? bytes : 1, 256); return at; };
  L.blocks = gsr::div_up(P, 4096) < 1024 ? gsr::div_up(P, 4096) : 1024;
  L.boxes = gsr::div_up(P, kBox);
  L.partial = take((size_t)L.blocks * 24);
  L.bounds = take(24);
  L.codes = take((size_t)P * 4);
  L.codes_sorted = take((size_t)P * 4);
  L.order = take((size_t)P * 4);
  L.sorted = take((size_t)P * 16);
  L.box_bounds = take((size_t)L.boxes * 24);
  L.temp_bytes = sort_temp_bytes(P);
  L.temp = take(L.temp_bytes);
  L.total = o + 256;
  return L;
}

}  // namespace

extern "C" size_t gsr_knn3_bytes(int P) { return P <= 0 ? 256 : layout(P).total; }

extern "C" int gsr_knn3_mean_dist2(int P, const float* points, float* mean_dist2, void* scratch, size_t scratch_bytes,
                                   void* stream) {
  GSR_REQUIRE(P >= 0, "knn3: bad P");
  if (P == 0) return GSR_OK;
  GSR_REQUIRE(points && mean_dist2 && scratch, "knn3: null pointer");
  const Layout L = layout(P);
  if (scratch_bytes < L.total || ((uintptr_t)scratch & 255))
    return gsr::fail(GSR_E_SCRATCH, "knn3: scratch too small (%zu < %zu) or not 256-byte aligned", scratch_bytes, L.total);
  hipStream_t st = (hipStream_t)stream;
  char* base = (char*)scratch;
  float* partial = (float*)(base + L.partial);
  float* bounds = (float*)(base + L.bounds);
  uint32_t* codes = (uint32_t*)(base + L.codes);
  uint32_t* codes_sorted = (uint32_t*)(base + L.codes_sorted);
  uint32_t* order = (uint32_t*)(base + L.order);
  float4* sorted = (float4*)(base + L.sorted);
  float* box_bounds = (float*)(base + L.box_bounds);
  const int per_block = gsr::div_up(P, L.blocks);
  hipLaunchKernelGGL(bounds_kernel, dim3(L.blocks), dim3(kBlock), 0, st, P, per_block, points, (const uint32_t*)nullptr,
                     partial);
  hipLaunchKernelGGL(bounds_final_kernel, dim3(1), dim3(kBlock), 0, st, L.blocks, partial, bounds);
  hipLaunchKernelGGL(morton_kernel, dim3(gsr::div_up(P, kBlock)), dim3(kBlock), 0, st, P, points, bounds, codes);
  int rc = gsr::check_launch("knn3: morton codes");
  if (rc) return rc;
  size_t temp_bytes = L.temp_bytes;
  hipError_t e = rocprim::radix_sort_pairs(base + L.temp, temp_bytes, (const uint32_t*)codes, codes_sorted,
                                           rocprim::counting_iterator<uint32_t>(0), order, (size_t)P, 0, 30, st);
  if (e != hipSuccess) return gsr::fail(GSR_E_LAUNCH, "knn3: sort: %s", hipGetErrorString(e));
  hipLaunchKernelGGL(gather_kernel, dim3(gsr::div_up(P, kBlock)), dim3(kBlock), 0, st, P, points, order, sorted);
  hipLaunchKernelGGL(bounds_kernel, dim3(L.boxes), dim3(kBlock), 0, st, P, kBox, points, (const uint32_t*)order,
                     box_bounds);
  hipLaunchKernelGGL(knn3_kernel, dim3(gsr::div_up(P, kBlock)), dim3(kBlock), 0, st, P, L.boxes, sorted, order,
                     box_bounds, mean_dist2);
  return gsr::check_launch("knn3");
}
