// Shared host-side plumbing of libgsr_hip.so: error reporting and launch checks.
#pragma once

#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>

#include "../../include/gsr.h"
#include "../../include/gsr_constants.h"

namespace gsr {

// Records the message returned by gsr_last_error() for this thread; returns code.
int fail(int code, const char* fmt, ...);

inline int check_launch(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(GSR_E_LAUNCH, "%s: %s", what, hipGetErrorString(e));
  return GSR_OK;
}

inline int div_up(int a, int b) { return (a + b - 1) / b; }
inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

inline unsigned bits_for(unsigned n) {  // smallest b with 2^b >= n, at least 1
  unsigned b = 1;
  while ((1ull << b) < n) b++;
  return b;
}

// Layout of the 64-bit word that travels through the depth sort of the binning:
// index | x0 << bi | (x1 - 1) << (bi + bx) | y0 << (bi + 2 bx) | (y1 - 1) << (bi + 2 bx + by),
// (x0, y0, x1, y1) the Gaussian's tile rectangle.  The preprocess writes it, the binning reads it.
struct Pack {
  uint32_t bi, bx, by;
  __host__ __device__ uint32_t bits() const { return bi + 2 * bx + 2 * by; }
  __device__ __forceinline__ unsigned long long pack(uint32_t i, uint32_t x0, uint32_t x1m, uint32_t y0,
                                                     uint32_t y1m) const {
    typedef unsigned long long u64;
    return (u64)i | ((u64)x0 << bi) | ((u64)x1m << (bi + bx)) | ((u64)y0 << (bi + 2 * bx)) |
           ((u64)y1m << (bi + 2 * bx + by));
  }
  __device__ __forceinline__ uint32_t index(unsigned long long v) const {
    return (uint32_t)(v & ((1ull << bi) - 1ull));
  }
  __device__ __forceinline__ void rect(unsigned long long v, uint32_t& x0, uint32_t& x1m, uint32_t& y0,
                                       uint32_t& y1m) const {
    const uint32_t mx = (1u << bx) - 1u, my = (1u << by) - 1u;
    x0 = (uint32_t)(v >> bi) & mx;
    x1m = (uint32_t)(v >> (bi + bx)) & mx;
    y0 = (uint32_t)(v >> (bi + 2 * bx)) & my;
    y1m = (uint32_t)(v >> (bi + 2 * bx + by)) & my;
  }
};

inline Pack make_pack(int P, int tiles_x, int tiles_y) {
  Pack k;
  k.bi = bits_for((unsigned)(P > 1 ? P : 2));
  k.bx = bits_for((unsigned)tiles_x);
  k.by = bits_for((unsigned)tiles_y);
  return k;
}

// Inclusive prefix sum over the 64 lanes of a wave with DPP alone: row_shr 1, 2, 4, 8 inside the rows of 16 lanes
// (lanes whose source lies outside their row add nothing), then lane 15 of rows 0 and 2 into the rows behind them,
// then lane 31 into rows 2 and 3.  Six dependent register operations; the same scan with __shfl_up is six round
// trips through the LDS crossbar (ds_bpermute), a tenth of a microsecond each, and the one-workgroup scan
// kernels of the binning are chains of exactly such steps.
__device__ __forceinline__ uint32_t wave_inclusive_sum(uint32_t v) {
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, false);   // row_shr:1
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, false);   // row_shr:2
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, false);   // row_shr:4
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xf, false);   // row_shr:8
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xa, 0xf, false);   // row_bcast:15 into rows 1, 3
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xc, 0xf, false);   // row_bcast:31 into rows 2, 3
  return v;
}

// Sum over the 64 lanes of a wave, returned to every lane (through a scalar register): the inclusive scan's last lane.
__device__ __forceinline__ uint32_t wave_sum(uint32_t v) {
  return (uint32_t)__builtin_amdgcn_readlane((int)wave_inclusive_sum(v), 63);
}
// The same for floats (an invalid source adds +0).
__device__ __forceinline__ float wave_sum_f32(float v) {
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x111, 0xf, 0xf, false));
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x112, 0xf, 0xf, false));
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x114, 0xf, 0xf, false));
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x118, 0xf, 0xf, false));
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x142, 0xa, 0xf, false));
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x143, 0xc, 0xf, false));
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}
// The value lane `from` holds, `from` the same in every lane (a register read, not a shuffle through LDS).
__device__ __forceinline__ uint32_t lane_value(uint32_t v, int from) {
  return (uint32_t)__builtin_amdgcn_readlane((int)v, __builtin_amdgcn_readfirstlane(from));
}

// The list sizes are 32-bit sums (as the published module's are: its instance count is an int), so a frame may hold
// at most kMaxInstances tile instances.  More would wrap the sums and let the list builder write past the lists, so
// the summing workgroup also adds the instance counts up as floats (exact to a few thousand at this size, hence
// the margin below 2^31) and raises a flag that stops the forward before any list is sized.
// (A block's own partial sum cannot wrap: 256 Gaussians on at most 4080 x 4080 tiles, the largest image accepted.)
constexpr float kMaxInstances = (float)GSR_MAX_INSTANCES;   // 2^31 - 2^22

// One workgroup of THREADS threads sums `blocks` partial records of four counters (partials[4 * b + k]) into
// totals[0..3], copies the caller's status word (null = 0) to totals[4] and sets totals[5] to 1 if the first
// counter's true sum is kMaxInstances or more (then totals[0..3] mean nothing).  Returns totals[t] to thread
// t < 6 (0 to the others).  s_part: THREADS / 64 rows of LDS.
template <int THREADS>
__device__ __forceinline__ uint32_t sum_totals(int blocks, const uint32_t* __restrict__ partials,
                                               const uint32_t* __restrict__ status, uint32_t* __restrict__ totals,
                                               uint32_t (*s_part)[5]) {
  uint32_t acc[4] = {0u, 0u, 0u, 0u};
  float instances = 0.f;
  // four independent 16-byte loads per thread and round
  for (int b0 = threadIdx.x; b0 < blocks; b0 += 4 * THREADS) {
    uint4 p[4];
#pragma unroll
    for (int k = 0; k < 4; k++) {
      const int b = b0 + k * THREADS;
      p[k] = b < blocks ? reinterpret_cast<const uint4*>(partials)[b] : make_uint4(0u, 0u, 0u, 0u);
    }
#pragma unroll
    for (int k = 0; k < 4; k++) {
      acc[0] += p[k].x; acc[1] += p[k].y; acc[2] += p[k].z; acc[3] += p[k].w;
      instances += (float)p[k].x;
    }
  }
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
#pragma unroll
  for (int k = 0; k < 4; k++) {
    acc[k] = wave_sum(acc[k]);
    if (lane == 0) s_part[wave][k] = acc[k];
  }
  instances = wave_sum_f32(instances);
  if (lane == 0) s_part[wave][4] = __float_as_uint(instances);
  __syncthreads();
  uint32_t mine = 0u;
  if (threadIdx.x < 4) {
#pragma unroll
    for (int w = 0; w < THREADS / 64; w++) mine += s_part[w][threadIdx.x];
    totals[threadIdx.x] = mine;
  }
  if (threadIdx.x == 4) totals[4] = mine = status ? *status : 0u;
  if (threadIdx.x == 5) {
    float t = 0.f;
#pragma unroll
    for (int w = 0; w < THREADS / 64; w++) t += __uint_as_float(s_part[w][4]);
    totals[5] = mine = t >= kMaxInstances ? 1u : 0u;
  }
  return mine;
}

// The same as a kernel of one workgroup; defined with the binning kernels.
int launch_totals_final(int blocks, const uint32_t* partials, const uint32_t* status, uint32_t* totals,
                        hipStream_t stream);
// What gsr_forward asks of the preprocess beyond gsr_preprocess_fwd[_raw]'s own arguments:
// defer_totals: leave the partial records of the list sizes unsummed (P > 0), the depth sort's first kernel sums
// them; clear_at / clear_words: memory the kernel's workgroups clear as their last act (the depth sort's histograms
// and look-back states, which then needs no clearing kernel of its own).  Defined with the preprocess kernels,
// as is where the partial records are and how many.
struct PreprocessExtras {
  bool defer_totals;
  uint32_t* clear_at;
  size_t clear_words;
};
// gsr_preprocess_fwd / gsr_preprocess_fwd_raw with the extras as an explicit argument (the C entry points pass none).
int preprocess_fwd(const PreprocessExtras& extras, int P, int D, int M, const float* means3D, const float* scales,
                   float scale_modifier, const float* rotations, const float* opacities, const float* shs,
                   const float* colors_precomp, const float* cov3D_precomp, const float* viewmatrix,
                   const float* projmatrix, const float* campos, int W, int H, float tanfovx, float tanfovy,
                   int prefiltered, int reference_lists, uint32_t* records, int32_t* radii, uint32_t* tiles_touched,
                   uint32_t* depth_keys, uint32_t* rects, uint64_t* sort_vals, uint32_t* totals, const uint32_t* status,
                   uint8_t* visible, void* stream);
int preprocess_fwd_raw(const PreprocessExtras& extras, int P, int D, int n_rest, const float* xyz, const float* scaling_raw,
                       float scale_modifier, const float* rotation_raw, const float* opacity_raw, const float* features_dc,
                       const float* features_rest, const float* viewmatrix, const float* projmatrix, const float* campos,
                       int W, int H, float tanfovx, float tanfovy, int reference_lists, uint32_t* records, int32_t* radii,
                       uint32_t* tiles_touched, uint32_t* depth_keys, uint32_t* rects, uint64_t* sort_vals, uint32_t* totals,
                       const uint32_t* status, uint8_t* visible, void* stream);
int totals_partial_blocks(int P);
const uint32_t* totals_partials(const uint32_t* totals);

// Depth sort of the binning (defined in depthsort.hip): sorted_vals[0 .. visible) = the values whose
// key is not 0xFFFFFFFF, by ascending key, equal keys in input order.  scratch: depth_sort_bytes(P)
// bytes, 256-byte aligned.
size_t depth_sort_bytes(int P);
// Optional extra duty of the sort's first kernel (the clearing kernel, or the histogram kernel when the caller
// has cleared the scratch): the list sizes go to the host from there.  Its first
// workgroup sums the partial records the preprocess left (as launch_totals_final does: partials, blocks,
// status, totals), stores totals[0..4] also at dst[0..4] (a device-side address of pinned host memory, eight
// words), and then the ticket at dst[7]: the host polls that word instead of waiting for an event, which
// would cost the stream a gap of five microseconds behind the kernel.
struct SortMirror {
  const uint32_t* partials;
  int blocks;
  const uint32_t* status;
  uint32_t* totals;
  uint32_t* dst;
  uint32_t ticket;   // never 0
};
// cleared: the caller has zeroed depth_sort_clear_region() of the scratch on this stream already.
int depth_sort(int P, const uint32_t* keys, const uint64_t* vals, uint64_t* sorted_vals, void* scratch,
               hipStream_t stream, const SortMirror* mirror = nullptr, bool cleared = false);
// The words of a sort's scratch that must be zero when it starts.
void depth_sort_clear_region(int P, void* scratch, uint32_t** at, size_t* words);
// Waits for the stream, then GSR_OK or GSR_E_SORT for the sort that last used this scratch.
int depth_sort_status(const void* scratch, hipStream_t stream);
// Device address of the word that is non-zero after a sort that gave up (inside its scratch).
const uint32_t* depth_sort_error_word(const void* scratch);
// Look-back steps a workgroup waits before it gives up (0 = the default); for tests of that path.
void set_sort_spin_limit(uint32_t limit);
// The caller has the GPU to itself: sorts may skip the per-workgroup ticket (see depthsort.hip).
void set_exclusive_device(int on);

// gsr_blend_fwd_timed with a zero-fill: every workgroup clears an equal share of it as its last act; and with
// the block for the depth segments of long walks (null = none; see blend.hip): segments_bytes(W, H) bytes,
// 256-byte aligned, whose last 256 bytes (from segments_head) must be zero when the kernel starts; segment_length
// (a multiple of 256, from segment_length_for) entries per segment.
int blend_fwd(int W, int H, const uint32_t* ranges, const uint32_t* point_list, const uint32_t* records, const float* bg,
              float* out_color, float* final_T, uint32_t* n_contrib, void* zero_fill, size_t zero_fill_bytes,
              void* segments, uint32_t segment_length, void* start_event, void* stop_event, hipStream_t stream);
size_t segments_bytes(int W, int H);
uint32_t* segments_head(void* block, int W, int H);
uint32_t segment_length_for(uint32_t N);   // entries per depth segment for lists of N entries, 0 = no segments

}  // namespace gsr

#define GSR_REQUIRE(cond, ...)                                   \
  do {                                                           \
    if (!(cond)) return gsr::fail(GSR_E_INVALID, __VA_ARGS__);   \
  } while (0)
