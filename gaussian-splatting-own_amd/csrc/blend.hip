// Per-tile alpha compositing for gfx950, forward and backward.
//
// One 256-thread workgroup per 16x16 tile = 4 wave64; each wave owns one 8x8
// pixel quadrant (one pixel per lane), so that a wave finishes, and skips
// splats, on the evidence of its own 64 pixels only.  The tile's splat list is
// consumed in batches of 256: every thread gathers one 64-byte splat record
// (one cache line), tests it against the four quadrants with the splat's
// alpha >= 1/255 bounding box, and publishes it to LDS together with a 64-bit
// survivor mask per (quadrant, staging wave).  Each wave then walks only the
// set bits of its own masks — scalar bit iteration, order preserved — and reads
// the splat with LDS broadcast loads.  Splats that cannot reach any pixel of a
// quadrant cost that wave nothing.
#include <hip/hip_ext.h>

#include <atomic>
#include <cstdlib>
#include <string>
#include <type_traits>

#include "gsr_common.h"

namespace {

constexpr int kBatch = 256;
constexpr float kLog2e = 1.4426950408889634f;

struct Pixel {
  int x, y;       // pixel coordinates
  bool inside;    // inside the image
  int quad;       // quadrant = wave index
};

__device__ __forceinline__ Pixel my_pixel(int W, int H) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  Pixel p;
  p.quad = wave;
  p.x = blockIdx.x * GSR_TILE_X + (wave & 1) * 8 + (lane & 7);
  p.y = blockIdx.y * GSR_TILE_Y + (wave >> 1) * 8 + (lane >> 3);
  p.inside = p.x < W && p.y < H;
  return p;
}

// Does the ellipse A dx^2 + 2B dx dy + C dy^2 <= tau around the splat centre reach the box
// [a0,a1] x [b0,b1] (offsets from the centre)?  The form is convex with its minimum at the centre,
// so over a box that does not hold the centre it is smallest on one of the two edges facing the
// centre: on dx = ax at dy = -B ax / C, on dy = by at dx = -B by / A, each clamped to the edge.
// nBA = -B/A, nBC = -B/C.  Written so that a NaN anywhere answers "yes".
__device__ __forceinline__ bool ellipse_reaches_box(float A, float B, float C, float nBA, float nBC, float tau,
                                                    float a0, float a1, float b0, float b1) {
  const float ax = fminf(fmaxf(0.0f, a0), a1), by = fminf(fmaxf(0.0f, b0), b1);
  const float dy1 = fminf(fmaxf(nBC * ax, b0), b1);
  const float dx2 = fminf(fmaxf(nBA * by, a0), a1);
  const float f1 = A * ax * ax + 2.0f * B * ax * dy1 + C * dy1 * dy1;
  const float f2 = A * dx2 * dx2 + 2.0f * B * dx2 * by + C * by * by;
  return !(fminf(f1, f2) > tau);
}

// 4-bit mask: which 8x8 quadrants of the tile can see a splat centred at (sx,sy) with conic
// (A,B,C) and opacity o.  A pixel blends the splat only where o exp(-q/2) >= 1/255, i.e. inside
// the ellipse q <= 2 ln(255 o); (hx,hy) are the half extents of that ellipse's bounding box
// (negative = never).  Box test first, then the ellipse itself against each quadrant's pixel
// centres; tau is widened a little so that rounding can never make the test exclusive.
__device__ __forceinline__ unsigned quadrant_mask(float sx, float sy, float hx, float hy, float A, float B, float C,
                                                  float o, float tx0, float ty0) {
  if (hx < 0.0f) return 0u;
  const float lx = sx - hx, rx = sx + hx, ly = sy - hy, ry = sy + hy;
  const bool xl = (rx >= tx0) && (lx <= tx0 + 7.0f);
  const bool xr = (rx >= tx0 + 8.0f) && (lx <= tx0 + 15.0f);
  const bool yt = (ry >= ty0) && (ly <= ty0 + 7.0f);
  const bool yb = (ry >= ty0 + 8.0f) && (ly <= ty0 + 15.0f);
  unsigned m = (unsigned)(xl && yt) | ((unsigned)(xr && yt) << 1) | ((unsigned)(xl && yb) << 2) | ((unsigned)(xr && yb) << 3);
  if (m == 0u) return 0u;
  const float tau = 2.0f * __logf(255.0f * o) * 1.001f + 0.01f;
  // (hardware reciprocals: the test only has to err on the side of "reaches", which tau's margin takes care of)
  const float nBA = -B * __builtin_amdgcn_rcpf(A), nBC = -B * __builtin_amdgcn_rcpf(C);
  const float ax0 = tx0 - sx, ay0 = ty0 - sy;
  unsigned e = 0u;
#pragma unroll
  for (int q = 0; q < 4; q++) {
    const float a0 = ax0 + (float)((q & 1) * 8), b0 = ay0 + (float)((q >> 1) * 8);
    e |= (unsigned)ellipse_reaches_box(A, B, C, nBA, nBC, tau, a0, a0 + 7.0f, b0, b0 + 7.0f) << q;
  }
  return m & e;
}

__device__ __forceinline__ unsigned long long lds_mask(const unsigned long long* m) {
  const unsigned long long v = *m;
  const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v);
  const unsigned hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
  return ((unsigned long long)hi << 32) | lo;
}

// Exponent of the Gaussian in the log2 domain, la*dx^2 + lb*dx*dy + lc*dy^2 with
// (la, lb, lc) = (-A/2, -B, -C/2) * log2(e), evaluated as dx * (la*dx + lb*dy) + (lc*dy)*dy: five
// operations, and of those the two products with dy alone are shared by pixels of the same row (the two
// pixels of a lane of the backward).  Forward and backward MUST take the same skip / keep decisions per
// (splat, pixel), so both go through these functions and the rounding sequence is pinned with explicit fma.
__device__ __forceinline__ float gauss_exponent_row(float lc, float dy) { return (lc * dy) * dy; }
__device__ __forceinline__ float gauss_exponent(float la, float lb_dy, float lc_dy_dy, float dx) {
  return fmaf(dx, fmaf(la, dx, lb_dy), lc_dy_dy);
}
__device__ __forceinline__ float gauss_exponent(float la, float lb, float lc, float dx, float dy) {
  return gauss_exponent(la, lb * dy, gauss_exponent_row(lc, dy), dx);
}

// What a staging thread publishes about one splat: one 48-byte LDS slot, so that the waves reading it
// back (every lane the same slot: broadcast reads) need a single address register per splat.
struct __align__(16) Staged {
  float4 geo;      // x, y, la, lb
  float4 col;      // lc, opacity, r, g
  float blue;
  uint32_t tag;    // whatever the walk wants back with the splat (the forward: its 1-based position in the tile's list)
  float pad[2];
};

__device__ __forceinline__ void stage_record(Staged* slot, const uint4 r0, const uint4 r1, const uint4 r2, uint32_t tag) {
  slot->geo = make_float4(__uint_as_float(r0.x), __uint_as_float(r0.y), -0.5f * kLog2e * __uint_as_float(r1.x),
                          -kLog2e * __uint_as_float(r1.y));
  slot->col = make_float4(-0.5f * kLog2e * __uint_as_float(r1.z), __uint_as_float(r1.w), __uint_as_float(r2.x),
                          __uint_as_float(r2.y));
  *reinterpret_cast<float2*>(&slot->blue) = make_float2(__uint_as_float(r2.z), __uint_as_float(tag));
}

// The backward's staging: the same ten words without the padding (the 2 KB saved per workgroup are what lets a
// CU hold eight workgroups of the backward instead of seven), as three arrays so that every read is one aligned
// instruction: 16 bytes, 16 bytes, 8 bytes.  (As one 40-byte slot the words came in as ds_read2_b64 pairs, and a
// ds_read2_b64 takes the LDS about twice as long as a ds_read_b128 of the same 16 bytes.)
struct StagedBack {
  float4 geo[kBatch];    // x, y, -log2e/2 A, -log2e B
  float4 col[kBatch];    // -log2e/2 C, opacity, r, g
  float2 tail[kBatch];   // blue, index bits
};

__device__ __forceinline__ void stage_record(StagedBack& s, int slot, const uint4 r0, const uint4 r1, const uint4 r2, uint32_t id) {
  s.geo[slot] = make_float4(__uint_as_float(r0.x), __uint_as_float(r0.y), -0.5f * kLog2e * __uint_as_float(r1.x),
                            -kLog2e * __uint_as_float(r1.y));
  s.col[slot] = make_float4(-0.5f * kLog2e * __uint_as_float(r1.z), __uint_as_float(r1.w), __uint_as_float(r2.x),
                            __uint_as_float(r2.y));
  s.tail[slot] = make_float2(__uint_as_float(r2.z), __uint_as_float(id));
}

// ---------------------------------------------------------------------------
// forward
// ---------------------------------------------------------------------------
// A zero-fill the blend kernels carry: every workgroup clears `share` 16-byte vectors of the caller's memory
// as its last act, while the kernels leave HBM almost idle.  (Ahead of the workgroup's own work the same
// stores cost the backward twice as much: its next wait for a load then also waits for them.)
struct ShareFill {
  uint4* at;
  size_t vecs;
  uint32_t share;   // vectors per workgroup: ceil(vecs / workgroups)
};

__device__ __forceinline__ void fill_my_share(const ShareFill& f, unsigned group);
__device__ __forceinline__ void fill_my_share(const ShareFill& f) { fill_my_share(f, blockIdx.y * gridDim.x + blockIdx.x); }

__device__ __forceinline__ void fill_my_share(const ShareFill& f, unsigned group) {
  if (f.vecs == 0) return;
  const size_t first = (size_t)group * f.share;
  const size_t end = first + f.share < f.vecs ? first + f.share : f.vecs;
  // Nontemporal stores: the zeros are not wanted in the L2, where the walk's gathered records live (as ordinary
  // stores they cost the backward 7 us of 236 at 1M Gaussians and 90 us of 540 at 6M, and the kernel that later
  // writes gradient rows into the cleared memory a quarter of its time).
  typedef uint32_t u4 __attribute__((ext_vector_type(4)));
  const u4 zero = {0u, 0u, 0u, 0u};
  for (size_t i = first + threadIdx.x; i < end; i += blockDim.x) __builtin_nontemporal_store(zero, reinterpret_cast<u4*>(f.at) + i);
}

// Depth segments of long tile walks.  A tile's backward walk is serial (every step needs the transmittance
// and the colour behind it that the step before left), so one tile whose pixels stay translucent ten times
// deeper than the others' holds the whole kernel up.  The forward therefore writes down every pixel's state
// each time its walk crosses a multiple of `length` list entries while any pixel of the tile is still
// open (a "dump": 256 float4, one per pixel of the tile in row-major order: the transmittance in front of
// the boundary and, once the walk has ended, the colour composited at and behind it), and queues
// (tile, boundary).  The backward then runs one workgroup per queued boundary k on the entries
// [(k - 1) length, k length) of that tile, starting from the dump, beside the tile's own workgroup, which only
// walks the part behind the tile's last dump.  Dumps come from a fixed pool in first-come order; a tile that
// finds it empty (or has taken kMaxTileDumps) simply stops dumping, and its own workgroup walks the rest.  The
// forward's host side chooses `length` >= all list entries / pool size, so the pool cannot run dry (a tile takes
// at most list length / length dumps); below that bound lies only the tests' smaller pool.
struct Segments {
  float4* dumps;          // [cap][256]
  uint2* queue;           // [cap]: tile, boundary (1-based: the boundary in front of entry boundary * length)
  uint32_t* head;         // [0] dumps asked for (zero before the forward's blend; may end above cap), [1] length, [2] cap
  uint32_t* tile_dumps;   // [tiles]: boundaries 1 .. tile_dumps[t] of tile t have a dump
  uint32_t cap, length;   // length: a multiple of 256; 0 = no segments
};
constexpr int kMaxTileDumps = 128;
constexpr uint32_t kSegmentCap = 8192;

inline size_t fill_share(size_t bytes, dim3 grid) {
  const size_t vecs = bytes / 16, groups = (size_t)grid.x * grid.y;
  return (vecs + groups - 1) / groups;
}

inline ShareFill share_fill(void* zero_fill, size_t bytes, dim3 grid) {
  return ShareFill{(uint4*)zero_fill, bytes / 16, (uint32_t)fill_share(bytes, grid)};
}

__global__ void __launch_bounds__(256) blend_fwd_kernel(int W, int H, const uint2* __restrict__ ranges,
                                                        const uint32_t* __restrict__ point_list,
                                                        const uint4* __restrict__ records,
                                                        const float* __restrict__ bg, float* __restrict__ out_color,
                                                        float* __restrict__ final_T, uint32_t* __restrict__ n_contrib,
                                                        ShareFill fill, Segments seg) {
  __shared__ Staged s_splat[kBatch];
  __shared__ unsigned long long s_mask[4][4];  // [quadrant][staging wave]
  __shared__ uint32_t s_slot[kMaxTileDumps];   // the tile's dumps, in boundary order

  const Pixel px = my_pixel(W, H);
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  // The pixel's coordinates as floats are what the splat loop uses; pinned in registers here, because with the
  // integer coordinates also wanted after the loop the compiler otherwise keeps only those and converts them
  // again for every splat (two of the loop's thirty vector instructions).  The integers are recomputed at the end.
  float pxf = (float)px.x, pyf = (float)px.y;
  asm volatile("" : "+v"(pxf), "+v"(pyf));
  const float tx0 = (float)(blockIdx.x * GSR_TILE_X), ty0 = (float)(blockIdx.y * GSR_TILE_Y);
  const uint2 range = ranges[blockIdx.y * gridDim.x + blockIdx.x];
  const int total = (int)(range.y - range.x);
  const int rounds = (total + kBatch - 1) / kBatch;

  // Per-pixel state lives in VGPRs only and the splat body is straight-line, predicated code:
  // a pixel that is finished (its threshold is infinite) or skips a splat simply blends with alpha = 0, which
  // leaves T and the colour sums bit for bit unchanged.  Branching per lane instead costs a
  // dozen scalar mask instructions per splat, and the scalar unit is shared by the CU's 4 SIMDs.
  // the smallest o*G a pixel still accepts: 1/255 while it is live, infinity once it has finished
  float accept = px.inside ? GSR_ALPHA_MIN : __builtin_inff();
  float T = 1.0f, C0 = 0.f, C1 = 0.f, C2 = 0.f;
  uint32_t last = 0;
  // depth segments: the batch in front of which the next dump is due, the dumps taken, this pixel's place in one
  const int seg_rounds = (int)(seg.length / kBatch);
  int next_dump = seg_rounds ? seg_rounds : 0x7fffffff, dumps = 0;
  const uint32_t tile = blockIdx.y * gridDim.x + blockIdx.x;
  // (recomputed from the thread index where it is needed: a value kept alive through the splat loop for these rare
  // uses costs the loop two instructions per splat, the compiler then rematerialises the pixel's coordinates there)
  auto in_tile = [] {
    const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
    return ((w >> 1) * 8 + (l >> 3)) * GSR_TILE_X + (w & 1) * 8 + (l & 7);
  };

  for (int b = 0; b < rounds; b++) {
    const bool wave_done = __ballot(accept < __builtin_inff()) == 0ull;
    if (__syncthreads_and(wave_done)) break;   // also fences the previous batch's LDS reads
#ifndef GSR_EXPERIMENT_NO_FWD_DUMPS
    if (b == next_dump) {                      // some pixel is still open at a segment boundary
      if (threadIdx.x == 0) {
        const uint32_t slot = atomicAdd(seg.head, 1u);
        s_slot[dumps] = slot;
        if (slot < seg.cap) seg.queue[slot] = make_uint2(tile, (uint32_t)dumps + 1u);
      }
      __syncthreads();
      const uint32_t slot = s_slot[dumps];
      if (slot < seg.cap) {
        seg.dumps[(size_t)slot * 256 + in_tile()] = make_float4(T, C0, C1, C2);
        dumps++;
        next_dump = dumps < kMaxTileDumps ? next_dump + seg_rounds : 0x7fffffff;
      } else {
        next_dump = 0x7fffffff;                // the pool is empty: the tile's own backward workgroup walks the rest
      }
    }
#endif

    // stage one batch
    const int s = b * kBatch + (int)threadIdx.x;
    unsigned qm = 0u;
    if (s < total) {
      const uint32_t g = point_list[range.x + s];
      const uint4* rec = records + (size_t)g * 4;
      const uint4 r0 = rec[0], r1 = rec[1], r2 = rec[2];
      const float sx = __uint_as_float(r0.x), sy = __uint_as_float(r0.y);
      qm = quadrant_mask(sx, sy, __uint_as_float(r0.z), __uint_as_float(r0.w), __uint_as_float(r1.x),
                         __uint_as_float(r1.y), __uint_as_float(r1.z), __uint_as_float(r1.w), tx0, ty0);
      stage_record(&s_splat[threadIdx.x], r0, r1, r2, (uint32_t)s + 1u);
    }
#pragma unroll
    for (int q = 0; q < 4; q++) {
      const unsigned long long m = __ballot((qm >> q) & 1u);
      if (lane == 0) s_mask[q][wave] = m;
    }
    __syncthreads();

    if (!wave_done) {
      for (int sw = 0; sw < 4; sw++) {
        unsigned long long m = lds_mask(&s_mask[px.quad][sw]);
        while (m) {
          const int bit = __ffsll((long long)m) - 1;
          m &= ~(1ull << bit);
          const int j = sw * 64 + bit;
          const Staged* sp = &s_splat[j];
          const float4 ga = sp->geo;
          const float4 co = sp->col;
          const float2 tail = *reinterpret_cast<const float2*>(&sp->blue);   // blue, position (one 8-byte read)
          const float blue = tail.x;
          const float dx = ga.x - pxf, dy = ga.y - pyf;
          const float p = gauss_exponent(ga.z, ga.w, co.x, dx, dy);
          // alpha = min(.99, oG) is kept iff the exponent is <= 0 and alpha >= 1/255, i.e. oG >= 1/255
          const float oG = co.y * __builtin_amdgcn_exp2f(p);
          const bool keep = (p <= 0.0f) & (oG >= accept);
          // Pixels that do not take the splat keep computing with whatever their alpha is; every
          // state update below selects on the decision masks, so none of it lands.  (No early "does any
          // pixel of the quadrant take it" test: the vote and its branch cost every splat more than the
          // few instructions below cost the rare splat that none takes.)
          const float alpha = __builtin_amdgcn_fmed3f(oG, 0.0f, GSR_ALPHA_MAX);   // min(.99, oG) for oG >= 0
          const float test_T = T * (1.0f - alpha);
          const bool stop = keep & (test_T < GSR_T_STOP);      // this splat would take the pixel below 1e-4: it ends here
          const bool blend = keep & !stop;
          const float w = blend ? alpha * T : 0.0f;
          accept = stop ? __builtin_inff() : accept;
          T = blend ? test_T : T;
          C0 = fmaf(co.z, w, C0);
          C1 = fmaf(co.w, w, C1);
          C2 = fmaf(blue, w, C2);
          last = blend ? __float_as_uint(tail.y) : last;
        }
        if (__ballot(accept < __builtin_inff()) == 0ull) break;
      }
    }
  }

#ifndef GSR_EXPERIMENT_NO_FWD_DUMPS
  if (seg.length) {
    // what the backward wants of a dump is the colour composited at and behind the boundary: the final sums
    // minus the sums in front of it (each thread rewrites its own words)
    for (int k = 0; k < dumps; k++) {
      float4* d = seg.dumps + (size_t)s_slot[k] * 256 + in_tile();
      const float4 v = *d;
      *d = make_float4(v.x, C0 - v.y, C1 - v.z, C2 - v.w);
    }
    if (threadIdx.x == 0) {
      seg.tile_dumps[tile] = (uint32_t)dumps;
      if (tile == 0) { seg.head[1] = seg.length; seg.head[2] = seg.cap; }
    }
  }
#endif
  int t = (int)threadIdx.x;
  asm volatile("" : "+v"(t));
  const int ox = (int)blockIdx.x * GSR_TILE_X + ((t >> 6) & 1) * 8 + (t & 7), oy = (int)blockIdx.y * GSR_TILE_Y + (t >> 7) * 8 + ((t & 63) >> 3);
  if (ox < W && oy < H) {
    const size_t pix = (size_t)oy * W + ox, plane = (size_t)W * H;
    final_T[pix] = T;
    n_contrib[pix] = last;
    out_color[pix] = C0 + T * bg[0];
    out_color[plane + pix] = C1 + T * bg[1];
    out_color[2 * plane + pix] = C2 + T * bg[2];
  }
  fill_my_share(fill);
}

// ---------------------------------------------------------------------------
// backward
// ---------------------------------------------------------------------------
// The backward's reduction scratch holds 8 rows of 64 lane values per wave, lane l's value of row k
// at word 64 k + l, with no padding between the rows (LDS is what limits how many workgroups a CU
// holds).  Lane 8k+s of blend_bwd_kernel reads words 4s..4s+3 and 32+4s..32+4s+3 of row k with two
// 16-byte loads; the lanes of rows 2, 3, 6, 7 take the two halves in the other order, which keeps
// each of ds_read_b128's four groups of 16 lanes on 64 distinct banks (rows 2m and 2m+1 would
// otherwise meet in the same 32 banks).
constexpr int kRedStride = 64;

// Stores v0..v7 as rows 0..7 of the wave's reduction scratch at LDS byte offset `base` (wave-uniform)
// with ds_write_addtid_b32: address = M0 + offset + 4 * lane.  The instruction takes no address
// register, so it moves half as many registers to the LDS as ds_write_b32 does, and that transfer, not
// the LDS array, is what a 4-byte store costs (blend_bwd_kernel 0.325 -> 0.315 ms at 1M Gaussians).
// One wait state between the write of M0 and its use.  The scratch is written here only: the
// "memory" clobber orders the stores against the loads around them, and `base` carries the
// scratch's address into the statement, so the compiler has to assume it is written.
__device__ __forceinline__ void store_rows_addtid(unsigned base, float v0, float v1, float v2, float v3, float v4,
                                                  float v5, float v6, float v7) {
  asm volatile(
      "s_mov_b32 m0, %8\n\t"
      "s_nop 0\n\t"
      "ds_write_addtid_b32 %0 offset:0\n\t"
      "ds_write_addtid_b32 %1 offset:256\n\t"
      "ds_write_addtid_b32 %2 offset:512\n\t"
      "ds_write_addtid_b32 %3 offset:768\n\t"
      "ds_write_addtid_b32 %4 offset:1024\n\t"
      "ds_write_addtid_b32 %5 offset:1280\n\t"
      "ds_write_addtid_b32 %6 offset:1536\n\t"
      "ds_write_addtid_b32 %7 offset:1792"
      :
      : "v"(v0), "v"(v1), "v"(v2), "v"(v3), "v"(v4), "v"(v5), "v"(v6), "v"(v7), "s"(base)
      : "m0", "memory");
}

template <int CTRL>
__device__ __forceinline__ float dpp_add(float v) {
  const int moved = __builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, false);
  return v + __int_as_float(moved);
}

// Sum over the wave's 64 lanes with DPP alone: four steps inside the rows of 16 lanes, then lane 15 of every row
// added into the row behind it, then lane 31 into the two rows behind it: the lanes of the LAST row (48..63) end
// with the wave's total.
__device__ __forceinline__ float sum_into_last_row(float v) {
  v = v + __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xB1, 0xf, 0xf, false));    // quad_perm [1,0,3,2]
  v = v + __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x4E, 0xf, 0xf, false));    // quad_perm [2,3,0,1]
  v = v + __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x141, 0xf, 0xf, false));   // row_half_mirror
  v = v + __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x140, 0xf, 0xf, false));   // row_mirror
  // the two broadcast steps by hand: through the builtin each comes out as three instructions (a zero, a move,
  // an addition).  Two wait states between a vector write of a register and a DPP read of it.
  asm volatile(
      "s_nop 1\n\t"
      "v_add_f32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
      "s_nop 1\n\t"
      "v_add_f32_dpp %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf"
      : "+v"(v));
  return v;
}

__global__ void __launch_bounds__(256) blend_bwd_kernel(int W, int H, const uint2* __restrict__ ranges,
                                                        const uint32_t* __restrict__ point_list,
                                                        const uint4* __restrict__ records,
                                                        const float* __restrict__ bg,
                                                        const float* __restrict__ final_T,
                                                        const uint32_t* __restrict__ n_contrib,
                                                        const float* __restrict__ dL_dpix,
                                                        float* __restrict__ dL_dsplat,
                                                        uint8_t* __restrict__ touched, ShareFill fill) {
  __shared__ __align__(16) StagedBack s_splat;
  __shared__ unsigned long long s_mask[4][4];
  __shared__ int s_last[4];
  __shared__ __align__(16) float s_red[4][8 * kRedStride];   // per wave: 8 rows of 64 lane values

  const Pixel px = my_pixel(W, H);
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const float pxf = (float)px.x, pyf = (float)px.y;
  const float tx0 = (float)(blockIdx.x * GSR_TILE_X), ty0 = (float)(blockIdx.y * GSR_TILE_Y);
  const uint2 range = ranges[blockIdx.y * gridDim.x + blockIdx.x];
  const int total = (int)(range.y - range.x);

  const size_t pix = (size_t)px.y * W + px.x, plane = (size_t)W * H;
  const float Tf = px.inside ? final_T[pix] : 0.f;
  const int last = px.inside ? (int)n_contrib[pix] : 0;
  float dp0 = 0.f, dp1 = 0.f, dp2 = 0.f;
  if (px.inside) { dp0 = dL_dpix[pix]; dp1 = dL_dpix[plane + pix]; dp2 = dL_dpix[2 * plane + pix]; }
  // final transmittance times (background . dL/dcolour): the background's share of every dL/dalpha of the pixel
  const float Tf_bg = Tf * (bg[0] * dp0 + bg[1] * dp1 + bg[2] * dp2);
  // the tile starts at the deepest position any of its pixels used
  int wl = last;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) wl = max(wl, __shfl_xor(wl, o));
  if (lane == 0) s_last[wave] = wl;
  __syncthreads();
  const int wave_last = __builtin_amdgcn_readfirstlane(wl);
  const int tile_last = max(max(s_last[0], s_last[1]), max(s_last[2], s_last[3]));
  if (tile_last == 0 || total == 0) { fill_my_share(fill); return; }

  // behind = the colour composited behind the current splat (everything already visited in this back-to-front
  // walk) dotted with the pixel's dL/dcolour: the colours enter dL/dalpha only through that dot product, so one
  // running scalar stands for the three channel sums.
  float T = Tf, behind = 0.f;
  // the wave's reduction scratch: its LDS byte offset for the stores, and where this lane reads its two quarters of row lane/8
  typedef __attribute__((address_space(3))) float lds_float;
  const unsigned red_base = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)(lds_float*)s_red[wave]);
  const int my_row = lane >> 3, my_first = (my_row & 2) ? 32 : 0, my_seg = 4 * (lane & 7);
  const float4* const red_lo = reinterpret_cast<const float4*>(s_red[wave] + my_row * kRedStride + my_first + my_seg);
  const float4* const red_hi = reinterpret_cast<const float4*>(s_red[wave] + my_row * kRedStride + (my_first ^ 32) + my_seg);

  for (int b = (tile_last - 1) / kBatch; b >= 0; b--) {
    __syncthreads();
    const int s = b * kBatch + (int)threadIdx.x;
    unsigned qm = 0u;
    if (s < total && s < tile_last) {
      const uint32_t g = point_list[range.x + s];
      const uint4* rec = records + (size_t)g * 4;
      const uint4 r0 = rec[0], r1 = rec[1], r2 = rec[2];
      const float sx = __uint_as_float(r0.x), sy = __uint_as_float(r0.y);
      qm = quadrant_mask(sx, sy, __uint_as_float(r0.z), __uint_as_float(r0.w), __uint_as_float(r1.x),
                         __uint_as_float(r1.y), __uint_as_float(r1.z), __uint_as_float(r1.w), tx0, ty0);
      stage_record(s_splat, (int)threadIdx.x, r0, r1, r2, g);
      // only staged splats can receive sums: the per-Gaussian backward need not look at the rows of the others
      if (touched != nullptr && qm != 0u) touched[g] = 1;
    }
#pragma unroll
    for (int q = 0; q < 4; q++) {
      const unsigned long long m = __ballot((qm >> q) & 1u);
      if (lane == 0) s_mask[q][wave] = m;
    }
    __syncthreads();

    if (b * kBatch >= wave_last) continue;   // nothing this deep for this quadrant
    for (int sw = 3; sw >= 0; sw--) {
      unsigned long long m = lds_mask(&s_mask[px.quad][sw]);
      // positions at or beyond the deepest one any pixel of this quadrant used contribute nothing
      const int room = wave_last - (b * kBatch + sw * 64);
      if (room <= 0) continue;
      if (room < 64) m &= (1ull << room) - 1ull;
      while (m) {
        const int bit = 63 - __clzll((long long)m);
        m &= ~(1ull << bit);
        const int j = sw * 64 + bit;
        const int pos = b * kBatch + j;   // 0-based position in the tile list
        const float4 geo = s_splat.geo[j], col = s_splat.col[j];
        const float2 xy = make_float2(geo.x, geo.y), lab = make_float2(geo.z, geo.w), lco = make_float2(col.x, col.y);
        const float dx = xy.x - pxf, dy = xy.y - pyf;
        // same alpha and same keep / skip decision as the forward (min(.99, oG) >= 1/255 exactly when
        // oG >= 1/255), as straight-line code: a pixel that did not blend this splat gets oG = 0,
        // hence alpha = 0, which leaves T and acc unchanged below and zeroes its contributions
        const float p = gauss_exponent(lab.x, lab.y, lco.x, dx, dy);
        float oG = lco.y * __builtin_amdgcn_exp2f(p);
        const bool hit = (p <= 0.0f) & (oG >= GSR_ALPHA_MIN) & (pos < last);
        if (__ballot(hit) == 0ull) continue;
        oG = hit ? oG : 0.0f;
        const float alpha = __builtin_amdgcn_fmed3f(oG, 0.0f, GSR_ALPHA_MAX);   // min(.99, oG) for oG >= 0

        const float inv = __builtin_amdgcn_rcpf(1.0f - alpha);     // alpha == 0: exactly 1
        T = T * inv;                                               // transmittance in front of the splat
        const float dch = alpha * T;
        const float2 bi = s_splat.tail[j];
        const float d = fmaf(bi.x, dp2, fmaf(col.w, dp1, col.z * dp0)) - behind;   // (c - behind) . dL/dcolour
        const float dL_dalpha = fmaf(d, T, -(Tf_bg * inv));
        behind = fmaf(alpha, d, behind);                           // alpha c + (1 - alpha) behind
        // u = dL/dG * G.  Everything geometric is a moment of u over the pixels:
        //   dL/dopacity = sum u / o,  dL/dmean ~ (sum u dx, sum u dy),  dL/dconic ~ (sum u dx^2, ...)
        // the per-splat factors are applied once per Gaussian by the preprocess backward.
        const float u = dL_dalpha * oG;
        const float ux = u * dx, uy = u * dy;
        // Nine sums over the wave's 64 lanes.  Cross-lane VALU operations (DPP) are the expensive
        // way to do this on gfx950 (36 of them cost as much as the rest of the loop), so the sums
        // are transposed through LDS instead: every lane stores eight of its values as columns
        // (conflict-free), then lane 8k+s adds up eight elements of row k, fetched with two 16-byte
        // reads (conflict-free in the order chosen above), and three DPP steps combine the
        // eight lanes of a row.  The ninth sum stays in registers: four DPP steps within the rows of
        // 16 lanes, two row-broadcast DPP steps across them (sum_into_last_row; two v_permlane swaps and their
        // additions here cost the two-pixel kernel 6 % of its time).  A wave's LDS operations execute in order, so
        // no barrier is needed.  (Pairing the eight values through v_permlane32_swap / v_permlane16_swap
        // instead, one swap and one add per pair and level, then four DPP steps on the two registers left,
        // needs no LDS at all and takes the kernel from 0.30 to 0.40 ms: a swap costs over ten cycles.)
        store_rows_addtid(red_base, u, ux, uy, ux * dx, ux * dy, uy * dy, dch * dp0, dch * dp1);
        const float4 a0 = *red_lo, a1 = *red_hi;
        float v = ((a0.x + a0.y) + (a0.z + a0.w)) + ((a1.x + a1.y) + (a1.z + a1.w));
        float v8 = dch * dp2;
        v = dpp_add<0xB1>(v); v = dpp_add<0x4E>(v); v = dpp_add<0x141>(v);       // over the 8 segment lanes
        v8 = sum_into_last_row(v8);
        // lane 8k holds sum k (k < 8), the lanes of the last row hold sum 8: lanes 0,8,..,56 and lane 57 issue ONE
        // wave-instruction on nine consecutive dwords of one 64-byte row = one atomic request
        const bool ninth = lane == 57;
        if ((lane & 7) == 0 || ninth)
          atomicAdd(dL_dsplat + (size_t)__float_as_uint(bi.y) * GSR_GRAD_WORDS + (ninth ? 8 : (lane >> 3)), ninth ? v8 : v);
      }
    }
  }
  fill_my_share(fill);
}


// ---------------------------------------------------------------------------
// backward with two pixels per lane
// ---------------------------------------------------------------------------
// blend_bwd_kernel spends a (wave, splat) turn mostly waiting: three or four LDS round trips and two dozen
// dependent cross-lane steps in a row for some sixty vector instructions, at the hardware's limit of eight waves
// per SIMD.  Here a wave owns a 16 x 8 half tile instead of an 8 x 8 quadrant, a lane the pixels (x, y) and
// (x + 8, y): two independent chains of arithmetic per turn (real instruction-level parallelism: the two pixels
// share nothing but the splat), ONE set of LDS reads, ONE reduction and ONE atomic for 128 pixels, and because
// the two pixels have the same dy, three of the six geometric moments of the pair are products of the other
// three with dy.  The price is coarser culling: a splat that reaches one of the two quadrants costs both.
// A workgroup is two waves and stages 128 splats per batch.
constexpr int kPairBatch = 128;

struct StagedPair {
  float4 geo[kPairBatch];    // x, y, -log2e/2 A, -log2e B
  float4 col[kPairBatch];    // -log2e/2 C, opacity, r, g
  float2 tail[kPairBatch];   // blue, index bits
};

// The grid is one-dimensional: first the `extras` workgroups of the queued depth segments (see Segments; the
// longest items, so they start first), then one workgroup per tile in row-major order.
__global__ void __launch_bounds__(128) blend_bwd_pair_kernel(int W, int H, const uint2* __restrict__ ranges,
                                                             const uint32_t* __restrict__ point_list,
                                                             const uint4* __restrict__ records,
                                                             const float* __restrict__ bg,
                                                             const float* __restrict__ final_T,
                                                             const uint32_t* __restrict__ n_contrib,
                                                             const float* __restrict__ dL_dpix,
                                                             float* __restrict__ dL_dsplat,
                                                             uint8_t* __restrict__ touched, ShareFill fill, Segments seg,
                                                             int tiles_x, int tiles, int extras) {
  __shared__ __align__(16) StagedPair s_splat;
  __shared__ unsigned long long s_mask[2][2];     // [half tile][staging wave]
  __shared__ int s_last[2];
  __shared__ __align__(16) float s_red[2][8 * kRedStride];

  // this workgroup's item: entries [lo, hi) of one tile's list, from the dump in front of hi if there is one
  uint32_t tile;
  int lo = 0, hi = 0x7fffffff;
  const float4* dump = nullptr;
  if ((int)blockIdx.x < extras) {
    const uint32_t queued = min(seg.head[0], min(seg.head[2], seg.cap)), length = seg.head[1];
    const uint2 q = (uint32_t)blockIdx.x < queued ? seg.queue[blockIdx.x] : make_uint2(0xFFFFFFFFu, 0u);
    if (q.x >= (uint32_t)tiles || q.y == 0u || q.y > (uint32_t)kMaxTileDumps) return;
    tile = q.x;
    hi = (int)(q.y * length);
    lo = hi - (int)length;
    dump = seg.dumps + (size_t)blockIdx.x * 256;
  } else {
    tile = blockIdx.x - (uint32_t)extras;
    if (extras) lo = (int)(seg.tile_dumps[tile] * seg.head[1]);
  }
  const int tile_x = (int)(tile % (uint32_t)tiles_x), tile_y = (int)(tile / (uint32_t)tiles_x);

  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;     // wave 0: rows 0..7 of the tile, wave 1: rows 8..15
  const int x0 = tile_x * GSR_TILE_X + (lane & 7), x1 = x0 + 8, y = tile_y * GSR_TILE_Y + wave * 8 + (lane >> 3);
  const bool in0 = x0 < W && y < H, in1 = x1 < W && y < H;
  const float x0f = (float)x0, x1f = (float)x1, yf = (float)y;
  const float tx0 = (float)(tile_x * GSR_TILE_X), ty0 = (float)(tile_y * GSR_TILE_Y);
  const uint2 range = ranges[tile];
  const int total = (int)(range.y - range.x);

  const size_t plane = (size_t)W * H, pix0 = (size_t)y * W + x0, pix1 = pix0 + 8;
  const float Tf0 = in0 ? final_T[pix0] : 0.f, Tf1 = in1 ? final_T[pix1] : 0.f;
  const int last0 = in0 ? (int)n_contrib[pix0] : 0, last1 = in1 ? (int)n_contrib[pix1] : 0;
  float r0 = 0.f, g0 = 0.f, b0 = 0.f, r1 = 0.f, g1 = 0.f, b1 = 0.f;      // dL/dcolour of the two pixels
  if (in0) { r0 = dL_dpix[pix0]; g0 = dL_dpix[plane + pix0]; b0 = dL_dpix[2 * plane + pix0]; }
  if (in1) { r1 = dL_dpix[pix1]; g1 = dL_dpix[plane + pix1]; b1 = dL_dpix[2 * plane + pix1]; }
  const float bgr = bg[0], bgg = bg[1], bgb = bg[2];
  const float Tf_bg0 = Tf0 * (bgr * r0 + bgg * g0 + bgb * b0), Tf_bg1 = Tf1 * (bgr * r1 + bgg * g1 + bgb * b1);
  int wl = max(last0, last1), wf = min(last0, last1);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { wl = max(wl, __shfl_xor(wl, o)); wf = min(wf, __shfl_xor(wf, o)); }
  if (lane == 0) s_last[wave] = wl;
  __syncthreads();
  // entries in front of this position are blended by every pixel of the half tile (0 where the tile leaves the image)
  const int wave_first_end = __builtin_amdgcn_readfirstlane(wf);
  // the deepest position this item visits (exclusive), for the half tile and for the tile
  const int wave_last = min(__builtin_amdgcn_readfirstlane(wl), hi);
  const int tile_last = min(max(s_last[0], s_last[1]), hi);
  // the zero-fill is the tiles' workgroups' (a workgroup of the queue that finds no item would clear its share
  // at once, ahead of everybody's work, where the stores are in the way)
  const bool fills = (int)blockIdx.x >= extras;
  if (tile_last <= lo || total == 0) { if (fills) fill_my_share(fill, tile); return; }

  // A pixel that ended in front of the boundary has its final state in the dump (transmittance = final_T,
  // nothing behind), one that went on has there what the walk from its own end would have brought here.
  float T0 = Tf0, T1 = Tf1, behind0 = 0.f, behind1 = 0.f;
  if (dump != nullptr) {
    const float4 d0 = dump[(wave * 8 + (lane >> 3)) * GSR_TILE_X + (lane & 7)], d1 = dump[(wave * 8 + (lane >> 3)) * GSR_TILE_X + (lane & 7) + 8];
    T0 = d0.x;
    T1 = d1.x;
    behind0 = (d0.y * r0 + d0.z * g0 + d0.w * b0) / d0.x;
    behind1 = (d1.y * r1 + d1.z * g1 + d1.w * b1) / d1.x;
  }
  typedef __attribute__((address_space(3))) float lds_float;
  const unsigned red_base = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)(lds_float*)s_red[wave]);
  const int my_row = lane >> 3, my_first = (my_row & 2) ? 32 : 0, my_seg = 4 * (lane & 7);
  const float4* const red_lo = reinterpret_cast<const float4*>(s_red[wave] + my_row * kRedStride + my_first + my_seg);
  const float4* const red_hi = reinterpret_cast<const float4*>(s_red[wave] + my_row * kRedStride + (my_first ^ 32) + my_seg);

  const bool ninth = lane == 57, adds = (lane & 7) == 0 || ninth;    // lanes 0, 8, .., 56 hold sums 0..7, the last row's lanes the ninth
  const int my_word = ninth ? 8 : (lane >> 3);

  for (int b = (tile_last - 1) / kPairBatch; b >= lo / kPairBatch; b--) {
    __syncthreads();
    const int s = b * kPairBatch + (int)threadIdx.x;
    unsigned qm = 0u;
    if (s < total && s < tile_last) {
      const uint32_t g = point_list[range.x + s];
      const uint4* rec = records + (size_t)g * 4;
      const uint4 q0 = rec[0], q1 = rec[1], q2 = rec[2];
      const float sx = __uint_as_float(q0.x), sy = __uint_as_float(q0.y);
      qm = quadrant_mask(sx, sy, __uint_as_float(q0.z), __uint_as_float(q0.w), __uint_as_float(q1.x),
                         __uint_as_float(q1.y), __uint_as_float(q1.z), __uint_as_float(q1.w), tx0, ty0);
      s_splat.geo[threadIdx.x] = make_float4(sx, sy, -0.5f * kLog2e * __uint_as_float(q1.x), -kLog2e * __uint_as_float(q1.y));
      s_splat.col[threadIdx.x] = make_float4(-0.5f * kLog2e * __uint_as_float(q1.z), __uint_as_float(q1.w),
                                             __uint_as_float(q2.x), __uint_as_float(q2.y));
      s_splat.tail[threadIdx.x] = make_float2(__uint_as_float(q2.z), __uint_as_float(g));
      if (touched != nullptr && qm != 0u) touched[g] = 1;
    }
    {
      const unsigned long long top = __ballot((qm & 3u) != 0u), bottom = __ballot((qm & 12u) != 0u);
      if (lane == 0) { s_mask[0][wave] = top; s_mask[1][wave] = bottom; }
    }
    __syncthreads();

    if (b * kPairBatch >= wave_last) continue;   // nothing this deep for this half tile
    // one (wave, splat) turn; kCheckEnds = false where every pixel of the half tile is known to go deeper than the
    // splat (then the two comparisons with the pixels' own ends are left out)
    auto turn = [&](const int j, auto check_ends) {
      constexpr bool kCheckEnds = decltype(check_ends)::value;
      const int pos = b * kPairBatch + j;
      const float4 geo = s_splat.geo[j], col = s_splat.col[j];
      // the same subtractions and the same exponent as the forward's, per pixel
      const float dxa = geo.x - x0f, dxb = geo.x - x1f, dy = geo.y - yf;
      const float row = gauss_exponent_row(col.x, dy), lb_dy = geo.w * dy;
      const float pa = gauss_exponent(geo.z, lb_dy, row, dxa), pb = gauss_exponent(geo.z, lb_dy, row, dxb);
      float oGa = col.y * __builtin_amdgcn_exp2f(pa), oGb = col.y * __builtin_amdgcn_exp2f(pb);
      const bool hita = (pa <= 0.0f) & (oGa >= GSR_ALPHA_MIN) & (!kCheckEnds || pos < last0);
      const bool hitb = (pb <= 0.0f) & (oGb >= GSR_ALPHA_MIN) & (!kCheckEnds || pos < last1);
      if (__ballot(hita | hitb) == 0ull) return;
      oGa = hita ? oGa : 0.0f;
      oGb = hitb ? oGb : 0.0f;
      const float alphaa = __builtin_amdgcn_fmed3f(oGa, 0.0f, GSR_ALPHA_MAX), alphab = __builtin_amdgcn_fmed3f(oGb, 0.0f, GSR_ALPHA_MAX);
      const float inva = __builtin_amdgcn_rcpf(1.0f - alphaa), invb = __builtin_amdgcn_rcpf(1.0f - alphab);
      T0 *= inva;
      T1 *= invb;
      const float2 bi = s_splat.tail[j];
      const float da = fmaf(bi.x, b0, fmaf(col.w, g0, col.z * r0)) - behind0;
      const float db = fmaf(bi.x, b1, fmaf(col.w, g1, col.z * r1)) - behind1;
      const float ua = fmaf(da, T0, -(Tf_bg0 * inva)) * oGa, ub = fmaf(db, T1, -(Tf_bg1 * invb)) * oGb;
      behind0 = fmaf(alphaa, da, behind0);
      behind1 = fmaf(alphab, db, behind1);
      const float wa = alphaa * T0, wb = alphab * T1;
      // the pair's nine contributions; dy is common to the two pixels
      const float u = ua + ub;
      const float uxa = ua * dxa, uxb = ub * dxb;
      const float ux = uxa + uxb;
      const float uxx = fmaf(uxb, dxb, uxa * dxa);
      const float uy = u * dy;
      // From here to the atomic the wave goes ahead of the SIMD's other waves: the reduction is a chain of dependent
      // steps (LDS round trip, nine cross-lane additions), and a wave that is served late at each of them holds its
      // turn's end, and with it the next turn's start, back (-2.8 % at 1M Gaussians, -3.7 % without early
      // termination; raised from the vote on, or to other levels, the same or less).
      __builtin_amdgcn_s_setprio(2);
      store_rows_addtid(red_base, u, ux, uy, uxx, ux * dy, uy * dy, fmaf(wb, r1, wa * r0), fmaf(wb, g1, wa * g0));
      const float4 a0 = *red_lo, a1 = *red_hi;
      float v = ((a0.x + a0.y) + (a0.z + a0.w)) + ((a1.x + a1.y) + (a1.z + a1.w));
      float v8 = fmaf(wb, b1, wa * b0);
      v = dpp_add<0xB1>(v); v = dpp_add<0x4E>(v); v = dpp_add<0x141>(v);
      // (the sum is wanted by a few lanes only, and left to itself the compiler moves the last addition behind
      // their branch, where it cannot be part of the cross-lane instruction any more: three instructions for one)
      asm volatile("" : "+v"(v));
      v8 = sum_into_last_row(v8);
      // the row's address as a scalar (every lane holds the same index), the lane's word as a constant offset
      float* const grad_row = dL_dsplat + (size_t)(uint32_t)__builtin_amdgcn_readfirstlane(__float_as_uint(bi.y)) * GSR_GRAD_WORDS;
      if (adds) atomicAdd(grad_row + my_word, ninth ? v8 : v);
      __builtin_amdgcn_s_setprio(0);
    };
    for (int sw = 1; sw >= 0; sw--) {
      unsigned long long m = lds_mask(&s_mask[wave][sw]);
      const int room = wave_last - (b * kPairBatch + sw * 64);
      if (room <= 0) continue;
      if (room < 64) m &= (1ull << room) - 1ull;
      if (b * kPairBatch + sw * 64 + 64 <= wave_first_end) {
        while (m) {
          const int bit = 63 - __clzll((long long)m);
          m &= ~(1ull << bit);
          turn(sw * 64 + bit, std::false_type{});
        }
      } else {
        while (m) {
          const int bit = 63 - __clzll((long long)m);
          m &= ~(1ull << bit);
          turn(sw * 64 + bit, std::true_type{});
        }
      }
    }
  }
  if (fills) fill_my_share(fill, tile);
}

}  // namespace

namespace {

// Entries per depth segment (0 = none); gsr_debug_segment_length changes it.  2048: deeper than any tile of an
// ordinary scene is walked (a few hundred entries; 2400 at six million Gaussians with a tenth of the usual
// opacity), so only walks that would otherwise be the kernel's tail are cut.
std::atomic<uint32_t> g_segment_length{2048u};
std::atomic<uint32_t> g_segment_pool{kSegmentCap};     // dumps a forward may take; tests make it small

// The block gsr_forward reserves in its workspace, in the order dumps | queue | tile_dumps | head; the head's
// 256 bytes come last so that whoever clears memory right behind the block can clear the head with it.
struct SegmentsLayout {
  size_t dumps, queue, tile_dumps, head, total;
};

SegmentsLayout segments_layout(int W, int H) {
  const size_t tiles = (size_t)gsr::div_up(W, GSR_TILE_X) * gsr::div_up(H, GSR_TILE_Y);
  SegmentsLayout L;
  size_t o = 0;
  auto take = [&](size_t bytes) { const size_t at = o; o += gsr::align_up(bytes, 256); return at; };
  L.dumps = take((size_t)kSegmentCap * 256 * sizeof(float4));
  L.queue = take((size_t)kSegmentCap * sizeof(uint2));
  L.tile_dumps = take(tiles * 4);
  L.head = take(256);
  L.total = o;
  return L;
}

Segments segments_at(void* block, int W, int H, uint32_t length, uint32_t pool) {
  if (block == nullptr || length == 0u) return Segments{nullptr, nullptr, nullptr, nullptr, 0u, 0u};
  const SegmentsLayout L = segments_layout(W, H);
  char* b = (char*)block;
  return Segments{(float4*)(b + L.dumps), (uint2*)(b + L.queue), (uint32_t*)(b + L.head), (uint32_t*)(b + L.tile_dumps),
                  pool < kSegmentCap ? pool : kSegmentCap, length};
}

// With events the launch goes through hipExtLaunchKernelGGL, which stamps them with the kernel's own
// start and end (what a profiler's kernel trace reports), instead of with the stream's progress around it.
int launch_blend_fwd(int W, int H, const uint32_t* ranges, const uint32_t* point_list, const uint32_t* records,
                     const float* bg, float* out_color, float* final_T, uint32_t* n_contrib, void* zero_fill,
                     size_t zero_fill_bytes, void* segments, uint32_t segment_length, hipEvent_t start, hipEvent_t stop,
                     void* stream) {
  GSR_REQUIRE(W > 0 && H > 0, "blend_fwd: bad image size %dx%d", W, H);
  GSR_REQUIRE(zero_fill_bytes == 0 || (zero_fill && ((uintptr_t)zero_fill & 15) == 0 && zero_fill_bytes % 16 == 0),
              "blend_fwd: zero_fill must be 16-byte aligned and a multiple of 16 bytes");
  GSR_REQUIRE(ranges && bg && out_color && final_T && n_contrib, "blend_fwd: null pointer");
  GSR_REQUIRE(((uintptr_t)segments & 255) == 0 && (segments == nullptr || (segment_length != 0u && segment_length % 256u == 0u)),
              "blend_fwd: the segments block must be 256-byte aligned, the segment length a multiple of 256");
  dim3 grid(gsr::div_up(W, GSR_TILE_X), gsr::div_up(H, GSR_TILE_Y)), block(256);
  GSR_REQUIRE(fill_share(zero_fill_bytes, grid) <= 0xFFFFFFFFull, "blend: zero_fill of %zu bytes is too large for %u workgroups",
              zero_fill_bytes, grid.x * grid.y);
  const ShareFill fill = share_fill(zero_fill, zero_fill_bytes, grid);
  const Segments seg = segments_at(segments, W, H, segment_length, g_segment_pool.load(std::memory_order_relaxed));
  if (start || stop)
    hipExtLaunchKernelGGL(blend_fwd_kernel, grid, block, 0, (hipStream_t)stream, start, stop, 0, W, H, (const uint2*)ranges,
                          point_list, (const uint4*)records, bg, out_color, final_T, n_contrib, fill, seg);
  else
    hipLaunchKernelGGL(blend_fwd_kernel, grid, block, 0, (hipStream_t)stream, W, H, (const uint2*)ranges, point_list,
                       (const uint4*)records, bg, out_color, final_T, n_contrib, fill, seg);
  return gsr::check_launch("blend_fwd");
}

int launch_blend_bwd(int W, int H, const uint32_t* ranges, const uint32_t* point_list, const uint32_t* records,
                     const float* bg, const float* final_T, const uint32_t* n_contrib, const float* dL_dpix,
                     float* dL_dsplat, uint8_t* touched, void* zero_fill, size_t zero_fill_bytes, const void* segments,
                     hipEvent_t start, hipEvent_t stop, void* stream) {
  GSR_REQUIRE(W > 0 && H > 0, "blend_bwd: bad image size %dx%d", W, H);
  GSR_REQUIRE(zero_fill_bytes == 0 || (zero_fill && ((uintptr_t)zero_fill & 15) == 0 && zero_fill_bytes % 16 == 0),
              "blend_bwd: zero_fill must be 16-byte aligned and a multiple of 16 bytes");
  GSR_REQUIRE(ranges && bg && final_T && n_contrib && dL_dpix && dL_dsplat, "blend_bwd: null pointer");
  GSR_REQUIRE(((uintptr_t)dL_dsplat & 63) == 0, "blend_bwd: dL_dsplat must be 64-byte aligned");
  GSR_REQUIRE(((uintptr_t)segments & 255) == 0, "blend_bwd: the segments block must be 256-byte aligned");
  const dim3 tiles2d(gsr::div_up(W, GSR_TILE_X), gsr::div_up(H, GSR_TILE_Y));
  // GSR_BLEND_BWD=lanes: the one-pixel-per-lane formulation (four waves per tile, whole walks only), for
  // measurements of one against the other and as an independent implementation in the tests
  static const bool lanes = [] {
    const char* e = getenv("GSR_BLEND_BWD");
    return e && std::string(e) == "lanes";
  }();
  if (lanes) {
    GSR_REQUIRE(fill_share(zero_fill_bytes, tiles2d) <= 0xFFFFFFFFull, "blend: zero_fill of %zu bytes is too large for %u workgroups",
                zero_fill_bytes, tiles2d.x * tiles2d.y);
    const ShareFill fill = share_fill(zero_fill, zero_fill_bytes, tiles2d);
    if (start || stop)
      hipExtLaunchKernelGGL(blend_bwd_kernel, tiles2d, dim3(256), 0, (hipStream_t)stream, start, stop, 0, W, H, (const uint2*)ranges,
                            point_list, (const uint4*)records, bg, final_T, n_contrib, dL_dpix, dL_dsplat, touched, fill);
    else
      hipLaunchKernelGGL(blend_bwd_kernel, tiles2d, dim3(256), 0, (hipStream_t)stream, W, H, (const uint2*)ranges, point_list,
                         (const uint4*)records, bg, final_T, n_contrib, dL_dpix, dL_dsplat, touched, fill);
    return gsr::check_launch("blend_bwd");
  }
  // the segment length the kernel works with is the one the forward left in the block
  const Segments seg = segments_at(const_cast<void*>(segments), W, H, 256u, kSegmentCap);
  const int tiles = (int)(tiles2d.x * tiles2d.y), extras = segments ? (int)kSegmentCap : 0;
  const dim3 grid((unsigned)(tiles + extras));
  GSR_REQUIRE(fill_share(zero_fill_bytes, tiles2d) <= 0xFFFFFFFFull, "blend: zero_fill of %zu bytes is too large for %u workgroups",
              zero_fill_bytes, tiles2d.x * tiles2d.y);
  const ShareFill fill = share_fill(zero_fill, zero_fill_bytes, tiles2d);
  if (start || stop)
    hipExtLaunchKernelGGL(blend_bwd_pair_kernel, grid, dim3(128), 0, (hipStream_t)stream, start, stop, 0, W, H,
                          (const uint2*)ranges, point_list, (const uint4*)records, bg, final_T, n_contrib, dL_dpix, dL_dsplat,
                          touched, fill, seg, (int)tiles2d.x, tiles, extras);
  else
    hipLaunchKernelGGL(blend_bwd_pair_kernel, grid, dim3(128), 0, (hipStream_t)stream, W, H, (const uint2*)ranges, point_list,
                       (const uint4*)records, bg, final_T, n_contrib, dL_dpix, dL_dsplat, touched, fill, seg, (int)tiles2d.x,
                       tiles, extras);
  return gsr::check_launch("blend_bwd");
}

}  // namespace

extern "C" void gsr_debug_segments(uint32_t entries, uint32_t pool) {
  g_segment_length.store((entries + 255u) / 256u * 256u, std::memory_order_relaxed);
  g_segment_pool.store(pool ? pool : kSegmentCap, std::memory_order_relaxed);
}

// Entries per segment for lists of N entries in all: the configured length, or as many more as it takes for the
// pool to hold every boundary (sum over the tiles of floor(list length / length) <= N / length <= kSegmentCap).
// A longer segment than configured costs nothing there: with N / kSegmentCap entries per segment and a few
// thousand workgroups resident, the longest item is still a fraction of a workgroup slot's share of the work.
uint32_t gsr::segment_length_for(uint32_t N) {
  const uint32_t configured = g_segment_length.load(std::memory_order_relaxed);
  if (configured == 0u) return 0u;
  const uint32_t fits = (N / kSegmentCap + 256u) / 256u * 256u;
  return configured > fits ? configured : fits;
}

size_t gsr::segments_bytes(int W, int H) { return segments_layout(W, H).total; }

uint32_t* gsr::segments_head(void* block, int W, int H) {
  return block ? (uint32_t*)((char*)block + segments_layout(W, H).head) : nullptr;
}

extern "C" int gsr_blend_fwd(int W, int H, const uint32_t* ranges, const uint32_t* point_list,
                             const uint32_t* records, const float* bg, float* out_color, float* final_T,
                             uint32_t* n_contrib, void* stream) {
  return launch_blend_fwd(W, H, ranges, point_list, records, bg, out_color, final_T, n_contrib, nullptr, 0, nullptr, 0u, nullptr,
                          nullptr, stream);
}

extern "C" int gsr_blend_bwd(int W, int H, const uint32_t* ranges, const uint32_t* point_list,
                             const uint32_t* records, const float* bg, const float* final_T,
                             const uint32_t* n_contrib, const float* dL_dpix, float* dL_dsplat, void* stream) {
  return launch_blend_bwd(W, H, ranges, point_list, records, bg, final_T, n_contrib, dL_dpix, dL_dsplat, nullptr, nullptr, 0,
                          nullptr, nullptr, nullptr, stream);
}

extern "C" int gsr_blend_fwd_timed(int W, int H, const uint32_t* ranges, const uint32_t* point_list,
                                   const uint32_t* records, const float* bg, float* out_color, float* final_T,
                                   uint32_t* n_contrib, void* start_event, void* stop_event, void* stream) {
  return launch_blend_fwd(W, H, ranges, point_list, records, bg, out_color, final_T, n_contrib, nullptr, 0, nullptr, 0u,
                          (hipEvent_t)start_event, (hipEvent_t)stop_event, stream);
}

int gsr::blend_fwd(int W, int H, const uint32_t* ranges, const uint32_t* point_list, const uint32_t* records,
                   const float* bg, float* out_color, float* final_T, uint32_t* n_contrib, void* zero_fill,
                   size_t zero_fill_bytes, void* segments, uint32_t segment_length, void* start_event, void* stop_event,
                   hipStream_t stream) {
  return launch_blend_fwd(W, H, ranges, point_list, records, bg, out_color, final_T, n_contrib, zero_fill, zero_fill_bytes,
                          segments, segment_length, (hipEvent_t)start_event, (hipEvent_t)stop_event, stream);
}

extern "C" size_t gsr_segments_bytes(int W, int H) { return (W > 0 && H > 0) ? segments_layout(W, H).total : 0; }

extern "C" int gsr_blend_fwd_segments(int W, int H, const uint32_t* ranges, const uint32_t* point_list,
                                      const uint32_t* records, const float* bg, float* out_color, float* final_T,
                                      uint32_t* n_contrib, void* segments, size_t segments_bytes, uint32_t N,
                                      void* start_event, void* stop_event, void* stream) {
  GSR_REQUIRE(segments != nullptr && W > 0 && H > 0 && segments_bytes >= segments_layout(W, H).total,
              "blend_fwd_segments: the segments block is missing or smaller than gsr_segments_bytes");
  const uint32_t length = gsr::segment_length_for(N);
  return launch_blend_fwd(W, H, ranges, point_list, records, bg, out_color, final_T, n_contrib, nullptr, 0,
                          length ? segments : nullptr, length, (hipEvent_t)start_event, (hipEvent_t)stop_event, stream);
}

extern "C" int gsr_blend_bwd_timed(int W, int H, const uint32_t* ranges, const uint32_t* point_list,
                                   const uint32_t* records, const float* bg, const float* final_T,
                                   const uint32_t* n_contrib, const float* dL_dpix, float* dL_dsplat,
                                   void* start_event, void* stop_event, void* stream) {
  return launch_blend_bwd(W, H, ranges, point_list, records, bg, final_T, n_contrib, dL_dpix, dL_dsplat, nullptr, nullptr, 0,
                          nullptr, (hipEvent_t)start_event, (hipEvent_t)stop_event, stream);
}

extern "C" int gsr_blend_bwd_fill(int W, int H, const uint32_t* ranges, const uint32_t* point_list,
                                  const uint32_t* records, const float* bg, const float* final_T,
                                  const uint32_t* n_contrib, const float* dL_dpix, float* dL_dsplat, uint8_t* touched,
                                  void* zero_fill, size_t zero_fill_bytes, void* start_event, void* stop_event,
                                  void* stream) {
  return launch_blend_bwd(W, H, ranges, point_list, records, bg, final_T, n_contrib, dL_dpix, dL_dsplat, touched, zero_fill,
                          zero_fill_bytes, nullptr, (hipEvent_t)start_event, (hipEvent_t)stop_event, stream);
}

extern "C" int gsr_blend_bwd_segments(int W, int H, const uint32_t* ranges, const uint32_t* point_list,
                                      const uint32_t* records, const float* bg, const float* final_T,
                                      const uint32_t* n_contrib, const float* dL_dpix, float* dL_dsplat,
                                      uint8_t* touched, void* zero_fill, size_t zero_fill_bytes, const void* segments,
                                      void* start_event, void* stop_event, void* stream) {
  return launch_blend_bwd(W, H, ranges, point_list, records, bg, final_T, n_contrib, dL_dpix, dL_dsplat, touched, zero_fill,
                          zero_fill_bytes, segments, (hipEvent_t)start_event, (hipEvent_t)stop_event, stream);
}
