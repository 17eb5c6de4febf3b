// Tile binning for gfx950: builds, for every 16x16 tile, the front-to-back list
// of Gaussians whose tile rectangle covers it — entry for entry the list that a
// stable sort of (tile << 32 | depth) keys, emitted per Gaussian in ascending
// index and row-major tile order, produces.  Ties in (tile, depth) keep
// ascending Gaussian index.
//
// Sorting the N (tile, Gaussian) instances themselves (N ~ 20-35x the Gaussian
// count) is pure HBM traffic: 6+ passes over 12-byte pairs.  Here only the P
// Gaussians are sorted, once, by depth; everything after that is a cascade of
// STABLE RANGE PARTITIONS that exploits the rectangle structure.  A range
// partition takes an ordered list whose items each cover a small box of
// buckets and produces, per bucket, the ordered sub-list of the items covering
// it.  A block stages a piece of the list in LDS; each of its waves owns some
// buckets and sweeps the piece 64 items at a time: one ballot tells which items
// cover the bucket, its popcount is the count, the bits below a lane are that
// lane's rank.  A counting sweep, one prefix sum over (bucket, piece), then a
// writing sweep that repeats the ballots.  Three levels:
//   1. the depth-sorted Gaussians -> one list per SUPERCELL of 16x16 tiles
//      (40 at 1080p; entries carry index + rectangle local to the supercell);
//   2. every supercell list -> one list per CELL = tile row x strip of 16 tiles
//      (entries carry index + x range);
//   3. every cell list -> the 16 tile lists of the cell, written straight into
//      point_list in row-major tile order; the tile ranges fall out of the
//      prefix sum.
// Every level keeps the input order, so every tile list is in depth order.
// Work per level ~ items x buckets / 64 wave steps; HBM traffic ~ 4 N bytes
// (the list itself) + 16 bytes per cell entry.
//
// Nothing is gathered by index: the depth sort carries, as its 64-bit value, the
// Gaussian's index packed with its tile rectangle (bit widths chosen per call
// from P and the tile grid), and that word is what levels 1 and 2 read.
//
// The depth sort itself lives in depthsort.hip.
#include <cstring>

#include "gsr_common.h"

namespace {

constexpr int kBlock = 256;
constexpr int kSuper = GSR_SUPERCELL;      // supercell side in tiles = tiles per cell
constexpr uint32_t kPiece1 = 2048;         // list entries per block, per level
constexpr uint32_t kPiece2 = 2048;
constexpr uint32_t kPiece3 = 4096;
constexpr int kScanBlock = 1024;

typedef unsigned long long u64;

struct Grid {
  int gx, gy, sx, sy;             // tiles per row, tile rows, supercells per row, supercell rows
  __host__ __device__ int supercells() const { return sx * sy; }
  __host__ __device__ int cells() const { return sx * gy; }
  __host__ __device__ int tiles() const { return gx * gy; }
};

inline Grid make_grid(int W, int H) {
  Grid g;
  g.gx = gsr::div_up(W, GSR_TILE_X);
  g.gy = gsr::div_up(H, GSR_TILE_Y);
  g.sx = gsr::div_up(g.gx, kSuper);
  g.sy = gsr::div_up(g.gy, kSuper);
  return g;
}

inline uint32_t div_up_u(uint32_t a, uint32_t b) { return (a + b - 1) / b; }

using gsr::Pack;

// ---------------------------------------------------------------------------
// totals: N = tile instances, E1 = supercell entries, E2 = cell entries, V = visible Gaussians
// ---------------------------------------------------------------------------
// Sums the per-block partial records the preprocess geometry kernel leaves behind.
// totals[4] = the caller's status word as it stands (0 without one): it rides to the host with the sizes;
// totals[5] = 1 if the frame has more tile instances than the 32-bit sizes hold (gsr::kMaxInstances).
__global__ void __launch_bounds__(kScanBlock) totals_final_kernel(int blocks, const uint32_t* __restrict__ partials,
                                                                  const uint32_t* __restrict__ status,
                                                                  uint32_t* __restrict__ totals) {
  __shared__ uint32_t s_part[kScanBlock / 64][5];
  gsr::sum_totals<kScanBlock>(blocks, partials, status, totals, s_part);
}

// ---------------------------------------------------------------------------
// range partition, one level
// ---------------------------------------------------------------------------
// A piece is a run of at most kPiece consecutive entries of one parent list.  Its counts
// live in counts[slot0 + f * np] for bucket f, np = pieces of the parent, so that ONE
// exclusive prefix sum over the whole array, which is ordered (parent, bucket, piece),
// turns every count into the absolute output position of that piece's first entry for
// that bucket.
// What the host believed when it sized the buffers, against what the device knows.  The list sizes
// (N, E1, E2, V) travel to the host and back as plain numbers, and the depth sort may have given up:
// a builder that ran on other numbers than the preprocess produced, or on an incomplete sort, would
// write past its buffers.  The three level-1 kernels check; a refusing scan1_kernel leaves no level-2
// pieces behind, which empties every later kernel of the chain, and all tile ranges come out (0,0).
struct Guard {
  const uint32_t* sort_error;   // non-zero = the depth sort that produced the input gave up (may be null)
  const uint32_t* totals;       // (N, E1, E2, V) as the preprocess left them on the device (may be null)
  uint32_t N, E1, E2, V;        // the host's copy
  __device__ __forceinline__ uint32_t refusal() const {
    uint32_t r = 0u;
    if (sort_error && *sort_error != 0u) r |= GSR_STATUS_SORT;
    if (totals && (totals[0] != N || totals[1] != E1 || totals[2] != E2 || totals[3] != V)) r |= GSR_STATUS_TOTALS;
    return r;
  }
};

struct LevelArgs {
  Grid g;
  Pack pk;
  Guard guard;
  uint32_t V;                          // level 1: number of visible Gaussians
  const u64* in;                       // level 1: packed (index, rectangle) words front to back;
                                       // level 2: the same words per supercell list;
                                       // level 3: index | (x0 | x1 << 4) << 32, x relative to the cell
  const uint32_t* piece_first;         // levels 2, 3: first piece of every parent, [parents + 1]
  const uint4* desc;                   // levels 2, 3: (parent, start, length, index in parent) per piece
  uint32_t* counts;                    // counts, after the scan exclusive positions
  u64* out64;                          // levels 1, 2
  uint32_t* out32;                     // level 3: point_list
};

template <int LEVEL> struct Level;
// entries per piece; threads per block (level 1 has many buckets: more waves share a staged piece)
// and buckets a wave carries through one sweep of the piece
template <> struct Level<1> { static constexpr uint32_t kPiece = kPiece1; static constexpr int kThreads = 1024, kChunk = 4; };
template <> struct Level<2> { static constexpr uint32_t kPiece = kPiece2; static constexpr int kThreads = 256, kChunk = 4; };
template <> struct Level<3> { static constexpr uint32_t kPiece = kPiece3; static constexpr int kThreads = 512, kChunk = 2; };

__device__ __forceinline__ uint32_t lanes_below(unsigned long long ballot) {
  return __builtin_amdgcn_mbcnt_hi((uint32_t)(ballot >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)ballot, 0u));
}

// LDS word per staged entry: lo | (hi - lo) << 8 of the covered bucket interval, for level 1
// in x (bits 0-15) and y (bits 16-31); level 2 keeps the entry's x range in bits 16-23.
// Padding entries have lo = 255: no bucket index reaches it.
constexpr uint32_t kNoBox = 0x000000ffu;

template <int LEVEL, bool WRITE>
__global__ void __launch_bounds__(Level<LEVEL>::kThreads) partition_kernel(LevelArgs a) {
  constexpr uint32_t L = Level<LEVEL>::kPiece;
  constexpr int T = Level<LEVEL>::kThreads;
  constexpr int kIter = (int)L / T;
  __shared__ uint32_t s_box[L];
  __shared__ uint32_t s_idx[(WRITE && LEVEL != 1) ? L : 1];
  __shared__ u64 s_val[(WRITE && LEVEL == 1) ? L : 1];       // level 1 passes the packed word on unchanged

  uint32_t parent = 0, start, len, np, slot0, F, FX;
  if (LEVEL == 1) {
    np = (a.V + L - 1) / L;
    if (blockIdx.x >= np || a.guard.refusal()) return;
    start = blockIdx.x * L;
    len = min(L, a.V - start);
    slot0 = blockIdx.x;
    F = (uint32_t)a.g.supercells();
    FX = (uint32_t)a.g.sx;
  } else {
    const uint32_t parents = LEVEL == 2 ? (uint32_t)a.g.supercells() : (uint32_t)a.g.cells();
    if (blockIdx.x >= a.piece_first[parents]) return;
    const uint4 d = a.desc[blockIdx.x];
    const uint32_t pf0 = a.piece_first[d.x];
    np = a.piece_first[d.x + 1] - pf0;
    parent = d.x;
    start = d.y;
    len = d.z;
    slot0 = pf0 * kSuper + d.w;
    F = kSuper;
    FX = kSuper;
  }

  // stage the piece: all global loads of a thread are issued before any is used
  const uint32_t padded = (len + 63u) & ~63u;
  {
    u64 en[kIter];
#pragma unroll
    for (int it = 0; it < kIter; it++) {
      const uint32_t i = it * T + threadIdx.x;
      en[it] = i < len ? a.in[start + i] : 0ull;
    }
    const uint32_t cx = (parent % (uint32_t)a.g.sx) * kSuper, cy = (parent / (uint32_t)a.g.sx) * kSuper;  // level 2
#pragma unroll
    for (int it = 0; it < kIter; it++) {
      const uint32_t i = it * T + threadIdx.x;
      if (i < padded) {
        uint32_t box = kNoBox;
        if (i < len) {
          if (LEVEL == 1) {
            uint32_t x0, x1, y0, y1;
            a.pk.rect(en[it], x0, x1, y0, y1);
            const uint32_t bx0 = x0 / kSuper, bx1 = x1 / kSuper, by0 = y0 / kSuper, by1 = y1 / kSuper;
            box = bx0 | ((bx1 - bx0) << 8) | (by0 << 16) | ((by1 - by0) << 24);
            if (WRITE) s_val[i] = en[it];
          } else if (LEVEL == 2) {
            // rectangle clipped to this supercell, in tiles relative to its corner
            uint32_t x0, x1, y0, y1;
            a.pk.rect(en[it], x0, x1, y0, y1);
            const uint32_t lx0 = max(x0, cx) - cx, lx1 = min(x1, cx + kSuper - 1u) - cx;
            const uint32_t ly0 = max(y0, cy) - cy, ly1 = min(y1, cy + kSuper - 1u) - cy;
            box = ly0 | ((ly1 - ly0) << 8) | ((lx0 | (lx1 << 4)) << 16);
            if (WRITE) s_idx[i] = a.pk.index(en[it]);
          } else {
            const uint32_t lb = (uint32_t)(en[it] >> 32);
            const uint32_t x0 = lb & 15u, x1 = (lb >> 4) & 15u;
            box = x0 | ((x1 - x0) << 8);
            if (WRITE) s_idx[i] = (uint32_t)en[it];
          }
        }
        s_box[i] = box;
      }
    }
  }
  __syncthreads();

  // the wave index as a scalar: everything per bucket below then lives in scalar registers
  const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  constexpr uint32_t kWaves = T / 64;
  constexpr int CH = Level<LEVEL>::kChunk;
  const uint32_t per_wave = (F + kWaves - 1u) / kWaves;
  const uint32_t f_end = min(F, (wave + 1) * per_wave);
  for (uint32_t fb = wave * per_wave; fb < f_end; fb += CH) {
    // buckets past f_end get an index no interval reaches: they count nothing and write nothing
    uint32_t run[CH], fx[CH], fy[CH];
#pragma unroll
    for (int j = 0; j < CH; j++) {
      const uint32_t f = fb + j;
      const bool real = f < f_end;
      fx[j] = real ? (LEVEL == 1 ? f % FX : f) : 0xffffu;
      fy[j] = LEVEL == 1 ? f / FX : 0u;
      run[j] = (WRITE && real) ? a.counts[slot0 + f * np] : 0u;
    }
    for (uint32_t base = 0; base < len; base += 64) {
      const uint32_t w = s_box[base + lane];
      const uint32_t lo = w & 0xffu, wd = (w >> 8) & 0xffu, lo2 = (w >> 16) & 0xffu, wd2 = w >> 24;
      uint32_t idx = 0u;
      u64 val = 0ull;
      if (WRITE) {
        if (LEVEL == 1) val = s_val[base + lane]; else idx = s_idx[base + lane];
      }
#pragma unroll
      for (int j = 0; j < CH; j++) {
        bool hit = (fx[j] - lo) <= wd;
        if (LEVEL == 1) hit = hit && ((fy[j] - lo2) <= wd2);
        const unsigned long long bal = __ballot(hit);
        if (WRITE && hit) {
          const uint32_t pos = run[j] + lanes_below(bal);
          if (LEVEL == 1) {
            a.out64[pos] = val;
          } else if (LEVEL == 2) {
            a.out64[pos] = (u64)idx | ((u64)lo2 << 32);
          } else {
            a.out32[pos] = idx;
          }
        }
        run[j] += (uint32_t)__popcll(bal);
      }
    }
    if (!WRITE && lane == 0) {
#pragma unroll
      for (int j = 0; j < CH; j++)
        if (fb + j < f_end) a.counts[slot0 + (fb + j) * np] = run[j];
    }
  }
}

// Level 1 counts without a sweep over the buckets.  An entry covers a rectangle of supercells
// [bx0, bx1] x [by0, by1], so the number of entries covering supercell (fx, fy) is a 2-D prefix
// sum: add +1 / -1 at the four corners (bx0, by0), (bx1 + 1, by0), (bx0, by1 + 1),
// (bx1 + 1, by1 + 1) of a small LDS array with atomics, then sum along x and down y.  Work per
// entry is constant instead of proportional to the number of supercells.  (Levels 2 and 3 keep
// the sweep: with only 16 buckets the 64 lanes of a wave collide on the same few LDS words and
// the atomics take as long as the sweep.)
constexpr uint32_t kMaxDiffCells = 1024;   // (supercells in x + 1) * (supercells in y + 1) must fit

__global__ void __launch_bounds__(Level<1>::kThreads) count1_kernel(LevelArgs a) {
  constexpr uint32_t L = Level<1>::kPiece;
  constexpr int T = Level<1>::kThreads;
  constexpr int kIter = (int)L / T;
  __shared__ int s_d[kMaxDiffCells];
  const uint32_t np = (a.V + L - 1) / L;
  if (blockIdx.x >= np || a.guard.refusal()) return;
  const uint32_t start = blockIdx.x * L, len = min(L, a.V - start);
  const uint32_t FX = (uint32_t)a.g.sx, FY = (uint32_t)a.g.sy, stride = FX + 1u, rows = FY + 1u;
  for (uint32_t i = threadIdx.x; i < stride * rows; i += T) s_d[i] = 0;
  u64 en[kIter];
#pragma unroll
  for (int it = 0; it < kIter; it++) {
    const uint32_t i = it * T + threadIdx.x;
    en[it] = i < len ? a.in[start + i] : 0ull;
  }
  __syncthreads();
#pragma unroll
  for (int it = 0; it < kIter; it++) {
    if (it * T + threadIdx.x < len) {
      uint32_t x0, x1, y0, y1;
      a.pk.rect(en[it], x0, x1, y0, y1);
      const uint32_t bx0 = x0 / kSuper, bx1 = x1 / kSuper + 1u, by0 = y0 / kSuper, by1 = y1 / kSuper + 1u;
      atomicAdd(&s_d[by0 * stride + bx0], 1);
      atomicAdd(&s_d[by0 * stride + bx1], -1);
      atomicAdd(&s_d[by1 * stride + bx0], -1);
      atomicAdd(&s_d[by1 * stride + bx1], 1);
    }
  }
  __syncthreads();
  if (threadIdx.x < rows) {
    int run = 0;
    for (uint32_t x = 0; x < stride; x++) { run += s_d[threadIdx.x * stride + x]; s_d[threadIdx.x * stride + x] = run; }
  }
  __syncthreads();
  if (threadIdx.x < stride) {
    int run = 0;
    for (uint32_t y = 0; y < rows; y++) { run += s_d[y * stride + threadIdx.x]; s_d[y * stride + threadIdx.x] = run; }
  }
  __syncthreads();
  for (uint32_t f = threadIdx.x; f < FX * FY; f += T)
    a.counts[blockIdx.x + f * np] = (uint32_t)s_d[(f / FX) * stride + f % FX];
}

// Level 1 writing without a sweep over the buckets, for grids of at most 64 supercells.  Every entry
// turns its rectangle of supercells into a 64-bit coverage mask (bit f = covers supercell f).  The
// masks of 64 consecutive entries form a 64 x 64 bit matrix held one row per lane; its transpose,
// six exchange steps between lanes, leaves in lane f the word "which of the 64 entries cover
// supercell f" -- the ballot the sweep computes once per bucket, for all buckets at once.  The
// popcounts of those words, prefix-summed over the piece's steps of 64 entries on top of the piece's
// positions from the scan, give every (step, supercell) its first output slot; an entry's rank
// within its step is the number of lower lanes in the supercell's word.  Each entry then walks
// over the bits of its own mask (one or two supercells for most; the wave loops as long as its
// widest entry).
__device__ __forceinline__ u64 shfl_xor64(u64 v, int lanes) {
  const uint32_t lo = (uint32_t)__shfl_xor((int)(uint32_t)v, lanes), hi = (uint32_t)__shfl_xor((int)(uint32_t)(v >> 32), lanes);
  return ((u64)hi << 32) | lo;
}

__device__ __forceinline__ u64 transpose_bits64(u64 x, int lane) {
  constexpr u64 kLow[6] = {0x00000000ffffffffull, 0x0000ffff0000ffffull, 0x00ff00ff00ff00ffull,
                           0x0f0f0f0f0f0f0f0full, 0x3333333333333333ull, 0x5555555555555555ull};
#pragma unroll
  for (int k = 0; k < 6; k++) {
    const int s = 32 >> k;
    const u64 y = shfl_xor64(x, s);
    // blocks of s x s bits: the lane without bit s keeps its low blocks and takes the partner's low
    // blocks as its high ones, the lane with bit s the other way round
    x = (lane & s) ? (((y & ~kLow[k]) >> s) | (x & ~kLow[k])) : ((x & kLow[k]) | ((y & kLow[k]) << s));
  }
  return x;
}

// Levels 2 and 3 count without a sweep over the buckets either.  Their entries cover a run of the parent's 16 buckets,
// a 16-bit mask; a lane packs the masks of four entries (64 entries apart) into one 64-bit word, and the transpose
// of the wave's 64 words leaves in lane 16 c + f the word "which entries of group c cover bucket f": the sixteen
// ballots of the sweep for four groups of 64 entries at once, from a dozen lane exchanges.  Their popcounts are the
// counts.  (The writing sweeps keep the ballots per bucket: they need every entry's rank bucket by bucket, and
// their stores are what they wait for.)
template <int LEVEL>
__global__ void __launch_bounds__(Level<LEVEL>::kThreads) count_kernel(LevelArgs a) {
  static_assert(LEVEL == 2 || LEVEL == 3, "level 1 counts with a difference array");
  constexpr uint32_t L = Level<LEVEL>::kPiece;
  constexpr int T = Level<LEVEL>::kThreads, kWaves = T / 64;
  static_assert(kSuper == 16 && L % (256 * kWaves) == 0, "four groups of 64 entries per wave and round");
  __shared__ uint32_t s_cnt[kSuper];
  const uint32_t parents = LEVEL == 2 ? (uint32_t)a.g.supercells() : (uint32_t)a.g.cells();
  if (blockIdx.x >= a.piece_first[parents]) return;
  const uint4 d = a.desc[blockIdx.x];
  const uint32_t pf0 = a.piece_first[d.x], np = a.piece_first[d.x + 1] - pf0;
  const uint32_t parent = d.x, start = d.y, len = d.z, slot0 = pf0 * kSuper + d.w;
  const uint32_t cy = (parent / (uint32_t)a.g.sx) * kSuper;                  // level 2: the supercell's first tile row
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (threadIdx.x < kSuper) s_cnt[threadIdx.x] = 0u;
  __syncthreads();
  uint32_t mine = 0u;                  // lane 16 c + f: entries of the wave's groups c that cover bucket f
  for (uint32_t base = (uint32_t)wave * 256u; base < len; base += (uint32_t)kWaves * 256u) {
    u64 en[4];
#pragma unroll
    for (int c = 0; c < 4; c++) {
      const uint32_t i = base + 64u * c + (uint32_t)lane;
      en[c] = i < len ? a.in[start + i] : ~0ull;
    }
    u64 word = 0ull;
#pragma unroll
    for (int c = 0; c < 4; c++) {
      const uint32_t i = base + 64u * c + (uint32_t)lane;
      uint32_t lo, hi;
      if (LEVEL == 2) {
        uint32_t x0, x1, y0, y1;
        a.pk.rect(en[c], x0, x1, y0, y1);
        lo = max(y0, cy) - cy;
        hi = min(y1, cy + kSuper - 1u) - cy;
      } else {
        const uint32_t lb = (uint32_t)(en[c] >> 32);
        lo = lb & 15u;
        hi = (lb >> 4) & 15u;
      }
      const uint32_t mask = i < len ? ((2u << (hi - lo)) - 1u) << lo : 0u;
      word |= (u64)(mask & 0xffffu) << (16 * c);
    }
    mine += (uint32_t)__popcll(transpose_bits64(word, lane));
  }
  mine += (uint32_t)__shfl_xor((int)mine, 16);
  mine += (uint32_t)__shfl_xor((int)mine, 32);
  if (lane < kSuper && mine) atomicAdd(&s_cnt[lane], mine);
  __syncthreads();
  if (threadIdx.x < kSuper) a.counts[slot0 + threadIdx.x * np] = s_cnt[threadIdx.x];
}

constexpr int kSteps1 = (int)(kPiece1 / 64);                       // steps of 64 entries per piece
static_assert(kPiece1 == 2 * Level<1>::kThreads && kSteps1 <= 64, "write1_kernel: two entries per thread, steps scanned by one wave");

__global__ void __launch_bounds__(Level<1>::kThreads) write1_kernel(LevelArgs a) {
  constexpr uint32_t L = Level<1>::kPiece;
  constexpr int kWaves = Level<1>::kThreads / 64;
  __shared__ uint32_t s_base[kSteps1][65];     // [step][supercell]: counts, then first output slots; 65: the scan reads columns
  const uint32_t np = (a.V + L - 1) / L;
  if (blockIdx.x >= np || a.guard.refusal()) return;
  const uint32_t start = blockIdx.x * L, len = min(L, a.V - start);
  const uint32_t F = (uint32_t)a.g.supercells(), FX = (uint32_t)a.g.sx;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;

  // the piece's positions of this wave's supercells (used after the first barrier)
  constexpr int kPerWave = 64 / kWaves;
  uint32_t piece_pos = 0;
  if (lane < kPerWave && (uint32_t)(wave * kPerWave + lane) < F)
    piece_pos = a.counts[blockIdx.x + (uint32_t)(wave * kPerWave + lane) * np];

  u64 en[2], mask[2], cover[2];
#pragma unroll
  for (int k = 0; k < 2; k++) {
    const uint32_t i = (uint32_t)((wave * 2 + k) * 64 + lane);
    en[k] = i < len ? a.in[start + i] : 0ull;
  }
#pragma unroll
  for (int k = 0; k < 2; k++) {
    const int step = wave * 2 + k;
    const uint32_t i = (uint32_t)(step * 64 + lane);
    u64 m = 0ull;
    if (i < len) {
      uint32_t x0, x1, y0, y1;
      a.pk.rect(en[k], x0, x1, y0, y1);
      const uint32_t bx0 = x0 / kSuper, bx1 = x1 / kSuper, by0 = y0 / kSuper, by1 = y1 / kSuper;
      const u64 row = ((2ull << (bx1 - bx0)) - 1ull) << bx0;       // a width of 64 wraps to all ones
      for (uint32_t y = by0; y <= by1; y++) m |= row << (y * FX);
    }
    mask[k] = m;
    cover[k] = transpose_bits64(m, lane);                         // lane = supercell
    s_base[step][lane] = (uint32_t)__popcll(cover[k]);
  }
  __syncthreads();

  // exclusive prefix sum over the steps, per supercell: lane = step, kPerWave supercells per wave
#pragma unroll
  for (int q = 0; q < kPerWave; q++) {
    const int f = wave * kPerWave + q;
    const uint32_t c = lane < kSteps1 ? s_base[lane][f] : 0u;
    const uint32_t inc = gsr::wave_inclusive_sum(c);       // lanes from kSteps1 on hold zeros
    const uint32_t first = (uint32_t)__builtin_amdgcn_readlane((int)piece_pos, q);
    if (lane < kSteps1) s_base[lane][f] = first + inc - c;
  }
  __syncthreads();

#pragma unroll
  for (int k = 0; k < 2; k++) {
    const int step = wave * 2 + k;
    u64 m = mask[k];
    const uint32_t cover_lo = (uint32_t)cover[k], cover_hi = (uint32_t)(cover[k] >> 32);
    while (__ballot(m != 0ull) != 0ull) {
      // every lane takes part in the exchange; the ones with nothing left fetch supercell 0's word and write nothing
      const int f = m ? __ffsll((long long)m) - 1 : 0;
      const uint32_t lo = (uint32_t)__shfl((int)cover_lo, f), hi = (uint32_t)__shfl((int)cover_hi, f);
      const uint32_t rank = __builtin_amdgcn_mbcnt_hi(hi, __builtin_amdgcn_mbcnt_lo(lo, 0u));
      if (m) a.out64[s_base[step][f] + rank] = en[k];
      m &= m - 1ull;
    }
  }
}

// ---------------------------------------------------------------------------
// prefix sums between the sweeps: one block each
// ---------------------------------------------------------------------------
// In-place exclusive prefix sum of a[0..n) by one block of kScanBlock threads; every thread
// gets the total.  a must be 16-byte aligned.
__device__ uint32_t block_scan_inplace(uint32_t* a, uint32_t n, uint32_t* s_wave /* kScanBlock / 64 words */) {
  const int t = threadIdx.x, lane = t & 63, w = t >> 6;
  constexpr int kVec = 4;                       // 16-byte loads per thread and round, all in flight together
  uint32_t carry = 0;
  for (uint32_t base = 0; base < n; base += 4 * kVec * kScanBlock) {
    const uint32_t i0 = base + 4 * kVec * t;
    uint4 v[kVec];
    uint32_t s = 0;
#pragma unroll
    for (int k = 0; k < kVec; k++) {
      const uint32_t i = i0 + 4 * k;
      v[k] = make_uint4(0u, 0u, 0u, 0u);
      if (i + 3 < n) {
        v[k] = *reinterpret_cast<const uint4*>(a + i);
      } else {
        if (i < n) v[k].x = a[i];
        if (i + 1 < n) v[k].y = a[i + 1];
        if (i + 2 < n) v[k].z = a[i + 2];
      }
      s += v[k].x + v[k].y + v[k].z + v[k].w;
    }
    const uint32_t inc = gsr::wave_inclusive_sum(s);
    if (lane == 63) s_wave[w] = inc;
    __syncthreads();
    uint32_t before = 0, tot = 0;
#pragma unroll
    for (int k = 0; k < kScanBlock / 64; k++) {
      const uint32_t x = s_wave[k];
      before += k < w ? x : 0u;
      tot += x;
    }
    uint32_t ex = carry + before + inc - s;
#pragma unroll
    for (int k = 0; k < kVec; k++) {
      const uint32_t i = i0 + 4 * k;
      const uint4 o4 = make_uint4(ex, ex + v[k].x, ex + v[k].x + v[k].y, ex + v[k].x + v[k].y + v[k].z);
      ex = o4.w + v[k].w;
      if (i + 3 < n) {
        *reinterpret_cast<uint4*>(a + i) = o4;
      } else {
        if (i < n) a[i] = o4.x;
        if (i + 1 < n) a[i + 1] = o4.y;
        if (i + 2 < n) a[i + 2] = o4.z;
      }
    }
    carry += tot;
    __syncthreads();
  }
  return carry;
}

// Largest b with first[b] <= q (first is non-decreasing, first[0] = 0, first[n] > q).
__device__ __forceinline__ uint32_t owner_of(const uint32_t* first, uint32_t n, uint32_t q) {
  uint32_t lo = 0, hi = n;
  while (hi - lo > 1) {
    const uint32_t mid = (lo + hi) >> 1;
    if (first[mid] <= q) lo = mid; else hi = mid;
  }
  return lo;
}

// Cuts the lists (list_start, list_len)[0..n) into pieces of at most piece_len entries:
// piece_first (n + 1 words, holds the piece counts on entry) and the descriptors.
// Every step that reads from global memory what an earlier step of the same block wrote is a round trip of
// a microsecond in a kernel that has nothing to hide it behind, so when the lists fit (n + 1 <= kFirstLds:
// any image up to 4096 cells) they never leave LDS: LdsLists holds start, length and piece count of every
// list, the scan of the piece counts runs there, and the descriptors are built from there; only piece_first
// and the descriptors, which the next kernels read, are written out.
constexpr uint32_t kFirstLds = 4096;

struct LdsLists {
  uint32_t* start;   // [kFirstLds]
  uint32_t* len;     // [kFirstLds]
  uint32_t* first;   // [kFirstLds]: piece counts in, first piece of every list out (n + 1 words)
};

// Exclusive scan of a[0..n) in LDS, n <= 4 * kScanBlock; returns the total.  Contains barriers.
__device__ uint32_t block_scan_lds(uint32_t* a, uint32_t n, uint32_t* s_wave) {
  static_assert(kFirstLds <= 4 * kScanBlock, "four words per thread");
  const uint32_t lane = threadIdx.x & 63u, w = threadIdx.x >> 6, i0 = 4u * threadIdx.x;
  uint32_t v[4], s = 0;
#pragma unroll
  for (int k = 0; k < 4; k++) { v[k] = i0 + k < n ? a[i0 + k] : 0u; s += v[k]; }
  const uint32_t inc = gsr::wave_inclusive_sum(s);
  __syncthreads();
  if (lane == 63) s_wave[w] = inc;
  __syncthreads();
  uint32_t before = 0, tot = 0;
#pragma unroll
  for (int k = 0; k < kScanBlock / 64; k++) {
    const uint32_t x = s_wave[k];
    before += (uint32_t)k < w ? x : 0u;
    tot += x;
  }
  uint32_t ex = before + inc - s;
#pragma unroll
  for (int k = 0; k < 4; k++) {
    if (i0 + k < n) a[i0 + k] = ex;
    ex += v[k];
  }
  __syncthreads();
  return tot;
}

__device__ void build_pieces_lds(uint32_t n, const LdsLists& L, uint32_t piece_len, uint32_t* piece_first, uint4* desc,
                                 uint32_t* s_wave) {
  __syncthreads();
  const uint32_t pieces = block_scan_lds(L.first, n + 1, s_wave);
  for (uint32_t b = threadIdx.x; b <= n; b += kScanBlock) piece_first[b] = L.first[b];
  for (uint32_t q = threadIdx.x; q < pieces; q += kScanBlock) {
    const uint32_t b = owner_of(L.first, n, q);
    const uint32_t j = q - L.first[b];
    desc[q] = make_uint4(b, L.start[b] + j * piece_len, min(piece_len, L.len[b] - j * piece_len), j);
  }
}

__device__ void build_pieces(uint32_t n, const uint32_t* list_start, const uint32_t* list_len, uint32_t piece_len,
                             uint32_t* piece_first, uint4* desc, uint32_t* s_wave) {
  __syncthreads();
  const uint32_t pieces = block_scan_inplace(piece_first, n + 1, s_wave);
  __syncthreads();
  for (uint32_t q = threadIdx.x; q < pieces; q += kScanBlock) {
    const uint32_t b = owner_of(piece_first, n, q);
    const uint32_t j = q - piece_first[b];
    const uint32_t len = list_len[b];
    desc[q] = make_uint4(b, list_start[b] + j * piece_len, min(piece_len, len - j * piece_len), j);
  }
}

// After the level-1 counting sweep: positions, the supercell lists, the level-2 pieces.
__global__ void __launch_bounds__(kScanBlock) scan1_kernel(Grid g, Guard guard, uint32_t* status, uint32_t V,
                                                           uint32_t* counts, uint32_t* list_start,
                                                           uint32_t* list_len, uint32_t* piece_first, uint4* desc) {
  __shared__ uint32_t s_wave[kScanBlock / 64];
  __shared__ uint32_t s_start[kFirstLds], s_len[kFirstLds], s_first[kFirstLds];
  const uint32_t F = (uint32_t)g.supercells(), np = (V + kPiece1 - 1) / kPiece1, slots = np * F;
  const uint32_t refused = guard.refusal();
  if (refused) {
    // no level-2 pieces: every later kernel of the chain finds nothing to do, all tile ranges become (0,0)
    for (uint32_t b = threadIdx.x; b <= F; b += kScanBlock) piece_first[b] = 0u;
    if (threadIdx.x == 0 && status) atomicOr(status, refused);
    return;
  }
  const uint32_t total = block_scan_inplace(counts, slots, s_wave);
  __syncthreads();
  const bool in_lds = F + 1 <= kFirstLds;
  for (uint32_t b = threadIdx.x; b < F; b += kScanBlock) {
    const uint32_t begin = np ? counts[b * np] : 0u;
    const uint32_t end = (np && b + 1 < F) ? counts[(b + 1) * np] : total;
    const uint32_t pieces = (end - begin + kPiece2 - 1) / kPiece2;
    if (in_lds) {
      s_start[b] = begin; s_len[b] = end - begin; s_first[b] = pieces;
    } else {
      list_start[b] = begin; list_len[b] = end - begin; piece_first[b] = pieces;
    }
  }
  if (threadIdx.x == 0) { if (in_lds) s_first[F] = 0u; else piece_first[F] = 0u; }
  if (in_lds) build_pieces_lds(F, LdsLists{s_start, s_len, s_first}, kPiece2, piece_first, desc, s_wave);
  else build_pieces(F, list_start, list_len, kPiece2, piece_first, desc, s_wave);
}

// After the level-2 counting sweep: positions, the cell lists (row-major cell ids), the level-3 pieces.
__global__ void __launch_bounds__(kScanBlock) scan2_kernel(Grid g, const uint32_t* piece_first2, uint32_t* counts,
                                                           uint32_t* list_start, uint32_t* list_len,
                                                           uint32_t* piece_first, uint4* desc) {
  __shared__ uint32_t s_wave[kScanBlock / 64];
  __shared__ uint32_t s_start[kFirstLds], s_len[kFirstLds], s_first[kFirstLds];
  const uint32_t parents = (uint32_t)g.supercells(), cells = (uint32_t)g.cells();
  const uint32_t slots = piece_first2[parents] * kSuper;
  const uint32_t total = block_scan_inplace(counts, slots, s_wave);
  __syncthreads();
  const bool in_lds = cells + 1 <= kFirstLds;
  for (uint32_t q = threadIdx.x; q < parents * kSuper; q += kScanBlock) {
    const uint32_t p = q / kSuper, f = q % kSuper;
    const uint32_t y = (p / g.sx) * kSuper + f;
    if (y >= (uint32_t)g.gy) continue;
    const uint32_t pf0 = piece_first2[p], np = piece_first2[p + 1] - pf0;
    uint32_t begin = 0u, end = 0u;
    if (np) {
      const uint32_t s0 = pf0 * kSuper + f * np;
      begin = counts[s0];
      end = s0 + np < slots ? counts[s0 + np] : total;
    }
    const uint32_t cell = y * g.sx + p % g.sx;
    const uint32_t pieces = (end - begin + kPiece3 - 1) / kPiece3;
    if (in_lds) {
      s_start[cell] = begin; s_len[cell] = end - begin; s_first[cell] = pieces;
    } else {
      list_start[cell] = begin; list_len[cell] = end - begin; piece_first[cell] = pieces;
    }
  }
  if (threadIdx.x == 0) { if (in_lds) s_first[cells] = 0u; else piece_first[cells] = 0u; }
  if (in_lds) build_pieces_lds(cells, LdsLists{s_start, s_len, s_first}, kPiece3, piece_first, desc, s_wave);
  else build_pieces(cells, list_start, list_len, kPiece3, piece_first, desc, s_wave);
}

// After the level-3 counting sweep: positions in point_list and the tile ranges,
// [start, end) per tile, (0,0) for empty tiles.  The tiles of a cell come from two dependent loads each
// (the cell's pieces, then the scanned counts at the ends of the tile's run of pieces); a thread's tiles are
// handled kTilesAtOnce at a time with the loads of all of them in flight together, not one round trip
// after the other.
constexpr int kTilesAtOnce = 4;

__global__ void __launch_bounds__(kScanBlock) scan3_kernel(Grid g, const uint32_t* __restrict__ piece_first3,
                                                           uint32_t* __restrict__ counts, uint32_t* __restrict__ ranges) {
  __shared__ uint32_t s_wave[kScanBlock / 64];
  const uint32_t parents = (uint32_t)g.cells();
  const uint32_t slots = piece_first3[parents] * kSuper;
  const uint32_t total = block_scan_inplace(counts, slots, s_wave);
  __syncthreads();
  const uint32_t work = parents * kSuper;
  for (uint32_t q0 = threadIdx.x; q0 < work; q0 += kTilesAtOnce * kScanBlock) {
    uint32_t pf0[kTilesAtOnce], pf1[kTilesAtOnce];
#pragma unroll
    for (int k = 0; k < kTilesAtOnce; k++) {
      const uint32_t q = q0 + k * kScanBlock, c = min(q, work - 1u) / kSuper;
      pf0[k] = piece_first3[c];
      pf1[k] = piece_first3[c + 1];
    }
    uint32_t begin[kTilesAtOnce], end[kTilesAtOnce];
#pragma unroll
    for (int k = 0; k < kTilesAtOnce; k++) {
      const uint32_t q = q0 + k * kScanBlock, f = q % kSuper, np = pf1[k] - pf0[k];
      const uint32_t s0 = pf0[k] * kSuper + f * np;
      const bool any = q < work && np != 0u;
      begin[k] = any ? counts[s0] : 0u;
      end[k] = !any ? 0u : (s0 + np < slots ? counts[s0 + np] : total);
    }
#pragma unroll
    for (int k = 0; k < kTilesAtOnce; k++) {
      const uint32_t q = q0 + k * kScanBlock;
      if (q >= work) continue;
      const uint32_t c = q / kSuper, f = q % kSuper;
      const uint32_t x = (c % g.sx) * kSuper + f, y = c / g.sx;
      if (x >= (uint32_t)g.gx) continue;
      const uint32_t t = y * g.gx + x;
      reinterpret_cast<uint2*>(ranges)[t] = end[k] > begin[k] ? make_uint2(begin[k], end[k]) : make_uint2(0u, 0u);
    }
  }
}

__global__ void __launch_bounds__(kBlock) zero_u32_kernel(uint32_t* p, size_t n) {
  for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (size_t)gridDim.x * kBlock) p[i] = 0u;
}

// Optional: tile id of every sorted instance (from which the 64-bit sort keys are rebuilt).
__global__ void __launch_bounds__(kBlock) fill_tile_ids_kernel(const uint32_t* __restrict__ ranges,
                                                               uint32_t* __restrict__ tile_ids) {
  const uint32_t t = blockIdx.x, b = ranges[2 * t], e = ranges[2 * t + 1];
  for (uint32_t i = b + threadIdx.x; i < e; i += kBlock) tile_ids[i] = t;
}

// Scratch layout of gsr_bin_lists, in 256-byte aligned pieces.
struct ListsLayout {
  size_t counts1, list_start2, list_len2, piece_first2, desc2, entries1, counts2;
  size_t list_start3, list_len3, piece_first3, desc3, entries2, counts3, total;
  uint32_t pieces1, max_pieces2, max_pieces3;
};

ListsLayout lists_layout(uint32_t V, uint32_t E1, uint32_t E2, const Grid& g) {
  ListsLayout L;
  size_t o = 0;
  auto take = [&](size_t bytes) { const size_t at = o; o += gsr::align_up(bytes ? bytes : 1, 256); return at; };
  L.pieces1 = div_up_u(V, kPiece1);
  L.max_pieces2 = E1 / kPiece2 + (uint32_t)g.supercells() + 1;
  L.max_pieces3 = E2 / kPiece3 + (uint32_t)g.cells() + 1;
  L.counts1 = take((size_t)L.pieces1 * g.supercells() * 4);
  L.list_start2 = take((size_t)g.supercells() * 4);
  L.list_len2 = take((size_t)g.supercells() * 4);
  L.piece_first2 = take(((size_t)g.supercells() + 1) * 4);
  L.desc2 = take((size_t)L.max_pieces2 * 16);
  L.entries1 = take((size_t)E1 * 8);
  L.counts2 = take((size_t)L.max_pieces2 * kSuper * 4);
  L.list_start3 = take((size_t)g.cells() * 4);
  L.list_len3 = take((size_t)g.cells() * 4);
  L.piece_first3 = take(((size_t)g.cells() + 1) * 4);
  L.desc3 = take((size_t)L.max_pieces3 * 16);
  L.entries2 = take((size_t)E2 * 8);
  L.counts3 = take((size_t)L.max_pieces3 * kSuper * 4);
  L.total = o + 256;
  return L;
}

}  // namespace

int gsr::launch_totals_final(int blocks, const uint32_t* partials, const uint32_t* status, uint32_t* totals,
                             hipStream_t stream) {
  hipLaunchKernelGGL(totals_final_kernel, dim3(1), dim3(kScanBlock), 0, stream, blocks, partials, status, totals);
  return gsr::check_launch("totals");
}

extern "C" size_t gsr_bin_sort_bytes(int P) { return gsr::depth_sort_bytes(P); }

extern "C" int gsr_bin_sort(int P, const uint32_t* depth_keys, const uint64_t* sort_vals, uint64_t* sorted_vals,
                            void* scratch, size_t scratch_bytes, void* stream) {
  GSR_REQUIRE(P >= 0 && P < (1 << 30), "bin_sort: bad P");
  if (P == 0) return GSR_OK;
  GSR_REQUIRE(depth_keys && sort_vals && sorted_vals && scratch, "bin_sort: null pointer");
  if (scratch_bytes < gsr_bin_sort_bytes(P) || ((uintptr_t)scratch & 255))
    return gsr::fail(GSR_E_SCRATCH, "bin_sort: scratch too small (%zu < %zu) or not 256-byte aligned", scratch_bytes,
                     gsr_bin_sort_bytes(P));
  return gsr::depth_sort(P, depth_keys, sort_vals, sorted_vals, scratch, (hipStream_t)stream);
}

extern "C" int gsr_bin_sort_status(int P, const void* scratch, size_t scratch_bytes, void* stream) {
  GSR_REQUIRE(P >= 0 && P < (1 << 30), "bin_sort_status: bad P");
  if (P == 0) return GSR_OK;
  GSR_REQUIRE(scratch != nullptr, "bin_sort_status: null pointer");
  if (scratch_bytes < gsr_bin_sort_bytes(P) || ((uintptr_t)scratch & 255))
    return gsr::fail(GSR_E_SCRATCH, "bin_sort_status: not the scratch of a sort of %d keys", P);
  return gsr::depth_sort_status(scratch, (hipStream_t)stream);
}

extern "C" void gsr_debug_sort_spin_limit(uint32_t limit) { gsr::set_sort_spin_limit(limit); }
extern "C" void gsr_assume_exclusive_device(int on) { gsr::set_exclusive_device(on); }

extern "C" size_t gsr_bin_lists_bytes(uint32_t V, uint32_t E1, uint32_t E2, int W, int H) {
  if (V == 0 || W <= 0 || H <= 0) return 256;
  return lists_layout(V, E1, E2, make_grid(W, H)).total;
}

extern "C" const uint32_t* gsr_bin_sort_error_word(const void* sort_scratch) {
  return sort_scratch ? gsr::depth_sort_error_word(sort_scratch) : nullptr;
}

extern "C" int gsr_bin_lists(int P, uint32_t V, uint32_t N, uint32_t E1, uint32_t E2, int W, int H,
                             const uint64_t* sorted_vals, uint32_t* point_list, uint32_t* ranges,
                             uint32_t* tile_ids_sorted, const uint32_t* sort_error, const uint32_t* totals,
                             uint32_t* status, void* scratch, size_t scratch_bytes, void* stream) {
  const hipStream_t st = (hipStream_t)stream;
  GSR_REQUIRE(P >= 0 && W > 0 && H > 0, "bin_lists: bad sizes");
  GSR_REQUIRE(ranges, "bin_lists: null ranges");
  const Grid g = make_grid(W, H);
  GSR_REQUIRE(g.sx <= 255 && g.sy <= 255, "bin_lists: image %dx%d too large", W, H);
  if (N == 0 || V == 0 || P == 0) {
    hipLaunchKernelGGL(zero_u32_kernel, dim3((unsigned)gsr::div_up(2 * g.tiles(), kBlock)), dim3(kBlock), 0, st, ranges,
                       (size_t)2 * g.tiles());
    return gsr::check_launch("bin_lists");
  }
  GSR_REQUIRE(sorted_vals && point_list && scratch, "bin_lists: null pointer");
  GSR_REQUIRE(V <= (uint32_t)P && E1 >= V && E2 >= E1 && N >= E2, "bin_lists: inconsistent totals V=%u E1=%u E2=%u N=%u",
              V, E1, E2, N);
  const ListsLayout L = lists_layout(V, E1, E2, g);
  if (scratch_bytes < L.total || ((uintptr_t)scratch & 255))
    return gsr::fail(GSR_E_SCRATCH, "bin_lists: scratch too small (%zu < %zu) or not 256-byte aligned", scratch_bytes,
                     L.total);
  char* base = (char*)scratch;
  LevelArgs a{};
  a.g = g;
  a.pk = gsr::make_pack(P, g.gx, g.gy);
  GSR_REQUIRE(a.pk.bits() <= 64, "bin_lists: sizes do not fit the 64-bit sort value");
  a.V = V;
  a.guard = Guard{sort_error, totals, N, E1, E2, V};

  // level 1: depth-sorted Gaussians -> supercell lists
  a.in = (const u64*)sorted_vals;
  a.counts = (uint32_t*)(base + L.counts1);
  a.out64 = (u64*)(base + L.entries1);
  if ((uint32_t)(g.sx + 1) * (uint32_t)(g.sy + 1) <= kMaxDiffCells)
    hipLaunchKernelGGL(count1_kernel, dim3(L.pieces1), dim3(Level<1>::kThreads), 0, st, a);
  else
    hipLaunchKernelGGL((partition_kernel<1, false>), dim3(L.pieces1), dim3(Level<1>::kThreads), 0, st, a);
  hipLaunchKernelGGL(scan1_kernel, dim3(1), dim3(kScanBlock), 0, st, g, a.guard, status, V, a.counts,
                     (uint32_t*)(base + L.list_start2), (uint32_t*)(base + L.list_len2), (uint32_t*)(base + L.piece_first2),
                     (uint4*)(base + L.desc2));
  if (g.supercells() <= 64)
    hipLaunchKernelGGL(write1_kernel, dim3(L.pieces1), dim3(Level<1>::kThreads), 0, st, a);
  else
    hipLaunchKernelGGL((partition_kernel<1, true>), dim3(L.pieces1), dim3(Level<1>::kThreads), 0, st, a);
  int rc = gsr::check_launch("bin_lists: level 1");
  if (rc) return rc;

  // level 2: supercell lists -> cell lists
  a.in = (const u64*)(base + L.entries1);
  a.piece_first = (const uint32_t*)(base + L.piece_first2);
  a.desc = (const uint4*)(base + L.desc2);
  a.counts = (uint32_t*)(base + L.counts2);
  a.out64 = (u64*)(base + L.entries2);
  hipLaunchKernelGGL(count_kernel<2>, dim3(L.max_pieces2), dim3(Level<2>::kThreads), 0, st, a);
  hipLaunchKernelGGL(scan2_kernel, dim3(1), dim3(kScanBlock), 0, st, g, a.piece_first, a.counts,
                     (uint32_t*)(base + L.list_start3), (uint32_t*)(base + L.list_len3),
                     (uint32_t*)(base + L.piece_first3), (uint4*)(base + L.desc3));
  hipLaunchKernelGGL((partition_kernel<2, true>), dim3(L.max_pieces2), dim3(Level<2>::kThreads), 0, st, a);
  rc = gsr::check_launch("bin_lists: level 2");
  if (rc) return rc;

  // level 3: cell lists -> tile lists = point_list
  a.in = (const u64*)(base + L.entries2);
  a.piece_first = (const uint32_t*)(base + L.piece_first3);
  a.desc = (const uint4*)(base + L.desc3);
  a.counts = (uint32_t*)(base + L.counts3);
  a.out64 = nullptr;
  a.out32 = point_list;
  hipLaunchKernelGGL(count_kernel<3>, dim3(L.max_pieces3), dim3(Level<3>::kThreads), 0, st, a);
  hipLaunchKernelGGL(scan3_kernel, dim3(1), dim3(kScanBlock), 0, st, g, a.piece_first, a.counts,
                     ranges);
  hipLaunchKernelGGL((partition_kernel<3, true>), dim3(L.max_pieces3), dim3(Level<3>::kThreads), 0, st, a);
  if (tile_ids_sorted)
    hipLaunchKernelGGL(fill_tile_ids_kernel, dim3(g.tiles()), dim3(kBlock), 0, st, ranges, tile_ids_sorted);
  return gsr::check_launch("bin_lists: level 3");
}
