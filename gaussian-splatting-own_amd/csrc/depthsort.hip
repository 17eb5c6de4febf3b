// Depth sort of the binning for gfx950: a stable least-significant-digit radix sort of
// (32-bit depth key, 64-bit value) pairs, 8 bits per pass, each pass one kernel ("Onesweep":
// the digit histograms of all four passes are taken up front, and within a pass a workgroup
// learns how many equal digits sit in earlier workgroups through a chained look-back instead
// of a separate count + scan launch).
//
// What the general-purpose library sort cannot know and this one uses:
//  * culled Gaussians (key 0xFFFFFFFF) are dropped by the first pass, so the later passes move
//    only the visible ones, and only the first `visible` entries of the output are defined
//    (that is all the list builder reads);
//  * after the last pass nobody needs the keys: it writes values only;
//  * all per-pass state (look-back words, workgroup tickets, histograms) is cleared by ONE
//    launch for the whole sort, and the bin scan runs inside the pass kernels:
//    6 launches per sort instead of 15;
//  * depth keys are IEEE floats: when every visible Gaussian lies within one factor of 4 of depth
//    (z in [2, 8) for a camera four units from a unit-sized scene) their top bytes are all equal
//    and the fourth pass would be the identity.  The histograms say so up front: the third pass then
//    writes the final result and the fourth returns at once;
//  * the look-back is two-level and windowed: a workgroup sums the counts of the (at most 15)
//    earlier workgroups of its group of 16 with all loads in flight together, the last workgroup
//    of a group publishes the group's count and later its inclusive prefix, and everybody looks
//    back over groups the same way.  When all workgroups of a pass are resident at once (a 1M-key
//    sort: 245 of them) nobody finishes early, so a one-level look-back degenerates into every
//    workgroup walking over all its predecessors, one memory latency per step; here it is about
//    three latencies whatever the position.
//
// Forward progress: a workgroup takes its number from an atomic ticket when it starts running,
// so every workgroup it waits for is already running or done; the waits are bounded by a spin
// limit anyway (on overrun the sort sets an error word and still terminates).  A caller that has
// the GPU to itself (gsr_assume_exclusive_device, or GSR_EXCLUSIVE_DEVICE=1 in the environment)
// may skip the ticket when the whole grid is resident at once: workgroup b then takes chunk b.
// That rests on every workgroup of the grid being able to start, which other kernels holding CU
// slots (a second stream, another process, a CU mask) can break; hence opt-in only.
#include <atomic>
#include <cstdlib>
#include <mutex>
#include <type_traits>

#include "gsr_common.h"

namespace {

typedef unsigned long long u64;

constexpr int kThreads = 1024;
constexpr int kWaves = kThreads / 64;
// keys per thread and pass: 4 keeps a 1M-key sort at one short workgroup per CU; 8 amortises the
// per-workgroup latencies (ticket, look-back) once there are several rounds of workgroups anyway; 2 shortens
// every workgroup's chain when the whole sort is a fraction of one round (100k keys: 48.8 -> 44.6 us; at 1M
// keys 2, 3 and 6 all lose against 4: 78, 74, 70 against 67 us)
constexpr int kTinyItems = 2, kFewItems = 4, kManyItems = 8;
constexpr uint32_t kFewFrom = 200000u, kManyFrom = 4500000u;

inline int items_per_thread(uint32_t P) { return P >= kManyFrom ? kManyItems : P >= kFewFrom ? kFewItems : kTinyItems; }
constexpr uint32_t kCulled = 0xFFFFFFFFu;
constexpr uint32_t kPartial = 1u << 30, kComplete = 2u << 30, kFlags = 3u << 30;
constexpr int kGroup = 16;                 // workgroups per look-back group
constexpr int kWindow = 16;                // predecessors inspected per look-back step
constexpr uint32_t kSpinLimit = 1u << 22;  // look-back steps before giving up (never reached in a healthy run)
// what the next sorts use; only tests of the give-up path set anything else
std::atomic<uint32_t> g_spin_limit{kSpinLimit};
// -1 = not decided yet (the environment is asked at first use), 0 = tickets, 1 = the caller has the GPU to itself
std::atomic<int> g_exclusive{-1};

bool exclusive_device() {
  int v = g_exclusive.load(std::memory_order_relaxed);
  if (v < 0) {
    const char* e = getenv("GSR_EXCLUSIVE_DEVICE");
    v = (e && e[0] && !(e[0] == '0' && e[1] == 0)) ? 1 : 0;
    int expected = -1;
    g_exclusive.compare_exchange_strong(expected, v, std::memory_order_relaxed);
    v = g_exclusive.load(std::memory_order_relaxed);
  }
  return v == 1;
}

// control words behind the 4 x 256 histogram bins
enum { kCtlTicket = 0, kCtlError = 4, kCtlValid = 8, kCtlWords = 16 };   // 4 tickets, the error word, 8 partial key counts
// The digit histograms of the four passes, kHistCopies times over: workgroup b of the histogram kernel adds
// its counts to copy b % kHistCopies, the readers add the copies up.  Atomics of many workgroups on ONE word
// retire one after the other (tens of nanoseconds each, device scope on a part with eight L2s); with 245
// workgroups flushing into one copy that was a third of the kernel's time.
constexpr int kHistCopies = 8;
static_assert(kCtlValid + kHistCopies <= kCtlWords, "one partial count of valid keys per histogram copy");
constexpr int kHistCopyWords = 4 * 256;
constexpr int kHistWords = kHistCopies * kHistCopyWords;

__device__ __forceinline__ uint32_t hist_bin(const uint32_t* __restrict__ hist, int pass, int digit) {
  uint32_t n = 0;
#pragma unroll
  for (int c = 0; c < kHistCopies; c++) n += hist[c * kHistCopyWords + 256 * pass + digit];
  return n;
}

__device__ __forceinline__ uint32_t valid_keys(const uint32_t* __restrict__ ctl) {
  uint32_t n = 0;
#pragma unroll
  for (int c = 0; c < kHistCopies; c++) n += ctl[kCtlValid + c];
  return n;
}

__device__ __forceinline__ uint32_t wave_incl_scan(uint32_t v) { return gsr::wave_inclusive_sum(v); }

// Clears the histograms, control words and look-back states.  With a mirror, the first workgroup first
// finishes the list sizes of the preprocess and sends them to the host (see SortMirror): the read-back then
// costs neither a kernel of its own, nor a copy command between the preprocess and the sort, nor an event.
template <int THREADS>
__device__ __forceinline__ void send_list_sizes(const gsr::SortMirror& m) {
  __shared__ uint32_t s_part[THREADS / 64][5];
  const uint32_t mine = gsr::sum_totals<THREADS>(m.blocks, m.partials, m.status, m.totals, s_part);
  if (threadIdx.x < 7) m.dst[threadIdx.x] = mine;          // word 5: too many instances; word 6 is spare: 0
  __threadfence_system();
  __syncthreads();
  if (threadIdx.x == 0) {
    m.dst[7] = m.ticket;                                    // behind the sizes
    __threadfence_system();
  }
}

__global__ void __launch_bounds__(256) sort_clear_kernel(uint32_t* __restrict__ words, size_t n, gsr::SortMirror m) {
  if (m.dst != nullptr && blockIdx.x == 0) send_list_sizes<256>(m);
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) words[i] = 0u;
}

// Digit histograms of all four passes over the keys of visible Gaussians, and their number.
__global__ void __launch_bounds__(kThreads) sort_histogram_kernel(const uint32_t* __restrict__ keys, uint32_t P,
                                                                  uint32_t* __restrict__ hist,
                                                                  uint32_t* __restrict__ ctl, uint32_t own_blocks,
                                                                  gsr::SortMirror m) {
  // the first kernel of a sort whose scratch the caller has cleared: the list sizes leave from here, from a
  // workgroup of their own behind the kernel's own_blocks
  if (blockIdx.x >= own_blocks) {
    send_list_sizes<kThreads>(m);
    return;
  }
  __shared__ uint32_t s_h[kHistCopyWords];
  __shared__ uint32_t s_valid;
  s_h[threadIdx.x] = 0u;
  if (threadIdx.x == 0) s_valid = 0u;
  __syncthreads();
  uint32_t valid = 0;
  constexpr int kPer = 4;   // keys per thread and round, their loads in flight together
  for (size_t i0 = (size_t)blockIdx.x * (kThreads * kPer); i0 < P; i0 += (size_t)own_blocks * (kThreads * kPer)) {
    uint32_t key[kPer];
#pragma unroll
    for (int j = 0; j < kPer; j++) {
      const size_t i = i0 + (size_t)j * kThreads + threadIdx.x;
      key[j] = i < P ? keys[i] : kCulled;
    }
#pragma unroll
    for (int j = 0; j < kPer; j++) {
      const uint32_t k = key[j];
      const bool ok = k != kCulled;
      valid += ok ? 1u : 0u;
      // Depth keys are floats: their upper bytes take few distinct values, and 64 lanes adding to
      // one LDS word serialise.  The first few distinct digits of a wave are counted with a ballot
      // and added once; what is left (well spread digits) goes through per-lane atomics.
#pragma unroll
      for (int place = 0; place < 4; place++) {
        const uint32_t d = (k >> (8 * place)) & 255u;
        unsigned long long rest = __ballot(ok);
#pragma unroll
        for (int it = 0; it < 3; it++) {
          if (rest == 0ull) break;
          const int first = __ffsll((long long)rest) - 1;
          const uint32_t dv = gsr::lane_value(d, first);
          const unsigned long long same = __ballot(ok && d == dv) & rest;
          if ((int)(threadIdx.x & 63) == first) atomicAdd(&s_h[256 * place + dv], (uint32_t)__popcll(same));
          rest &= ~same;
        }
        if ((rest >> (threadIdx.x & 63)) & 1ull) atomicAdd(&s_h[256 * place + d], 1u);
      }
    }
  }
  valid = gsr::wave_sum(valid);
  if ((threadIdx.x & 63) == 0 && valid) atomicAdd(&s_valid, valid);
  __syncthreads();
  const uint32_t c = s_h[threadIdx.x];
  if (c) atomicAdd(&hist[(blockIdx.x % kHistCopies) * kHistCopyWords + threadIdx.x], c);
  if (threadIdx.x == 0 && s_valid) atomicAdd(&ctl[kCtlValid + (blockIdx.x % kHistCopies)], s_valid);   // once per workgroup
}

struct PassArgs {
  const uint32_t* keys_in;
  const u64* vals_in;
  uint32_t* keys_out;     // unused by the last pass
  u64* vals_out;
  u64* final_out;         // passes 2 and 3: the sort's result (sorted_vals)
  const uint32_t* hist;   // all copies of the four digit histograms
  uint32_t* ctl;
  uint32_t* states;       // [workgroup][256] look-back words of this pass
  uint32_t* group_states; // [group][256]
  uint32_t P;
  uint32_t spin_limit;    // look-back steps before a workgroup gives up and flags the sort as failed
  uint32_t static_order;  // workgroup b takes chunk b (the whole grid is resident at once) instead of drawing a ticket
};

// Look-back over the counts published at states[q][t] for q = p, p-1, ..., stop, newest first: a window of
// kWindow words is loaded at once (device-coherent loads: a round trip past the L2s, a microsecond each
// round), then eaten from the newest on, up to the first word that is still empty (wait for it: load again
// from there) or flagged complete (it already holds everything before it: done).
struct LookBack {
  const uint32_t* states;
  int p, stop, t;
  uint32_t sum = 0, spins = 0;
  bool done;
  uint32_t st[kWindow];
  __device__ __forceinline__ LookBack(const uint32_t* s, int p_, int stop_, int t_) : states(s), p(p_), stop(stop_), t(t_), done(p_ < stop_) {}
  __device__ __forceinline__ void load() {
#pragma unroll
    for (int j = 0; j < kWindow; j++)
      st[j] = (!done && p - j >= stop) ? __hip_atomic_load(states + (size_t)(p - j) * 256 + t, __ATOMIC_RELAXED,
                                                            __HIP_MEMORY_SCOPE_AGENT)
                                       : kComplete;
  }
  __device__ __forceinline__ void eat() {
    if (done) return;
    int used = 0;
#pragma unroll
    for (int j = 0; j < kWindow; j++) {
      if (!done && used == j && (st[j] & kFlags) != 0u) {
        sum += st[j] & ~kFlags;
        used = j + 1;
        done = (st[j] & kFlags) == kComplete;
      }
    }
    p -= used;
    done = done || p < stop;
  }
  // the rest of the way, one window after the other
  __device__ __forceinline__ uint32_t finish(uint32_t spin_limit, uint32_t* error_word, uint32_t error_bit) {
    while (!done) {
      if (++spins > spin_limit) { atomicOr(error_word, error_bit); break; }
      load();
      eat();
    }
    return sum;
  }
};

__device__ __forceinline__ uint32_t look_back(const uint32_t* states, int p, int stop, int t, uint32_t spin_limit,
                                              uint32_t* error_word, uint32_t error_bit) {
  LookBack lb(states, p, stop, t);
  lb.load();
  lb.eat();
  return lb.finish(spin_limit, error_word, error_bit);
}

// One radix pass.  PASS 0 reads all P keys and drops the culled ones; later passes read the
// ctl[kCtlValid] survivors.  Items are owned wave-striped (item i of lane l of wave w is element
// w*256 + i*64 + l of the workgroup's chunk), ranked within the wave by matching digits with
// ballots, and brought into workgroup order through LDS so that the global writes are runs.
#ifdef GSR_SORT_TIMING
__device__ long long g_sort_times[4][512][8];
#define STAMP(k) do { if (tid == 0 && wg < 512) g_sort_times[PASS][wg][k] = wall_clock64(); } while (0)
#else
#define STAMP(k) do {} while (0)
#endif

template <int PASS, int kItemsPerThread>
__global__ void __launch_bounds__(kThreads) sort_pass_kernel(PassArgs a) {
  constexpr int kItems = kThreads * kItemsPerThread;   // keys per workgroup and pass
  __shared__ uint32_t s_count[kWaves][256];   // per-wave digit counts, later the wave's offset within its digit
  __shared__ uint32_t s_total[256];           // the workgroup's digit counts
  __shared__ uint32_t s_local[256];           // first local rank of each digit
  __shared__ uint32_t s_dest[256];            // global position of local rank 0 of each digit, minus that rank
  __shared__ uint32_t s_keys[kItems];
  __shared__ u64 s_vals[kItems];
  __shared__ uint32_t s_misc[16];
  constexpr int kShift = 8 * PASS;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;

  // A pass that follows one in which a look-back gave up would rank keys that were never written
  // against the up-front histograms, and its positions would no longer be bounded by the key count:
  // the failure is sticky, every later pass of the sort does nothing (one scalar load, uniform).  Only the
  // bits of EARLIER passes are looked at: those are final once this kernel runs, so every wave of every
  // workgroup reads the same answer; a failure raised during this pass by another workgroup goes through
  // the bounded look-back like any other, and nobody leaves without publishing its counts.
  constexpr uint32_t kEarlier = ((1u << PASS) - 1u) | (((1u << PASS) - 1u) << 4);
  if (PASS > 0 && (a.ctl[kCtlError] & kEarlier) != 0u) return;

  // One bin of the top byte's histogram holds every key: the last pass would move nothing.  Pass 2 then
  // writes the result itself and pass 3 has nothing to do.  The test reads this pass's view of the
  // top-byte histogram against the number of valid keys;
  // every workgroup of both kernels comes to the same answer.
  bool top_byte_is_constant = false;
  if (PASS >= 2) {
    const uint32_t all = valid_keys(a.ctl);
    top_byte_is_constant = __syncthreads_or(threadIdx.x < 256 && all != 0u && hist_bin(a.hist, 3, threadIdx.x & 255) == all) != 0;
    if (PASS == 3 && top_byte_is_constant) return;
  }
  const bool writes_result = PASS == 3 || (PASS == 2 && top_byte_is_constant);
  u64* const vals_dst = writes_result ? a.final_out : a.vals_out;

  // Which chunk: a look-back waits for the workgroups of all earlier chunks, so those must be running or done.
  // A ticket drawn at start-up guarantees that whatever order the hardware starts workgroups in; when the
  // whole grid fits on the GPU at once (the host checks) the workgroup's own index does, and saves 245
  // atomics on one word that retire one after the other (1.5 us of every pass at 1M keys).
  if (tid == 0) s_misc[0] = a.static_order ? blockIdx.x : atomicAdd(&a.ctl[kCtlTicket + PASS], 1u);
  if (tid < 256) s_total[tid] = 0u;
#pragma unroll
  for (int k = 0; k < kWaves * 256 / kThreads; k++) (&s_count[0][0])[k * kThreads + tid] = 0u;
  __syncthreads();
  const uint32_t wg = s_misc[0];
  STAMP(0);
  const uint32_t n = PASS == 0 ? a.P : valid_keys(a.ctl);
  const uint32_t base = wg * (uint32_t)kItems;
  if (base >= n) return;                       // the whole workgroup: nothing left for it

  const uint32_t bin = tid < 256 ? hist_bin(a.hist, PASS, tid) : 0u;

  uint32_t key[kItemsPerThread], digit[kItemsPerThread], rank[kItemsPerThread];
  u64 val[kItemsPerThread];
  bool valid[kItemsPerThread];
#pragma unroll
  for (int i = 0; i < kItemsPerThread; i++) {
    const uint32_t idx = base + (uint32_t)(wave * (64 * kItemsPerThread) + i * 64 + lane);
    valid[i] = idx < n;
    key[i] = valid[i] ? a.keys_in[idx] : kCulled;
    val[i] = valid[i] ? a.vals_in[idx] : 0ull;      // both loads in flight together
  }
#pragma unroll
  for (int i = 0; i < kItemsPerThread; i++) {
    if (PASS == 0) valid[i] = key[i] != kCulled;
    digit[i] = (key[i] >> kShift) & 255u;
  }

  // The workgroup's digit counts first, so that the look-back words go out before the ranking and
  // travel while it runs.  The most common digit of each wave is counted with a ballot (the upper
  // bytes of depth keys take few values, and 64 lanes adding to one LDS word serialise).
#pragma unroll
  for (int i = 0; i < kItemsPerThread; i++) {
    const unsigned long long all = __ballot(valid[i]);
    if (all != 0ull) {
      const int first = __ffsll((long long)all) - 1;
      const uint32_t dv = gsr::lane_value(digit[i], first);
      const unsigned long long same = __ballot(valid[i] && digit[i] == dv);
      if (lane == first) atomicAdd(&s_total[dv], (uint32_t)__popcll(same));
      if (valid[i] && digit[i] != dv) atomicAdd(&s_total[digit[i]], 1u);
    }
  }
  __syncthreads();
  const uint32_t total = tid < 256 ? s_total[tid] : 0u;
  const int group = (int)wg / kGroup;
  const bool closes_group = (int)wg % kGroup == kGroup - 1;
  uint32_t in_group = 0;
  uint32_t* const group_word = a.group_states + (size_t)group * 256 + (tid & 255);
  if (tid < 256) {
    __hip_atomic_store(a.states + (size_t)wg * 256 + tid, kPartial | total, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    // The workgroup that closes a group collects the group's count right away, ahead of its own
    // ranking: every later group waits for that word.
    if (closes_group) {
      in_group = look_back(a.states, (int)wg - 1, group * kGroup, tid, a.spin_limit, &a.ctl[kCtlError], 1u << PASS);
      __hip_atomic_store(group_word, kPartial | (in_group + total), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }

  STAMP(1);
  // rank within the wave: lanes holding the same digit find each other with one ballot per digit bit
  const u64 lanes_below = (1ull << lane) - 1ull;
#pragma unroll
  for (int i = 0; i < kItemsPerThread; i++) {
    u64 same = __ballot(valid[i]);
#pragma unroll
    for (int bit = 0; bit < 8; bit++) {
      const bool mine = (digit[i] >> bit) & 1u;
      const u64 set = __ballot(mine);
      same &= mine ? set : ~set;
    }
    const uint32_t before = (uint32_t)__popcll(same & lanes_below);
    uint32_t seen = 0;
    if (valid[i]) {
      seen = s_count[wave][digit[i]];
      if (before == 0) s_count[wave][digit[i]] = seen + (uint32_t)__popcll(same);
    }
    rank[i] = seen + before;
  }
  __syncthreads();
  STAMP(2);

  // per digit: offsets of the waves within the digit
  if (tid < 256) {
    uint32_t before = 0;
#pragma unroll
    for (int w = 0; w < kWaves; w++) {
      const uint32_t c = s_count[w][tid];
      s_count[w][tid] = before;
      before += c;
    }
  }
  // exclusive scans over the 256 digits: of the workgroup's counts (local ranks) and of the
  // global bins (where each digit starts in the output)
  const uint32_t incl_local = wave_incl_scan(total), incl_bin = wave_incl_scan(bin);
  if (lane == 63 && wave < 4) { s_misc[4 + wave] = incl_local; s_misc[8 + wave] = incl_bin; }
  __syncthreads();
  STAMP(3);
  if (tid < 256) {
    uint32_t local0 = incl_local - total, bin0 = incl_bin - bin;
    for (int w = 0; w < wave; w++) { local0 += s_misc[4 + w]; bin0 += s_misc[8 + w]; }
    if (tid == 255) s_misc[1] = local0 + total;                     // valid items in this workgroup

    // look-back: how many items with this digit sit in earlier workgroups of this group and in earlier groups;
    // the first windows of both are in flight together
    LookBack mine(a.states, closes_group ? -1 : (int)wg - 1, closes_group ? 0 : group * kGroup, tid);
    LookBack theirs(a.group_states, group - 1, 0, tid);
    mine.load();
    theirs.load();
    mine.eat();
    theirs.eat();
    if (!closes_group) in_group = mine.finish(a.spin_limit, &a.ctl[kCtlError], 1u << PASS);
    const uint32_t groups_before = theirs.finish(a.spin_limit, &a.ctl[kCtlError], 16u << PASS);
    if (closes_group)
      __hip_atomic_store(group_word, kComplete | (groups_before + in_group + total), __ATOMIC_RELAXED,
                         __HIP_MEMORY_SCOPE_AGENT);
    const uint32_t earlier = groups_before + in_group;
    s_local[tid] = local0;
    s_dest[tid] = bin0 + earlier - local0;
  }
  __syncthreads();
  STAMP(4);

  // into workgroup order
#pragma unroll
  for (int i = 0; i < kItemsPerThread; i++) {
    if (valid[i]) {
      const uint32_t r = s_local[digit[i]] + s_count[wave][digit[i]] + rank[i];
      s_keys[r] = key[i];
      s_vals[r] = val[i];
    }
  }
  __syncthreads();
  STAMP(5);
  const uint32_t mine_n = s_misc[1];
#pragma unroll
  for (int k = 0; k < kItemsPerThread; k++) {
    const uint32_t r = (uint32_t)(k * kThreads + tid);
    if (r < mine_n) {
      const uint32_t kk = s_keys[r];
      const uint32_t dst = s_dest[(kk >> kShift) & 255u] + r;
      if (dst < a.P) {     // always true in a healthy sort; holds the stores inside the buffers whatever a failed look-back returned
        if (!writes_result) a.keys_out[dst] = kk;
        vals_dst[dst] = s_vals[r];
      }
    }
  }
  STAMP(6);
}

struct Layout {
  size_t state_words, group_words;   // per pass
  size_t clear_words;   // histogram + control + look-back words: what one launch clears
  size_t keys_a, keys_b, vals_a, vals_b, total;   // byte offsets
  uint32_t workgroups;
};

Layout layout(uint32_t P) {
  Layout L;
  const uint32_t items = (uint32_t)kThreads * (uint32_t)items_per_thread(P);
  L.workgroups = (P + items - 1) / items;
  L.state_words = (size_t)L.workgroups * 256;
  L.group_words = (size_t)((L.workgroups + kGroup - 1) / kGroup) * 256;
  L.clear_words = kHistWords + kCtlWords + 4 * (L.state_words + L.group_words);
  size_t o = gsr::align_up(L.clear_words * 4, 256);
  L.keys_a = o; o += gsr::align_up((size_t)P * 4, 256);
  L.keys_b = o; o += gsr::align_up((size_t)P * 4, 256);
  L.vals_a = o; o += gsr::align_up((size_t)P * 8, 256);
  L.vals_b = o; o += gsr::align_up((size_t)P * 8, 256);
  L.total = o;
  return L;
}

}  // namespace

namespace {

// How many workgroups of the pass kernels (the most demanding of the four) the current device holds at once;
// 0 if that cannot be found out.  Asked once per device and keys-per-thread variant.
template <int I>
uint32_t resident_of() {
  int dev = 0, cus = 0, per_cu = 1 << 30;
  if (hipGetDevice(&dev) != hipSuccess) return 0;
  if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) return 0;
  const void* kernels[4] = {(const void*)sort_pass_kernel<0, I>, (const void*)sort_pass_kernel<1, I>,
                            (const void*)sort_pass_kernel<2, I>, (const void*)sort_pass_kernel<3, I>};
  for (const void* k : kernels) {
    int n = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, k, kThreads, 0) != hipSuccess) { (void)hipGetLastError(); return 0; }
    per_cu = n < per_cu ? n : per_cu;
  }
  return (uint32_t)(cus > 0 && per_cu > 0 ? cus * per_cu : 0);
}

uint32_t resident_workgroups(int items) {
  static std::mutex m;
  static uint32_t cache[64][3];                 // [device][variant], 0 = not asked yet
  static bool asked[64][3];
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 0;
  const int v = items == kManyItems ? 2 : items == kFewItems ? 1 : 0;
  std::lock_guard<std::mutex> lock(m);
  if (!asked[dev][v]) {
    cache[dev][v] = v == 2 ? resident_of<kManyItems>() : v == 1 ? resident_of<kFewItems>() : resident_of<kTinyItems>();
    asked[dev][v] = true;
  }
  return cache[dev][v];
}

}  // namespace

namespace gsr {

size_t depth_sort_bytes(int P) { return P > 0 ? layout((uint32_t)P).total : 256; }

// sorted_vals[0 .. visible) = the values of the keys != 0xFFFFFFFF in ascending key order, equal
// keys in input order; the rest of sorted_vals is left untouched.
int depth_sort(int P, const uint32_t* keys, const uint64_t* vals, uint64_t* sorted_vals, void* scratch, hipStream_t st,
               const SortMirror* mirror, bool cleared) {
  const Layout L = layout((uint32_t)P);
  const SortMirror no_mirror{nullptr, 0, nullptr, nullptr, nullptr, 0u};
  char* base = (char*)scratch;
  uint32_t* hist = (uint32_t*)base;
  uint32_t* ctl = hist + kHistWords;
  uint32_t* states = ctl + kCtlWords;
  uint32_t* ka = (uint32_t*)(base + L.keys_a);
  uint32_t* kb = (uint32_t*)(base + L.keys_b);
  u64* va = (u64*)(base + L.vals_a);
  u64* vb = (u64*)(base + L.vals_b);
  const unsigned clear_blocks = (unsigned)((L.clear_words + 1023) / 1024 < 1024 ? (L.clear_words + 1023) / 1024 : 1024);
  if (!cleared)
    hipLaunchKernelGGL(sort_clear_kernel, dim3(clear_blocks), dim3(256), 0, st, hist, L.clear_words, mirror ? *mirror : no_mirror);
  const unsigned hist_wanted = ((unsigned)P + 4095u) / 4096u;
  // 512: measured against 32 .. 2048 at 1M and 6M keys (fewer blocks serialise the LDS counting, more only add flush atomics)
  const unsigned hist_blocks = hist_wanted < 512u ? hist_wanted : 512u;
  const bool sizes_here = cleared && mirror;
  hipLaunchKernelGGL(sort_histogram_kernel, dim3(hist_blocks + (sizes_here ? 1u : 0u)), dim3(kThreads), 0, st, keys, (uint32_t)P,
                     hist, ctl, hist_blocks, sizes_here ? *mirror : no_mirror);
  const dim3 block(kThreads);
  const uint32_t spin_limit = g_spin_limit.load(std::memory_order_relaxed);
  const uint32_t static_order =
      exclusive_device() && L.workgroups <= resident_workgroups(items_per_thread((uint32_t)P)) ? 1u : 0u;
  const size_t per_pass = L.state_words + L.group_words;
  auto args = [&](int pass, const uint32_t* kin, const u64* vin, uint32_t* kout, u64* vout) {
    uint32_t* s = states + pass * per_pass;
    return PassArgs{kin, vin, kout, vout, (u64*)sorted_vals, hist, ctl, s, s + L.state_words, (uint32_t)P, spin_limit, static_order};
  };
  const dim3 grid(L.workgroups);
  auto passes = [&](auto items) {
    constexpr int I = decltype(items)::value;
    hipLaunchKernelGGL((sort_pass_kernel<0, I>), grid, block, 0, st, args(0, keys, (const u64*)vals, ka, va));
    hipLaunchKernelGGL((sort_pass_kernel<1, I>), grid, block, 0, st, args(1, ka, va, kb, vb));
    hipLaunchKernelGGL((sort_pass_kernel<2, I>), grid, block, 0, st, args(2, kb, vb, ka, va));
    hipLaunchKernelGGL((sort_pass_kernel<3, I>), grid, block, 0, st, args(3, ka, va, nullptr, nullptr));
  };
  switch (items_per_thread((uint32_t)P)) {
    case kManyItems: passes(std::integral_constant<int, kManyItems>{}); break;
    case kFewItems: passes(std::integral_constant<int, kFewItems>{}); break;
    default: passes(std::integral_constant<int, kTinyItems>{}); break;
  }
  return gsr::check_launch("depth_sort");
}

void depth_sort_clear_region(int P, void* scratch, uint32_t** at, size_t* words) {
  *at = (uint32_t*)scratch;
  *words = layout((uint32_t)P).clear_words;
}

// A look-back that ran into its spin limit leaves a bit per pass in the sort's control words.  In that
// pass the counts of earlier keys come out too small, so its positions collide (still below the key
// count); the passes after it see the word and do nothing, and every store is bounded by P: nothing
// outside the sort's buffers is written, but sorted_vals is incomplete.  The word is read here, after
// waiting for the stream; gsr_bin_lists reads it on the device (gsr_bin_sort_error_word).
int depth_sort_status(const void* scratch, hipStream_t st) {
  const uint32_t* ctl = (const uint32_t*)scratch + kHistWords;
  uint32_t word = 0;
  hipError_t e = hipMemcpyAsync(&word, ctl + kCtlError, sizeof(word), hipMemcpyDeviceToHost, st);
  if (e == hipSuccess) e = hipStreamSynchronize(st);
  if (e != hipSuccess) return gsr::fail(GSR_E_LAUNCH, "bin_sort_status: %s", hipGetErrorString(e));
  if (word != 0u)
    return gsr::fail(GSR_E_SORT, "bin_sort: a look-back gave up waiting (passes 0x%x in groups, 0x%x across groups)",
                     word & 15u, (word >> 4) & 15u);
  return GSR_OK;
}

const uint32_t* depth_sort_error_word(const void* scratch) { return (const uint32_t*)scratch + kHistWords + kCtlError; }

void set_sort_spin_limit(uint32_t limit) { g_spin_limit.store(limit ? limit : kSpinLimit, std::memory_order_relaxed); }

void set_exclusive_device(int on) { g_exclusive.store(on ? 1 : 0, std::memory_order_relaxed); }

}  // namespace gsr

// ---------------------------------------------------------------------------
// The depth sort for callers that cannot allocate: its scratch is a device array of the library
// ---------------------------------------------------------------------------
namespace {

constexpr int kStaticMaxKeys = 1 << 20;
// what layout(kStaticMaxKeys) asks for: look-back words and histograms (1.1 MB) + two key and two value buffers
constexpr size_t kStaticBytes = 26312960;

// Every GPU the library runs on holds its own copy of this array, at its own address, until the process ends.
static __device__ __align__(256) unsigned char scratch[kStaticBytes];

constexpr int kMaxDevices = 64;
std::mutex g_static_mutex;
void* g_static_ptr[kMaxDevices];          // the array's address on each device, looked up once per device

// The device address of `scratch` on the current device.  Host code never uses the array's name for its
// memory: the name is a host-side stand-in, the address comes from the runtime.
int static_scratch(void** out) {
  int dev = -1;
  hipError_t e = hipGetDevice(&dev);
  if (e != hipSuccess) return gsr::fail(GSR_E_LAUNCH, "bin_sort_static: hipGetDevice: %s", hipGetErrorString(e));
  if (dev < 0 || dev >= kMaxDevices) return gsr::fail(GSR_E_INVALID, "bin_sort_static: device %d out of range", dev);
  std::lock_guard<std::mutex> lock(g_static_mutex);
  if (!g_static_ptr[dev]) {
    void* ptr = nullptr;
    e = hipGetSymbolAddress((void**)&ptr, HIP_SYMBOL(scratch));
    if (e != hipSuccess || !ptr)
      return gsr::fail(GSR_E_LAUNCH, "bin_sort_static: hipGetSymbolAddress: %s", hipGetErrorString(e));
    if ((uintptr_t)ptr & 255) return gsr::fail(GSR_E_SCRATCH, "bin_sort_static: the static scratch is not 256-byte aligned");
    g_static_ptr[dev] = ptr;
  }
  *out = g_static_ptr[dev];
  return GSR_OK;
}

}  // namespace

// What the sort's layout asks for at the capacity, and what the array holds (a test ties the two together).
extern "C" size_t gsr_bin_sort_static_bytes(int which) {
  return which == 0 ? kStaticBytes : layout((uint32_t)kStaticMaxKeys).total;
}

extern "C" int gsr_bin_sort_static(int P, const uint32_t* depth_keys, const uint64_t* sort_vals, uint64_t* sorted_vals,
                                   const uint32_t** error_word, void* stream) {
  GSR_REQUIRE(P >= 0, "bin_sort_static: bad P");
  if (P > kStaticMaxKeys)
    return gsr::fail(GSR_E_TOO_LARGE, "bin_sort_static: %d keys exceed the static scratch's capacity of %d; use gsr_bin_sort "
                     "with caller memory", P, kStaticMaxKeys);
  // the layout of THIS key count (keys per thread change with it, so the size is not monotonic by construction)
  if (layout((uint32_t)P).total > kStaticBytes)
    return gsr::fail(GSR_E_SCRATCH, "bin_sort_static: the static scratch (%zu bytes) is smaller than the sort's layout for %d keys (%zu)",
                     kStaticBytes, P, layout((uint32_t)P).total);
  void* ptr = nullptr;
  int rc = static_scratch(&ptr);
  if (rc) return rc;
  if (error_word) *error_word = gsr::depth_sort_error_word(ptr);
  // everything a later call looks up once per device is looked up now, so that a P = 0 call warms the device up
  if (exclusive_device()) {
    (void)resident_workgroups(kTinyItems);
    (void)resident_workgroups(kFewItems);
    (void)resident_workgroups(kManyItems);
  }
  if (P == 0) return GSR_OK;
  GSR_REQUIRE(depth_keys && sort_vals && sorted_vals, "bin_sort_static: null pointer");
  // depth_sort clears the look-back words, tickets, histograms and the error word with a kernel of its own
  // before it uses them: whatever the previous call left in the array does not matter
  return gsr::depth_sort(P, depth_keys, sort_vals, sorted_vals, ptr, (hipStream_t)stream);
}
