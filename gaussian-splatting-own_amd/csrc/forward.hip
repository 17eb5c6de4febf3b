// The whole forward in one call: preprocess -> list sizes to the host -> depth sort -> tile lists -> blend.
// The shape of the replaced module's own native entry (one call per render, memory for the lists obtained
// through a callback once their size is known); built from the stage entry points of gsr.h, so the results
// are theirs bit for bit.  The point of it is the host side: one transition into native code per render
// instead of five, and one workspace instead of a dozen buffers, which is what a render of 100k
// Gaussians spends as much time on as on its kernels.
#include <atomic>
#include <chrono>
#include <cstdlib>

#include "gsr_common.h"

namespace {

struct Carve {
  char* base;
  size_t at = 0;
  explicit Carve(void* p) : base((char*)p) {}
  template <typename T> T* take(size_t bytes) {
    T* p = base ? (T*)(base + at) : nullptr;
    at += gsr::align_up(bytes ? bytes : 1, 256);
    return p;
  }
};

struct Workspace {
  uint32_t* records; uint32_t* depth_keys; uint64_t* sort_vals; uint64_t* sorted_vals; uint32_t* totals;
  void* segments; void* sort_scratch; size_t sort_bytes; uint32_t* ranges; float* final_T; uint32_t* n_contrib; size_t total;
};

Workspace carve(void* base, int P, int W, int H) {
  Carve c(base);
  Workspace w;
  const size_t p = (size_t)(P > 0 ? P : 0), tiles = (size_t)gsr::div_up(W, GSR_TILE_X) * gsr::div_up(H, GSR_TILE_Y);
  w.records = c.take<uint32_t>(p * GSR_REC_WORDS * 4);
  w.depth_keys = c.take<uint32_t>(p * 4);
  w.sort_vals = c.take<uint64_t>(p * 8);
  w.sorted_vals = c.take<uint64_t>(p * 8);
  w.totals = c.take<uint32_t>(gsr_totals_words(P) * 4);
  // the segments block ends with the words the blend wants zeroed, the sort's scratch begins with the words the
  // sort wants zeroed: side by side they are one region for the preprocess to clear
  w.segments = c.take<char>(gsr::segments_bytes(W, H));
  w.sort_bytes = gsr_bin_sort_bytes(P);
  w.sort_scratch = c.take<char>(w.sort_bytes);
  w.ranges = c.take<uint32_t>(tiles * 8);
  w.final_T = c.take<float>((size_t)W * H * 4);
  w.n_contrib = c.take<uint32_t>((size_t)W * H * 4);
  w.total = c.at;
  return w;
}

int hip_fail(const char* what, hipError_t e) { return gsr::fail(GSR_E_LAUNCH, "forward: %s: %s", what, hipGetErrorString(e)); }

// Tickets of the list-size read-back: unique per call in this process, never 0.
uint32_t next_ticket() {
  static std::atomic<uint32_t> counter{1u};
  uint32_t t = counter.fetch_add(1u, std::memory_order_relaxed);
  return t ? t : counter.fetch_add(1u, std::memory_order_relaxed);
}

// Spins until the GPU has stored `ticket` at landing[7] (behind the sizes at landing[0..4]).  The stream is asked
// how it is only once the wait has lasted 20 ms, and then every 20 ms: the query puts a marker into the stream,
// which costs the GPU five microseconds between the kernels it lands between, and a healthy wait is over in less
// than a millisecond.  An error ends the wait, and so does a stream that has run dry without the ticket.
int await_ticket(const uint32_t* landing, uint32_t ticket, hipStream_t st) {
  const volatile uint32_t* word = landing + 7;
  auto asked = std::chrono::steady_clock::now();
  for (uint32_t looks = 1;; looks++) {
    if (*word == ticket) break;
    if ((looks & 0xFFFu) == 0u && std::chrono::steady_clock::now() - asked > std::chrono::milliseconds(20)) {
      const hipError_t q = hipStreamQuery(st);
      if (q == hipSuccess) {
        if (*word == ticket) break;
        return gsr::fail(GSR_E_LAUNCH, "forward: the list sizes never arrived in the landing buffer");
      }
      if (q != hipErrorNotReady) return hip_fail("waiting for the list sizes", q);
      asked = std::chrono::steady_clock::now();
    }
    __builtin_ia32_pause();
  }
  std::atomic_thread_fence(std::memory_order_acquire);
  return GSR_OK;
}

}  // namespace

extern "C" size_t gsr_forward_workspace_bytes(int P, int W, int H) {
  if (P < 0 || W <= 0 || H <= 0) return 0;
  return carve(nullptr, P, W, H).total;
}

extern "C" int gsr_forward(const GsrForwardArgs* a, gsr_resize_fn resize, void* resize_ctx, GsrForwardOut* out,
                           void* stream) {
  GSR_REQUIRE(a && out && resize, "forward: null argument");
  GSR_REQUIRE(a->P >= 0 && a->W > 0 && a->H > 0, "forward: bad sizes P=%d W=%d H=%d", a->P, a->W, a->H);
  GSR_REQUIRE(a->bg && a->out_color && a->radii, "forward: null pointer");
  GSR_REQUIRE(a->workspace && ((uintptr_t)a->workspace & 255) == 0, "forward: workspace must be non-null and 256-byte aligned");
  GSR_REQUIRE(a->P == 0 || (a->h_totals && a->event), "forward: the landing buffer and its event are required");
  const int P = a->P, W = a->W, H = a->H;
  const Workspace w = carve(a->workspace, P, W, H);
  if (a->workspace_bytes < w.total)
    return gsr::fail(GSR_E_SCRATCH, "forward: workspace too small (%zu < %zu)", a->workspace_bytes, w.total);
  hipStream_t st = (hipStream_t)stream;
  *out = GsrForwardOut{};
  out->records = w.records; out->depth_keys = w.depth_keys; out->sorted_vals = w.sorted_vals; out->ranges = w.ranges;
  out->final_T = w.final_T; out->n_contrib = w.n_contrib;
  auto checkpoint = [&](const char* what) -> int {      // debug mode: wait and look for errors after every stage
    if (!a->debug) return GSR_OK;
    const hipError_t e = hipStreamSynchronize(st);
    return e == hipSuccess ? GSR_OK : hip_fail(what, e);
  };

  uint32_t N = 0, E1 = 0, E2 = 0, V = 0;
  const uint32_t* sort_error = nullptr;
  // depth segments for the backward (the preprocess clears their counter: none without Gaussians)
  static const bool no_segments = getenv("GSR_NO_SEGMENTS") != nullptr;   // for measurements of the difference
  void* const segments = (P > 0 && !no_segments && gsr::segment_length_for(0u) != 0u) ? w.segments : nullptr;
  out->segments = segments;
  GSR_REQUIRE(a->zero_fill_bytes == 0 || (a->zero_fill && ((uintptr_t)a->zero_fill & 15) == 0 && a->zero_fill_bytes % 16 == 0),
              "forward: the zero-fill must be 16-byte aligned and a multiple of 16 bytes");
  if (P > 0) {
    // The list sizes (and the carried status word) travel to the host while the GPU sorts by depth.  When
    // the landing buffer is pinned memory the GPU can address, the sort's first kernel sums the preprocess's
    // partial records and stores the sizes there itself, followed by a ticket the host polls for; otherwise
    // the preprocess sums them in a kernel of its own and a copy command goes ahead of the sort.
    static const bool copy_always = getenv("GSR_COPY_READBACK") != nullptr;   // for measurements of the difference
    void* landing = nullptr;
    hipError_t e = copy_always ? hipErrorInvalidValue : hipHostGetDevicePointer(&landing, a->h_totals, 0);
    if (e != hipSuccess) {
      (void)hipGetLastError();
      landing = nullptr;
    }
    int rc;
    // the preprocess's workgroups also clear what the depth sort wants zeroed, which saves the sort its clearing kernel
    static const bool sort_clears = getenv("GSR_SORT_CLEARS_ITSELF") != nullptr;   // for measurements of the difference
    gsr::PreprocessExtras extras{landing != nullptr, nullptr, 0};
    if (!sort_clears) gsr::depth_sort_clear_region(P, w.sort_scratch, &extras.clear_at, &extras.clear_words);
    if (segments) {
      uint32_t* head = gsr::segments_head(segments, W, H);
      const size_t head_words = (size_t)((uint32_t*)w.sort_scratch - head);
      if (extras.clear_words) extras.clear_words += head_words; else extras.clear_words = head_words;
      extras.clear_at = head;
    }
    if (a->raw)
      rc = gsr::preprocess_fwd_raw(extras, P, a->D, a->M - 1, a->means3D, a->scales, a->scale_modifier, a->rotations, a->opacities,
                                  a->shs, a->shs_rest, a->viewmatrix, a->projmatrix, a->campos, W, H, a->tanfovx,
                                  a->tanfovy, a->reference_lists, w.records, a->radii, nullptr, w.depth_keys, nullptr,
                                  w.sort_vals, w.totals, a->status, a->visible, stream);
    else
      rc = gsr::preprocess_fwd(extras, P, a->D, a->M, a->means3D, a->scales, a->scale_modifier, a->rotations, a->opacities, a->shs,
                              a->colors_precomp, a->cov3D_precomp, a->viewmatrix, a->projmatrix, a->campos, W, H,
                              a->tanfovx, a->tanfovy, a->prefiltered, a->reference_lists, w.records, a->radii, nullptr,
                              w.depth_keys, nullptr, w.sort_vals, w.totals, a->status, a->visible, stream);
    if (rc) return rc;
    if ((rc = checkpoint("preprocess"))) return rc;
    if (a->after_preprocess) a->after_preprocess(a->hook_ctx);
    if (!landing) {
      e = hipMemcpyAsync(a->h_totals, w.totals, 8 * sizeof(uint32_t), hipMemcpyDeviceToHost, st);
      if (e == hipSuccess) e = hipEventRecord((hipEvent_t)a->event, st);
      if (e != hipSuccess) return hip_fail("read-back of the list sizes", e);
    }
    const uint32_t ticket = next_ticket();
    const gsr::SortMirror mirror{gsr::totals_partials(w.totals), gsr::totals_partial_blocks(P), a->status, w.totals,
                                 (uint32_t*)landing, ticket};
    rc = gsr::depth_sort(P, w.depth_keys, w.sort_vals, w.sorted_vals, w.sort_scratch, st, landing ? &mirror : nullptr, !sort_clears);
    if (rc) return rc;
    sort_error = gsr_bin_sort_error_word(w.sort_scratch);
    if (a->debug && (rc = gsr_bin_sort_status(P, w.sort_scratch, w.sort_bytes, stream))) return rc;
    if (landing) {
      if ((rc = await_ticket(a->h_totals, ticket, st))) return rc;
    } else {
      e = hipEventSynchronize((hipEvent_t)a->event);
      if (e != hipSuccess) return hip_fail("waiting for the list sizes", e);
    }
    if (a->h_totals[5])
      return gsr::fail(GSR_E_TOO_LARGE, "forward: the frame has more than %.0f tile instances (%d Gaussians on %dx%d pixels): "
                       "the lists' sizes are 32-bit, as the published module's are; nothing was rendered", (double)gsr::kMaxInstances, P, W, H);
    N = a->h_totals[0]; E1 = a->h_totals[1]; E2 = a->h_totals[2]; V = a->h_totals[3];
    out->carried_status = a->h_totals[4];
    if (out->carried_status)
      return gsr::fail(GSR_E_EARLIER, "forward: an earlier forward on this status word built no tile lists (status 0x%x)",
                       out->carried_status);
  }
  out->N = N; out->E1 = E1; out->E2 = E2; out->V = V;

  const size_t lists_bytes = gsr_bin_lists_bytes(V, E1, E2, W, H);
  const size_t list_words = N > 0 ? N : 1;
  // point_list | (tile ids) | scratch of the list builder, as one request
  Carve need(nullptr);
  need.take<uint32_t>(list_words * 4);
  if (a->keep_tile_ids) need.take<uint32_t>(list_words * 4);
  need.take<char>(lists_bytes);
  void* mem = resize(resize_ctx, need.at);
  if (!mem || ((uintptr_t)mem & 255))
    return gsr::fail(GSR_E_SCRATCH, "forward: the resize callback returned %s for %zu bytes", mem ? "misaligned memory" : "nothing",
                     need.at);
  Carve got(mem);
  out->point_list = got.take<uint32_t>(list_words * 4);
  out->tile_ids_sorted = a->keep_tile_ids ? got.take<uint32_t>(list_words * 4) : nullptr;
  void* lists_scratch = got.take<char>(lists_bytes);
  int rc = gsr_bin_lists(P, V, N, E1, E2, W, H, w.sorted_vals, out->point_list, w.ranges, out->tile_ids_sorted, sort_error,
                         P > 0 ? w.totals : nullptr, a->status, lists_scratch, lists_bytes, stream);
  if (rc) return rc;
  if ((rc = checkpoint("bin_lists"))) return rc;
  if (a->before_blend) a->before_blend(a->hook_ctx);
  // the caller's zero-fill rides on the blend kernel, whose workgroups leave HBM almost idle
  rc = gsr::blend_fwd(W, H, w.ranges, out->point_list, w.records, a->bg, a->out_color, w.final_T, w.n_contrib, a->zero_fill,
                      a->zero_fill_bytes, segments, gsr::segment_length_for(N), a->blend_start_event, a->blend_stop_event, st);
  if (rc) return rc;
  return checkpoint("blend");
}
