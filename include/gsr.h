/*
 * gsr.h — C ABI of the MI355X (gfx950) Gaussian rasterizer library, libgsr_hip.so.
 *
 * This is the boundary a binding of the reference would attach to.  The
 * reference reaches its rasterizer through the Python extension module
 * `diff_gaussian_rasterization` (import: gaussian_renderer/__init__.py:14;
 * settings: :36-49; module construction: :51; call: :85-93).  That module's
 * native half is an un-vendored submodule (.gitmodules:4-6, directory empty in
 * the reference tree), so each entry point below cites the reference CALL SITE
 * whose work it performs rather than a native symbol of the reference.
 *
 * Conventions
 *  - extern "C", plain pointers and sizes.  No torch / STL types.
 *  - Every data pointer is a DEVICE pointer unless its name starts with h_.
 *    The camera tensors (viewmatrix, projmatrix, campos, bg) are device
 *    pointers too: the reference keeps them on the GPU
 *    (scene/cameras.py:54-57, gaussian_renderer/__init__.py:22).
 *  - The library allocates nothing.  All outputs, state and scratch are caller
 *    memory (the Python layer hands out torch caching-allocator memory);
 *    scratch sizes come from the *_bytes functions.  Scratch must be 256-byte
 *    aligned.
 *  - The library keeps no state between calls and every entry point may be
 *    called from several host threads at once, each on its own stream, as long
 *    as no two calls in flight share an output, scratch or status buffer.  The
 *    exceptions are two process-wide settings: the patience of
 *    gsr_debug_sort_spin_limit (a test hook) and gsr_assume_exclusive_device.
 *  - `stream` is a hipStream_t passed as void* (NULL = the null stream).
 *    Calls only enqueue work; none synchronises.  The one host read-back the
 *    pipeline needs (the list sizes, after gsr_preprocess_fwd) is done by the
 *    caller; gsr_bin_sort_status, a check for debugging, is the one exception.
 *  - Return value: 0 = success, negative = error (GSR_E_*); gsr_last_error()
 *    returns a message for the calling thread's last failure.
 *  - Floating tensors are fp32, contiguous.  Layouts follow the reference's
 *    getters (scene/gaussian_model.py:95-118): means3D (P,3); scales (P,3);
 *    rotations (P,4) as (w,x,y,z); opacities (P,1); shs (P,M,3)
 *    coefficient-major, channel-minor; colors_precomp (P,3); cov3D_precomp
 *    (P,6) as (xx,xy,xz,yy,yz,zz) (utils/general_utils.py:64-73).
 */
#ifndef GSR_H
#define GSR_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GSR_VERSION 6

#define GSR_OK 0
#define GSR_E_INVALID (-1)   /* bad argument (null pointer, negative size, unsupported M/D) */
#define GSR_E_LAUNCH (-2)    /* HIP reported an error on launch / enqueue                    */
#define GSR_E_SCRATCH (-3)   /* scratch buffer too small or misaligned                       */
#define GSR_E_SORT (-4)      /* the depth sort gave up waiting (see gsr_bin_sort_status)     */
#define GSR_E_TOO_LARGE (-5) /* the input exceeds a fixed capacity (gsr_bin_sort_static,
                                GSR_MAX_INSTANCES)                                             */
/* Tile instances (pairs of a Gaussian and a tile it is binned into) one frame may
 * hold: 2^31 - 2^22.  A million Gaussians that each cover a whole 1080p frame
 * would be four times that. */
#define GSR_MAX_INSTANCES 2143289344u
#define GSR_E_EARLIER (-6)   /* gsr_forward: the status word carries an earlier call's failure */

/* Bits of the caller's device-side status word (see gsr_bin_lists). */
#define GSR_STATUS_SORT 1u   /* a list builder found that its depth sort had given up        */
#define GSR_STATUS_TOTALS 2u /* a list builder was handed list sizes other than the preprocess's */

/* One 64-byte record per Gaussian, written by gsr_preprocess_fwd and read by
 * the binning and blending kernels: everything a tile needs about a splat in
 * one aligned cache-line-sized gather.  Indices are in 4-byte words. */
#define GSR_REC_WORDS 16
#define GSR_REC_X 0        /* pixel-space mean x                                  */
#define GSR_REC_Y 1        /* pixel-space mean y                                  */
#define GSR_REC_HX 2       /* half extent in x of the alpha >= 1/255 ellipse      */
#define GSR_REC_HY 3       /* half extent in y                                    */
#define GSR_REC_CONIC_A 4  /* inverse 2D covariance (A, B, C) ...                 */
#define GSR_REC_CONIC_B 5
#define GSR_REC_CONIC_C 6
#define GSR_REC_OPACITY 7  /* ... and opacity: the float4 "conic_opacity"          */
#define GSR_REC_R 8        /* colour after SH evaluation / colors_precomp ...      */
#define GSR_REC_G 9
#define GSR_REC_B 10
#define GSR_REC_CLAMPED 11  /* ... and uint32: bit c set = colour channel c was clamped at 0 */
#define GSR_REC_DEPTH 12    /* view-space z                                        */
#define GSR_REC_RECT_MIN 13 /* uint32: tile rect min, x | y << 16                  */
#define GSR_REC_RECT_MAX 14 /* uint32: tile rect max (exclusive), x | y << 16      */
#define GSR_REC_RADIUS 15   /* int32: screen radius in px, 0 = culled               */

int gsr_version(void);
const char* gsr_last_error(void);

/* ---- Stage 1: preprocess ------------------------------------------------
 * Work of GaussianRasterizer.forward up to per-Gaussian screen-space state
 * (reference call site gaussian_renderer/__init__.py:85-93 with the settings of
 * :36-49): frustum cull (view z <= 0.2), projection, 3D covariance from
 * scales/rotations (or cov3D_precomp; Python twin scene/gaussian_model.py:27-31),
 * EWA 2D covariance, conic, radius, tile rectangle, SH -> RGB with +0.5 and
 * clamp at 0 (or colors_precomp; Python twin utils/sh_utils.py:57-112 and
 * gaussian_renderer/__init__.py:74-78).
 * Exactly one of shs / colors_precomp and one of (scales, rotations) /
 * cov3D_precomp is non-NULL.  D = active SH degree (0..3), M = coefficients
 * stored per Gaussian and channel (>= (D+1)^2).
 * Outputs: records (P,16) words; radii (P) int32 (0 = not visible);
 * depth_keys (P) uint32 = IEEE bits of view z for visible Gaussians, 0xFFFFFFFF
 * for culled ones; optionally (NULL = not wanted; nothing downstream reads them)
 * tiles_touched (P) uint32 and rects (P,2) uint32 = the tile rectangle as in the
 * record (min, exclusive max; x | y << 16), zeros for culled Gaussians; and what
 * the binning starts from:
 * sort_vals (P) uint64 = the Gaussian's index packed with its tile rectangle
 * (bit widths chosen from P and the tile grid; the call fails if they exceed
 * 64 bits, e.g. more than 2^24 Gaussians on a grid of more than 1024x1024
 * tiles), the value the depth sort carries;
 * totals[0..3] = { N = tile instances (sum of tiles_touched), E1 = supercell
 * entries, E2 = cell entries, V = visible Gaussians }, which the caller reads
 * back (the pipeline's one host read-back; it can overlap gsr_bin_sort) to size
 * gsr_bin_lists.  totals has gsr_totals_words(P) words (the rest is the call's
 * scratch) and is 16-byte aligned.
 * status (may be NULL): the caller's device-side status word, see gsr_bin_lists;
 * its value at this point is copied to totals[4], so that it reaches the host with
 * the list sizes at no extra cost.
 * totals[5] = 1 if the frame holds GSR_MAX_INSTANCES tile instances or more: the
 * sizes are 32-bit sums (the published module's instance count is an int) and
 * have wrapped or are about to; the caller must not size lists by them
 * (gsr_forward stops with GSR_E_TOO_LARGE).  totals[6..7] are reserved.
 *
 * visible (P) uint8, optional (may be NULL): radii > 0, the `visibility_filter`
 * render() returns (gaussian_renderer/__init__.py:99).
 *
 * reference_lists: which tiles a Gaussian is binned into.  Non-zero (what the
 * Python module passes unless told otherwise): every tile of its 3-sigma
 * rectangle, as the module's duplicateWithKeys does; the tile lists, keys and
 * ranges of stage 2 are then the module's, bit for bit.  Zero:
 * only the tiles of that rectangle which the bounding box of its
 * alpha >= 1/255 ellipse reaches (no pixel outside that box can blend it):
 * shorter lists, the same images, radii and gradients.  A Gaussian whose box
 * reaches no tile keeps its radius and its place in `visible`, gets the culled
 * depth key and is not counted in totals[3].  records, radii, tiles_touched and
 * rects are the same in both modes. */
size_t gsr_totals_words(int P);
int gsr_preprocess_fwd(int P, int D, int M, const float* means3D, const float* scales,
                       float scale_modifier, const float* rotations, const float* opacities,
                       const float* shs, const float* colors_precomp, const float* cov3D_precomp,
                       const float* viewmatrix, const float* projmatrix, const float* campos,
                       int W, int H, float tanfovx, float tanfovy, int prefiltered,
                       int reference_lists, uint32_t* records, int32_t* radii, uint32_t* tiles_touched,
                       uint32_t* depth_keys, uint32_t* rects, uint64_t* sort_vals,
                       uint32_t* totals, const uint32_t* status, uint8_t* visible, void* stream);

/* GaussianRasterizer.markVisible of the module (never called by the
 * reference's Python): present[i] = view-space z > 0.2. */
int gsr_mark_visible(int P, const float* means3D, const float* viewmatrix, uint8_t* present,
                     void* stream);

/* ---- Stage 2: binning ----------------------------------------------------
 * Produces, for every 16x16 tile, the list of Gaussians whose tile rectangle
 * covers it, ordered front to back: the result of the module's
 * duplicateWithKeys -> 64-bit (tile | depth) radix sort -> identifyTileRanges
 * chain, bit for bit (ties in (tile, depth) keep ascending Gaussian index).
 * No per-tile instance is sorted: the Gaussians are depth-sorted once (P keys),
 * then three stable range partitions split that list by SUPERCELL (16x16
 * tiles), by CELL (tile row x strip of 16 tiles) and by tile.  Every step is
 * stable, so the lists equal the 64-bit sort's.
 *
 * No index is ever dereferenced on the way: the depth sort carries, as its
 * 64-bit value, the Gaussian's index packed with its tile rectangle
 * (sort_vals of stage 1), and that word is what the partitions read.
 * V, N, E1, E2 below are the totals of stage 1, read back by the caller. */

/* gsr_bin_sort: sorted_vals[0 .. V) = the sort_vals of the V visible Gaussians
 * (depth key != 0xFFFFFFFF) ordered by ascending depth key, ties by ascending
 * index (stable).  The culled ones are dropped: sorted_vals[V .. P) is not
 * written.  P < 2^30. */
size_t gsr_bin_sort_bytes(int P);
int gsr_bin_sort(int P, const uint32_t* depth_keys, const uint64_t* sort_vals,
                 uint64_t* sorted_vals, void* scratch, size_t scratch_bytes, void* stream);

/* The sort's workgroups wait for each other's digit counts; every wait is
 * bounded, and a workgroup that runs out of patience flags the sort as failed
 * instead of hanging (never seen in a healthy run: a workgroup only waits for
 * workgroups that started before it).  gsr_bin_sort_status WAITS for `stream`,
 * then returns GSR_OK, or GSR_E_SORT if the sort that last used this scratch
 * gave up (sorted_vals is then incomplete).  The Python module calls it when
 * the settings' `debug` flag is set (gaussian_renderer/__init__.py:48), as the
 * replaced module synchronises and checks for errors in that mode.
 * A sort that gave up writes nothing outside its scratch and sorted_vals[0 .. P).
 * gsr_bin_sort_error_word returns the DEVICE address, inside the sort's scratch,
 * of the word that is non-zero after such a sort: hand it to gsr_bin_lists, which
 * then refuses to build lists from an incomplete sort, without any host wait.
 * gsr_debug_sort_spin_limit sets the patience of later sorts in look-back
 * steps (0 = default); it exists for tests of this path. */
int gsr_bin_sort_status(int P, const void* scratch, size_t scratch_bytes, void* stream);
const uint32_t* gsr_bin_sort_error_word(const void* sort_scratch);
void gsr_debug_sort_spin_limit(uint32_t limit);

/* gsr_assume_exclusive_device(1): the caller asserts that nothing else runs on the GPU while
 * its sorts do (no other process, no second stream with kernels of its own, no CU mask).  A
 * sort whose whole grid is resident at once then gives workgroup b chunk b instead of drawing
 * a ticket per workgroup (1.5 us less per pass at a million keys).  Without the assertion
 * (the default; also GSR_EXCLUSIVE_DEVICE=1 in the environment at first use sets it) every
 * workgroup draws a ticket when it starts running, so a workgroup only ever waits for
 * workgroups that are running or done, whatever else occupies the GPU.  Process-wide; has no
 * counterpart in the reference (its sort is a library call without cross-workgroup waits). */
void gsr_assume_exclusive_device(int on);

/* gsr_bin_sort_static: gsr_bin_sort for callers that cannot allocate device memory or have
 * none to hand in (code under HIP graph capture, tight serving loops, harnesses that forbid
 * hipMalloc).  It has no counterpart in the reference.  The sort's scratch is a 26,312,960-byte
 * device array inside the library (every GPU the library runs on holds its own copy until the
 * process ends), so P is limited to GSR_BIN_SORT_STATIC_MAX = 2^20 keys: more is refused with
 * GSR_E_TOO_LARGE.  The call allocates nothing (no hipMalloc / hipFree, no allocating library
 * routine); it only enqueues kernels on `stream`, whose device must be current.
 * The FIRST call on each device also looks up the array's address and the kernels' occupancy
 * (which may load the code object and materialise the array): make that call, with P = 0 if
 * there is nothing to sort yet, outside graph capture or any other section that forbids
 * allocation; every later call on that device does neither.
 * gsr_bin_sort_static_bytes(0) = the array's size, (1) = what the sort's layout needs at
 * GSR_BIN_SORT_STATIC_MAX keys (never more than the former).
 * Same result as gsr_bin_sort, bit for bit.  *error_word (optional out, may be NULL) receives the
 * device address of the sort's error word, for gsr_bin_lists' sort_error argument.
 * All calls on one GPU share that GPU's array: calls that overlap in time on one GPU (two streams,
 * two threads) race and give undefined results; calls that follow each other on one stream are fine,
 * each clears what it needs to start from zero. */
#define GSR_BIN_SORT_STATIC_MAX (1 << 20)
int gsr_bin_sort_static(int P, const uint32_t* depth_keys, const uint64_t* sort_vals,
                        uint64_t* sorted_vals, const uint32_t** error_word, void* stream);
size_t gsr_bin_sort_static_bytes(int which);

/* gsr_bin_lists: point_list (N) uint32 = Gaussian index per sorted instance;
 * ranges (tiles,2) uint32 = [start,end) into point_list per tile, (0,0) for
 * empty tiles.
 * tile_ids_sorted (N) uint32 is optional (may be NULL): the tile id of each
 * sorted instance, from which the module's 64-bit sort keys are
 * (tile << 32) | depth_key[point_list].
 *
 * Three optional device pointers make the call safe against what the host cannot
 * see when it enqueues it (each may be NULL = not checked):
 *   sort_error  gsr_bin_sort_error_word of the sort that produced sorted_vals;
 *   totals      the totals array of stage 1 (the device's own N, E1, E2, V);
 *   status      one uint32 of the caller's, zeroed once, that outlives the call.
 * If the sort gave up, or V / N / E1 / E2 differ from the device's totals (a stale
 * or foreign read-back: the buffers were sized for other lists), the builder writes
 * no list: every tile range becomes (0,0), so the image is the background and the
 * backward a no-op, and GSR_STATUS_SORT / GSR_STATUS_TOTALS is OR-ed into *status.
 * The next gsr_preprocess_fwd that is given the same status word carries it to the
 * host in totals[4]: the failure surfaces one call late, for free, and nothing is
 * ever written out of bounds in between. */
size_t gsr_bin_lists_bytes(uint32_t V, uint32_t E1, uint32_t E2, int W, int H);
int gsr_bin_lists(int P, uint32_t V, uint32_t N, uint32_t E1, uint32_t E2, int W, int H,
                  const uint64_t* sorted_vals, uint32_t* point_list, uint32_t* ranges,
                  uint32_t* tile_ids_sorted, const uint32_t* sort_error, const uint32_t* totals,
                  uint32_t* status, void* scratch, size_t scratch_bytes, void* stream);

/* ---- Stage 3: blend forward ----------------------------------------------
 * Front-to-back alpha compositing per tile (the module's per-tile render
 * kernel): out_color (3,H,W) = sum_i c_i alpha_i T_i + T_final * bg, the
 * image the reference receives at gaussian_renderer/__init__.py:85.
 * final_T (H,W) fp32 and n_contrib (H,W) uint32 (1-based position in the
 * tile list of the pixel's last contributor) are saved for backward. */
int gsr_blend_fwd(int W, int H, const uint32_t* ranges, const uint32_t* point_list,
                  const uint32_t* records, const float* bg, float* out_color, float* final_T,
                  uint32_t* n_contrib, void* stream);

/* ---- The forward in one call -------------------------------------------------
 * Stages 1-3 behind one entry point, the shape of the replaced module's own
 * native forward (one call per GaussianRasterizer.forward,
 * gaussian_renderer/__init__.py:85-93; memory whose size is only known half way
 * obtained through a callback into the host's allocator).  Results are those of
 * the stage entry points, bit for bit; what it saves is host time: one transition
 * into native code and two or three buffers per render instead of a dozen.
 *
 * The caller passes the inputs of stage 1, the user-visible outputs (out_color,
 * radii, optionally visible), one WORKSPACE of gsr_forward_workspace_bytes(P, W, H)
 * bytes (256-byte aligned) for everything else whose size is known up front, a
 * page-locked host buffer of 8 words and an event (gsr_event_create) for the
 * read-back, and optionally a status word (see gsr_bin_lists).  When the GPU can
 * address the host buffer (hipHostMalloc / torch pinned memory: the usual case)
 * the depth sort's first kernel sums the list sizes, stores them there itself,
 * and behind them a ticket in the buffer's last word, which the call polls for:
 * no kernel of its own for the sums, no copy command, and the event stays
 * unused (an event behind that kernel costs the stream five microseconds);
 * pageable memory gets an ordinary copy and the event.  Once the list
 * sizes have arrived, `resize(resize_ctx, bytes)` is called ONCE for the memory
 * of the tile lists and the list builder's scratch (256-byte aligned device
 * memory, or NULL to fail); it must stay valid as long as the lists are used.
 * The call waits for the read-back (nothing else), and in `debug` mode after
 * every stage.  GsrForwardOut receives the list sizes and the device addresses
 * of what a backward pass (and a test) needs: records, ranges, final_T,
 * n_contrib, depth_keys, sorted_vals inside the workspace, point_list (and
 * tile_ids_sorted when keep_tile_ids is set) inside the resize memory.
 * raw != 0: the inputs are the model's raw parameters as for
 * gsr_preprocess_fwd_raw (shs = features_dc, shs_rest = features_rest, M = 1 +
 * n_rest, opacities / scales / rotations un-activated).
 * after_preprocess, before_blend (optional, both get hook_ctx): called on the host
 * right after stage 1 has been enqueued, and after the list builder has been
 * enqueued and before the blend is: e.g. to start work on another stream beside the
 * depth sort and to make `stream` wait for it before the blend.
 * zero_fill, zero_fill_bytes (optional; 16-byte aligned, a multiple of 16 bytes):
 * device memory of the caller's that is zero when the call's work on `stream`
 * is done, e.g. the (P,16) accumulator of gsr_blend_bwd.  Every workgroup of
 * the blend kernel clears an equal share of it as its last act: that kernel
 * leaves HBM almost idle, so 64 MB cost it 3 us where a fill kernel of its own
 * takes 10.
 * If totals[4] shows that an earlier call on the same status word built no
 * lists, the call stops before its own lists with GSR_E_EARLIER
 * (out->carried_status holds the bits; clearing the word is the caller's).
 * A frame of GSR_MAX_INSTANCES tile instances or more stops at the same point
 * with GSR_E_TOO_LARGE: nothing is rendered, later calls are not affected. */
typedef void* (*gsr_resize_fn)(void* ctx, size_t bytes);
typedef struct GsrForwardArgs {
  int P, D, M, raw;
  const float* means3D; const float* scales; const float* rotations; const float* opacities;
  const float* shs; const float* shs_rest; const float* colors_precomp; const float* cov3D_precomp;
  const float* viewmatrix; const float* projmatrix; const float* campos; const float* bg;
  float scale_modifier, tanfovx, tanfovy;
  int W, H, prefiltered, reference_lists, debug, keep_tile_ids;
  float* out_color; int32_t* radii; uint8_t* visible;
  void* workspace; size_t workspace_bytes;
  uint32_t* h_totals; void* event; uint32_t* status;
  void (*after_preprocess)(void* ctx); void (*before_blend)(void* ctx); void* hook_ctx;
  void* blend_start_event; void* blend_stop_event;   /* optional, as for gsr_blend_fwd_timed */
  void* zero_fill; size_t zero_fill_bytes;           /* optional, see above */
} GsrForwardArgs;
typedef struct GsrForwardOut {
  uint32_t N, V, E1, E2, carried_status;
  uint32_t* records; uint32_t* depth_keys; uint64_t* sorted_vals; uint32_t* ranges; float* final_T;
  uint32_t* n_contrib; uint32_t* point_list; uint32_t* tile_ids_sorted;
  void* segments;   /* inside the workspace; for gsr_blend_bwd_segments (NULL: none were kept) */
} GsrForwardOut;
size_t gsr_forward_workspace_bytes(int P, int W, int H);
int gsr_forward(const GsrForwardArgs* args, gsr_resize_fn resize, void* resize_ctx, GsrForwardOut* out,
                void* stream);

/* ---- Stage 4: blend backward ---------------------------------------------
 * dL_dpix (3,H,W) -> per-Gaussian sums, accumulated into dL_dsplat (P,16) fp32
 * (one 64-byte-aligned row per Gaussian, so that the nine sums of a splat leave
 * a wave as ONE atomic request) which the caller zero-fills first and which
 * must be 64-byte aligned.  With u = dL/dG * G per (pixel, splat) and
 * d = splat centre - pixel, the words are the moments
 *   0: sum u   1: sum u dx   2: sum u dy   3: sum u dx^2   4: sum u dx dy
 *   5: sum u dy^2   6-8: dL/dcolour   9-15: unused
 * from which gsr_preprocess_bwd forms dL/dopacity, dL/dmean2D and dL/dconic
 * once per Gaussian (they are linear in the moments). */
#define GSR_GRAD_WORDS 16
int gsr_blend_bwd(int W, int H, const uint32_t* ranges, const uint32_t* point_list,
                  const uint32_t* records, const float* bg, const float* final_T,
                  const uint32_t* n_contrib, const float* dL_dpix, float* dL_dsplat,
                  void* stream);
/* The same with two extras for gsr_preprocess_bwd's outputs_zeroed mode.
 * touched (optional, P bytes that the caller has zeroed, e.g. as part of
 * gsr_forward's zero_fill): set to 1 for every Gaussian whose row of dL_dsplat
 * the call may add to; the rows of the others are still all zeros.
 * zero_fill, zero_fill_bytes: a zero-fill like gsr_forward's (16-byte aligned, a
 * multiple of 16 bytes, none of it inside dL_dsplat or touched; 0 = none): zero
 * when the call's work on `stream` is done, cleared by the blend kernel's
 * workgroups as their last act.  Meant for the outputs of gsr_preprocess_bwd
 * (248 MB at a million Gaussians cost this kernel 9 us; a fill kernel of their
 * own takes 36).  start_event / stop_event as for gsr_blend_bwd_timed, both may
 * be NULL.  The reference has no counterpart: the
 * backward of its rasterizer submodule (diff-gaussian-rasterization, RasterizeGaussiansBackwardCUDA)
 * allocates its gradient tensors zeroed. */
int gsr_blend_bwd_fill(int W, int H, const uint32_t* ranges, const uint32_t* point_list,
                       const uint32_t* records, const float* bg, const float* final_T,
                       const uint32_t* n_contrib, const float* dL_dpix, float* dL_dsplat,
                       uint8_t* touched, void* zero_fill, size_t zero_fill_bytes, void* start_event,
                       void* stop_event, void* stream);

/* gsr_blend_bwd_fill for the lists of a gsr_forward call, with the depth segments that call kept
 * (GsrForwardOut.segments; NULL = gsr_blend_bwd_fill).  A tile's backward walk is serial, so a single
 * tile whose pixels stay translucent many times deeper than the others' (a dense cluster right after the
 * reference's reset_opacity, scene/gaussian_model.py:210-213, when every opacity is at most 0.01) holds the
 * whole kernel up.  gsr_forward's blend therefore writes down every pixel's transmittance and composited
 * colour each time a tile's walk crosses a multiple of L list entries with a pixel still open (L = 2048, or
 * N / 8192 rounded up to a multiple of 256 if that is more), into a pool of 8192 such records inside the
 * workspace (32 MB, enough for every boundary there can be), and this call walks each of those stretches of
 * L entries in a workgroup of its own, beside the workgroup of the tile.  Same sums as gsr_blend_bwd up to rounding (the atomic additions arrive in another order; the
 * state at a boundary is the forward's, not the one unwound from the pixel's end).  Scenes whose tiles are
 * walked less than 2048 entries deep (every ordinary one) cost the forward nothing and this call a
 * microsecond.  The reference has no counterpart: its backward (diff-gaussian-rasterization,
 * RasterizeGaussiansBackwardCUDA) walks every tile in one block.
 * For callers of the stage entry points: gsr_blend_fwd_segments is gsr_blend_fwd_timed (events optional) that
 * also keeps the segments, in a block of gsr_segments_bytes(W, H) bytes (256-byte aligned) whose LAST 256 bytes
 * the caller has zeroed on `stream`; N = the number of list entries (gsr_bin_lists' N).  A block with those
 * bytes zero that no forward has written to means "no segments" to gsr_blend_bwd_segments.
 * gsr_debug_segments: from now on, for this process, `entries` instead of 2048 (rounded up to a multiple of
 * 256; 0 = gsr_forward keeps no segments) and a pool of `pool` records (0 = all 8192; with fewer a forward
 * can find the pool empty, and the tiles that come too late are walked whole), for tests of the path on small
 * scenes. */
size_t gsr_segments_bytes(int W, int H);
int gsr_blend_fwd_segments(int W, int H, const uint32_t* ranges, const uint32_t* point_list,
                           const uint32_t* records, const float* bg, float* out_color, float* final_T,
                           uint32_t* n_contrib, void* segments, size_t segments_bytes, uint32_t N,
                           void* start_event, void* stop_event, void* stream);
int gsr_blend_bwd_segments(int W, int H, const uint32_t* ranges, const uint32_t* point_list,
                           const uint32_t* records, const float* bg, const float* final_T,
                           const uint32_t* n_contrib, const float* dL_dpix, float* dL_dsplat,
                           uint8_t* touched, void* zero_fill, size_t zero_fill_bytes, const void* segments,
                           void* start_event, void* stop_event, void* stream);
void gsr_debug_segments(uint32_t entries, uint32_t pool);

/* ---- Kernel-exact timing of the two blend kernels ----------------------------
 * The same calls with a pair of events (hipEvent_t passed as void*, from
 * gsr_event_create; either may be NULL) that the launch stamps with the kernel's
 * own start and end, the interval a profiler's kernel trace reports: events
 * recorded on the stream around a call also measure the gap before the kernel
 * starts.  gsr_event_elapsed_ms waits for stop_event and writes the milliseconds
 * between the two to *h_ms (host memory).  Measurement only: the reference has
 * no counterpart (its two torch events span a whole iteration, train.py:44-45,67,95). */
void* gsr_event_create(void);
void* gsr_event_create_untimed(void);   /* for ordering and waiting only (cheaper to record) */
void gsr_event_destroy(void* event);
int gsr_event_elapsed_ms(void* start_event, void* stop_event, float* h_ms);
int gsr_blend_fwd_timed(int W, int H, const uint32_t* ranges, const uint32_t* point_list,
                        const uint32_t* records, const float* bg, float* out_color, float* final_T,
                        uint32_t* n_contrib, void* start_event, void* stop_event, void* stream);
int gsr_blend_bwd_timed(int W, int H, const uint32_t* ranges, const uint32_t* point_list,
                        const uint32_t* records, const float* bg, const float* final_T,
                        const uint32_t* n_contrib, const float* dL_dpix, float* dL_dsplat,
                        void* start_event, void* stop_event, void* stream);

/* ---- Stage 5: preprocess backward ----------------------------------------
 * dL_dsplat -> gradients of the op's inputs (what autograd hands back to the
 * tensors passed at gaussian_renderer/__init__.py:85-93).  Outputs are fully
 * written (zeros for culled Gaussians): dL_dmeans3D (P,3); dL_dmeans2D (P,3)
 * = (x, y, 0); dL_dopacities (P,1); and, matching the inputs that were given,
 * dL_dshs (P,M,3) or dL_dcolors (P,3), dL_dscales (P,3) + dL_drotations (P,4)
 * or dL_dcov3D (P,6).  Pointers of the unused alternatives are NULL.
 * outputs_zeroed != 0: the caller promises that every output is zero already
 * (e.g. cleared by gsr_blend_bwd_fill).  Every gradient is linear in
 * the Gaussian's row of dL_dsplat, so a Gaussian with an all-zero row (culled, or
 * never blended because it is hidden: most Gaussians of a dense scene) has zero
 * gradients; the kernel then writes only the rows of the others and skips the
 * 248 bytes of zeros per Gaussian that dominate its traffic otherwise.
 * touched (optional, only with outputs_zeroed; P bytes as written by
 * gsr_blend_bwd_fill): a Gaussian whose byte is 0 is taken to have an all-zero
 * row without the row being read (64 bytes per Gaussian less to scan). */
int gsr_preprocess_bwd(int P, int D, int M, const float* means3D, const float* scales,
                       float scale_modifier, const float* rotations, const float* shs,
                       const float* cov3D_precomp, const float* viewmatrix,
                       const float* projmatrix, const float* campos, int W, int H, float tanfovx,
                       float tanfovy, const uint32_t* records, const float* dL_dsplat,
                       float* dL_dmeans3D, float* dL_dmeans2D, float* dL_dopacities,
                       float* dL_dshs, float* dL_dcolors, float* dL_dscales,
                       float* dL_drotations, float* dL_dcov3D, int outputs_zeroed,
                       const uint8_t* touched, void* stream);

/* ---- Raw-parameter variants of the per-Gaussian stages -----------------------
 * The reference applies its activations in separate PyTorch kernels on every
 * call (scene/gaussian_model.py:95-118: exp, sigmoid, normalize, and a
 * concatenation of _features_dc (P,1,3) with _features_rest (P,n_rest,3)) and
 * differentiates through them again in backward.  These two entry points take
 * the RAW parameters and fold the activations, and their chain rule, into the
 * kernels; everything downstream (binning, blending) is unchanged.
 * n_rest = coefficients in features_rest (15 for degree 3; 0 = no rest tensor).
 * Gradient outputs are with respect to the raw parameters. */
int gsr_preprocess_fwd_raw(int P, int D, int n_rest, const float* xyz, const float* scaling_raw,
                           float scale_modifier, const float* rotation_raw, const float* opacity_raw,
                           const float* features_dc, const float* features_rest,
                           const float* viewmatrix, const float* projmatrix, const float* campos,
                           int W, int H, float tanfovx, float tanfovy, int reference_lists,
                           uint32_t* records, int32_t* radii, uint32_t* tiles_touched,
                           uint32_t* depth_keys, uint32_t* rects, uint64_t* sort_vals,
                           uint32_t* totals, const uint32_t* status, uint8_t* visible, void* stream);
int gsr_preprocess_bwd_raw(int P, int D, int n_rest, const float* xyz, const float* scaling_raw,
                           float scale_modifier, const float* rotation_raw, const float* opacity_raw,
                           const float* features_dc, const float* features_rest,
                           const float* viewmatrix, const float* projmatrix, const float* campos,
                           int W, int H, float tanfovx, float tanfovy, const uint32_t* records,
                           const float* dL_dsplat, float* dL_dxyz, float* dL_dmeans2D,
                           float* dL_dopacity_raw, float* dL_dfeatures_dc, float* dL_dfeatures_rest,
                           float* dL_dscaling_raw, float* dL_drotation_raw, int outputs_zeroed,
                           const uint8_t* touched, void* stream);

/* ---- Photometric loss ------------------------------------------------------
 * The step that follows the rasterizer in every training iteration
 * (train.py:90-92): loss = (1 - lambda) * mean|img - gt| + lambda * (1 - SSIM),
 * SSIM as utils/loss_utils.py:33-63 (11x11 Gaussian window, sigma 1.5, zero
 * padding, per channel).  img, gt: (3,H,W).  dmaps: (3,3,H,W) scratch written by
 * forward and read by backward; partial: (gsr_loss_partial_count, 2) scratch;
 * out3: (loss, L1 mean, SSIM mean).  grad_out: device scalar, the upstream
 * gradient of the loss; grad_img: (3,H,W). */
size_t gsr_loss_partial_count(int C, int H, int W);
int gsr_loss_fwd(int C, int H, int W, const float* img, const float* gt, float lambda_dssim,
                 float* dmaps, float* partial, float* out3, void* stream);
int gsr_loss_bwd(int C, int H, int W, const float* img, const float* gt, const float* dmaps,
                 const float* grad_out, float lambda_dssim, float* grad_img, void* stream);

/* ---- Optimiser step ----------------------------------------------------------
 * The update that follows every backward in the reference's loop
 * (train.py:126-128): torch.optim.Adam over the model's six parameter tensors,
 * one learning rate per tensor, eps = 1e-15, betas (0.9, 0.999), no weight decay
 * (scene/gaussian_model.py:154-163).  One launch updates all tensors in place:
 * param, exp_avg, exp_avg_sq (n floats each) from grad * grad_scale.  `step`
 * is the 1-based count of updates including this one (bias correction).
 * h_tensors is a HOST array of `count` <= GSR_ADAM_MAX_TENSORS descriptors whose
 * pointers are device pointers; `vec4` is filled in by the library. */
#define GSR_ADAM_MAX_TENSORS 8
typedef struct GsrAdamTensor {
  float* param;
  const float* grad;
  float* exp_avg;
  float* exp_avg_sq;
  size_t n;
  float lr;
  int vec4;
} GsrAdamTensor;
int gsr_adam_step(int count, const GsrAdamTensor* h_tensors, double beta1, double beta2, double eps,
                  int step, float grad_scale, void* stream);

/* ---- Initial scales: mean squared distance to the three nearest neighbours ----
 * distCUDA2 of the reference's second native submodule, simple_knn (call site
 * scene/gaussian_model.py:134, used once, to initialise the scales; the
 * submodule directory is empty in the reference tree).  points (P,3) ->
 * mean_dist2 (P): mean of the three smallest squared distances to the other
 * points (of fewer when P < 4; 0 when P = 1), exact, each distance computed as
 * ((dx*dx + dy*dy) + dz*dz) in fp32 and the mean as ((d0 + d1) + d2) / 3 with
 * d0 <= d1 <= d2. */
size_t gsr_knn3_bytes(int P);
int gsr_knn3_mean_dist2(int P, const float* points, float* mean_dist2, void* scratch,
                        size_t scratch_bytes, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* GSR_H */
