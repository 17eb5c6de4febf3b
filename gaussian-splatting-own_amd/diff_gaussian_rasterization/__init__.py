"""Drop-in ``diff_gaussian_rasterization`` for AMD MI355X (gfx950).

Same module name and public surface as the extension the reference imports
(``gaussian_renderer/__init__.py:14``): ``GaussianRasterizationSettings`` with
the 12 fields the reference constructs by keyword (``:36-49``) and
``GaussianRasterizer`` (an ``nn.Module`` built with ``raster_settings=``,
``:51``) whose call takes ``means3D, means2D, shs, colors_precomp, opacities,
scales, rotations, cov3D_precomp`` and returns ``(color (3,H,W) float32,
radii (P,) int32)`` (``:85-93``).  Gradients reach means3D, the means2D sink
(as (x, y, 0) in NDC-scaled units, consumed by
``scene/gaussian_model.py:405-407``), shs / colors_precomp, opacities, scales,
rotations / cov3D_precomp.

All arithmetic runs in hand-written HIP kernels inside ``libgsr_hip.so``,
reached through the C ABI of ``include/gsr.h`` with raw device pointers.
PyTorch only supplies device memory (caching allocator) and the current
stream.  There is no fallback path: if the library is missing the import
fails.
"""
from __future__ import annotations

import ctypes as C
import os
import threading
from typing import NamedTuple, Optional

import torch
from torch import nn

_HERE = os.path.dirname(os.path.abspath(__file__))
# GSR_HIP_LIB points at another build of the same library (kernel experiments)
_LIB_PATH = os.environ.get("GSR_HIP_LIB") or os.path.join(_HERE, "libgsr_hip.so")

TILE = 16
REC_WORDS = 16
GRAD_WORDS = 16
# By default every Gaussian is binned into all tiles of its 3-sigma rectangle, as the module this
# replaces does (its internal tile lists, bit for bit).  GSR_SHORT_LISTS=1 leaves out the tiles that
# the bounding box of the Gaussian's alpha >= 1/255 ellipse does not reach: the same images, radii and
# gradients from lists a third shorter (render 2-3 % faster at 1M-6M Gaussians).
REFERENCE_LISTS = os.environ.get("GSR_SHORT_LISTS", "0") in ("", "0")
# GSR_OVERLAP=1 clears the backward's accumulator on a second stream beside the depth sort and the list
# builder instead of on the caller's stream right before the blend backward.  Off by default: kernels of
# two streams do not hide beside each other here, the binning slows down by more than the 13 us the fill
# takes (1M Gaussians: train step 0.905 ms without, 0.92 ms with; tools/exp_overlap.py).
_OVERLAP = os.environ.get("GSR_OVERLAP", "0") not in ("", "0")
# In the single call the library clears that accumulator itself, at the end of the blend kernel's workgroups
# (that kernel leaves HBM almost idle), and the blend backward clears the gradient tensors of the per-Gaussian
# backward the same way.  GSR_SEPARATE_FILLS=1, for measurements of the difference: the accumulator is cleared by
# a fill kernel in the backward and the per-Gaussian backward writes every row of its outputs.
_CARRIED_FILLS = os.environ.get("GSR_SEPARATE_FILLS", "0") in ("", "0")
# The forward is ONE call into the library (gsr_forward) unless GSR_STAGED=1 or a stage timer is installed;
# then it is the sequence of stage calls, with the same results bit for bit.
SINGLE_CALL = os.environ.get("GSR_STAGED", "0") in ("", "0")
# GSR_NO_SEGMENTS=1, for measurements of the difference: the forward keeps no depth segments for the blend backward
# (gsr_blend_bwd_segments in gsr.h); the library reads the same variable for the single call.
_SEGMENTS = os.environ.get("GSR_NO_SEGMENTS") is None


class _ForwardArgs(C.Structure):
    """GsrForwardArgs of gsr.h."""
    _fields_ = [("P", C.c_int), ("D", C.c_int), ("M", C.c_int), ("raw", C.c_int),
                ("means3D", C.c_void_p), ("scales", C.c_void_p), ("rotations", C.c_void_p), ("opacities", C.c_void_p),
                ("shs", C.c_void_p), ("shs_rest", C.c_void_p), ("colors_precomp", C.c_void_p),
                ("cov3D_precomp", C.c_void_p), ("viewmatrix", C.c_void_p), ("projmatrix", C.c_void_p),
                ("campos", C.c_void_p), ("bg", C.c_void_p),
                ("scale_modifier", C.c_float), ("tanfovx", C.c_float), ("tanfovy", C.c_float),
                ("W", C.c_int), ("H", C.c_int), ("prefiltered", C.c_int), ("reference_lists", C.c_int),
                ("debug", C.c_int), ("keep_tile_ids", C.c_int),
                ("out_color", C.c_void_p), ("radii", C.c_void_p), ("visible", C.c_void_p),
                ("workspace", C.c_void_p), ("workspace_bytes", C.c_size_t),
                ("h_totals", C.c_void_p), ("event", C.c_void_p), ("status", C.c_void_p),
                ("after_preprocess", C.c_void_p), ("before_blend", C.c_void_p), ("hook_ctx", C.c_void_p),
                ("blend_start_event", C.c_void_p), ("blend_stop_event", C.c_void_p),
                ("zero_fill", C.c_void_p), ("zero_fill_bytes", C.c_size_t)]


class _ForwardOut(C.Structure):
    """GsrForwardOut of gsr.h."""
    _fields_ = [("N", C.c_uint32), ("V", C.c_uint32), ("E1", C.c_uint32), ("E2", C.c_uint32),
                ("carried_status", C.c_uint32),
                ("records", C.c_void_p), ("depth_keys", C.c_void_p), ("sorted_vals", C.c_void_p),
                ("ranges", C.c_void_p), ("final_T", C.c_void_p), ("n_contrib", C.c_void_p),
                ("point_list", C.c_void_p), ("tile_ids_sorted", C.c_void_p), ("segments", C.c_void_p)]


_RESIZE_FN = C.CFUNCTYPE(C.c_void_p, C.c_void_p, C.c_size_t)
_HOOK_FN = C.CFUNCTYPE(None, C.c_void_p)
GSR_E_EARLIER = -6
GSR_E_TOO_LARGE = -5
MAX_INSTANCES = 2143289344      # GSR_MAX_INSTANCES in gsr.h


def _load():
    if not os.path.exists(_LIB_PATH):
        raise ImportError(
            f"{_LIB_PATH} not found: build it with `make -C gaussian-splatting-own_amd/csrc` "
            "(or python -c 'import __graft_entry__ as g; g.build()'); there is no non-HIP fallback")
    lib = C.CDLL(_LIB_PATH)
    lib.gsr_last_error.restype = C.c_char_p
    lib.gsr_bin_sort_bytes.restype = C.c_size_t
    lib.gsr_bin_sort_bytes.argtypes = [C.c_int]
    lib.gsr_bin_lists_bytes.restype = C.c_size_t
    lib.gsr_bin_lists_bytes.argtypes = [C.c_uint32, C.c_uint32, C.c_uint32, C.c_int, C.c_int]
    vp, ci, cf = C.c_void_p, C.c_int, C.c_float
    lib.gsr_totals_words.restype = C.c_size_t
    lib.gsr_totals_words.argtypes = [ci]
    lib.gsr_preprocess_fwd.argtypes = [ci, ci, ci, vp, vp, cf, vp, vp, vp, vp, vp, vp, vp, vp, ci, ci, cf, cf, ci, ci,
                                       vp, vp, vp, vp, vp, vp, vp, vp, vp, vp]
    lib.gsr_mark_visible.argtypes = [ci, vp, vp, vp, vp]
    u32 = C.c_uint32
    lib.gsr_bin_sort.argtypes = [ci, vp, vp, vp, vp, C.c_size_t, vp]
    lib.gsr_bin_sort_status.argtypes = [ci, vp, C.c_size_t, vp]
    lib.gsr_debug_sort_spin_limit.argtypes = [u32]
    lib.gsr_debug_sort_spin_limit.restype = None
    lib.gsr_assume_exclusive_device.argtypes = [ci]
    lib.gsr_assume_exclusive_device.restype = None
    lib.gsr_bin_sort_error_word.argtypes = [vp]
    lib.gsr_bin_sort_error_word.restype = vp
    lib.gsr_bin_lists.argtypes = [ci, u32, u32, u32, u32, ci, ci, vp, vp, vp, vp, vp, vp, vp, vp, C.c_size_t, vp]
    lib.gsr_blend_fwd.argtypes = [ci, ci, vp, vp, vp, vp, vp, vp, vp, vp]
    lib.gsr_blend_bwd.argtypes = [ci, ci, vp, vp, vp, vp, vp, vp, vp, vp, vp]
    lib.gsr_blend_fwd_timed.argtypes = [ci, ci, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp]
    lib.gsr_blend_bwd_timed.argtypes = [ci, ci, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp]
    lib.gsr_blend_bwd_fill.argtypes = [ci, ci, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, C.c_size_t, vp, vp, vp]
    lib.gsr_blend_bwd_segments.argtypes = [ci, ci, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, C.c_size_t, vp, vp, vp, vp]
    lib.gsr_segments_bytes.restype = C.c_size_t
    lib.gsr_segments_bytes.argtypes = [ci, ci]
    lib.gsr_blend_fwd_segments.argtypes = [ci, ci, vp, vp, vp, vp, vp, vp, vp, vp, C.c_size_t, u32, vp, vp, vp]
    lib.gsr_debug_segments.argtypes, lib.gsr_debug_segments.restype = [u32, u32], None
    lib.gsr_event_create.restype = vp
    lib.gsr_event_create_untimed.restype = vp
    lib.gsr_event_destroy.argtypes, lib.gsr_event_destroy.restype = [vp], None
    lib.gsr_event_elapsed_ms.argtypes = [vp, vp, C.POINTER(C.c_float)]
    lib.gsr_preprocess_bwd.argtypes = [ci, ci, ci, vp, vp, cf, vp, vp, vp, vp, vp, vp, ci, ci, cf, cf, vp, vp,
                                       vp, vp, vp, vp, vp, vp, vp, vp, ci, vp, vp]
    lib.gsr_preprocess_fwd_raw.argtypes = [ci, ci, ci, vp, vp, cf, vp, vp, vp, vp, vp, vp, vp, ci, ci, cf, cf, ci,
                                           vp, vp, vp, vp, vp, vp, vp, vp, vp, vp]
    lib.gsr_forward_workspace_bytes.restype = C.c_size_t
    lib.gsr_forward_workspace_bytes.argtypes = [ci, ci, ci]
    lib.gsr_forward.argtypes = [C.POINTER(_ForwardArgs), _RESIZE_FN, vp, C.POINTER(_ForwardOut), vp]
    lib.gsr_preprocess_bwd_raw.argtypes = [ci, ci, ci, vp, vp, cf, vp, vp, vp, vp, vp, vp, vp, ci, ci, cf, cf,
                                           vp, vp, vp, vp, vp, vp, vp, vp, vp, ci, vp, vp]
    return lib


_lib = _load()


def assume_exclusive_device(on: bool = True) -> None:
    """Tells the library that this process has the GPU to itself (no other process, no second stream with
    kernels of its own): the depth sort may then skip its per-workgroup tickets when its whole grid is resident
    at once.  Off by default; GSR_EXCLUSIVE_DEVICE=1 in the environment sets it too."""
    _lib.gsr_assume_exclusive_device(int(bool(on)))


def set_segment_length(entries: int, pool: int = 0) -> None:
    """Entries per depth segment of the blend backward from now on, and the number of boundary records a forward
    may take (gsr_debug_segments in gsr.h): the library's own 2048 and 8192 suit real scenes; tests of the path on
    small scenes set them lower, 0 entries switches the segments off, pool 0 is the whole pool."""
    _lib.gsr_debug_segments(int(entries), int(pool))


SEGMENT_POOL = 8192        # dumps in a forward's pool (kSegmentCap in blend.hip)


def segment_counts(st) -> tuple:
    """(dumps the forward's blend asked for, entries per segment it worked with, dumps per tile) of a single-call
    forward that kept depth segments, read from the block gsr_forward returned; waits for the device.  The block's
    layout is blend.hip's SegmentsLayout: pool of SEGMENT_POOL dumps of 4096 bytes | queue of 8 bytes each |
    one word per tile | 256 bytes that start with the two counters, every piece padded to 256 bytes."""
    if st.segments is None:
        return 0, 0, None
    tiles = ((st.W + TILE - 1) // TILE) * ((st.H + TILE - 1) // TILE)
    pad = lambda n: (n + 255) // 256 * 256
    storage = (st.records if st.segments_block is None else st.segments_block).untyped_storage()
    at = st.segments - storage.data_ptr() + SEGMENT_POOL * 4096 + pad(SEGMENT_POOL * 8)
    words = torch.empty(0, dtype=torch.int32, device=st.records.device).set_(storage, at // 4, (pad(tiles * 4) // 4 + 3,))
    host = words.cpu()
    head = pad(tiles * 4) // 4
    return int(host[head]) & 0xFFFFFFFF, int(host[head + 1]), host[:tiles].clone()


class StageTimer:
    """Optional per-stage device timing: while one is installed with
    ``set_stage_timer``, every C-ABI call is bracketed by a pair of events on
    the current stream (the stream the kernels run on).  ``only`` restricts the brackets to the
    named stages: each bracket costs a few microseconds of device idle time, so a timed region
    that is itself being measured should carry as few as it can.  Stages named in ``exact`` (the two
    blend kernels can be) are not bracketed at all: their launch stamps a pair of events with the
    kernel's own start and end (gsr_blend_*_timed), the duration a profiler's kernel trace reports,
    at no cost to the stream; ``exact_ms`` returns those."""

    EXACT_STAGES = ("blend_fwd", "blend_bwd")

    def __init__(self, only=None, exact=()):
        self.events = {}
        self.only = None if only is None else frozenset(only)
        self.exact = frozenset(exact)
        if not self.exact <= frozenset(self.EXACT_STAGES):
            raise ValueError(f"kernel-exact timing exists for {self.EXACT_STAGES} only")
        self.kernel_events = {}

    def kernel_pair(self, name):
        pair = (_lib.gsr_event_create(), _lib.gsr_event_create())
        if not pair[0] or not pair[1]:
            raise RuntimeError("libgsr_hip: " + _lib.gsr_last_error().decode())
        self.kernel_events.setdefault(name, []).append(pair)
        return C.c_void_p(pair[0]), C.c_void_p(pair[1])

    def exact_ms(self):
        """Waits for the kernels, then {stage: (mean, min, max) milliseconds of the kernel itself, launches}."""
        out = {}
        for name, pairs in self.kernel_events.items():
            ms = []
            for a, b in pairs:
                v = C.c_float()
                if _lib.gsr_event_elapsed_ms(C.c_void_p(a), C.c_void_p(b), C.byref(v)) != 0:
                    raise RuntimeError("libgsr_hip: " + _lib.gsr_last_error().decode())
                ms.append(float(v.value))
            out[name] = (sum(ms) / len(ms), min(ms), max(ms), len(ms))
        return out

    def close(self):
        for pairs in self.kernel_events.values():
            for a, b in pairs:
                _lib.gsr_event_destroy(C.c_void_p(a))
                _lib.gsr_event_destroy(C.c_void_p(b))
        self.kernel_events = {}

    def bracket(self, name):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        self.events.setdefault(name, []).append((e0, e1))
        return e0, e1

    def mean_ms(self):
        """Synchronises, then {stage: mean milliseconds per call}."""
        torch.cuda.synchronize()
        return {k: sum(a.elapsed_time(b) for a, b in v) / len(v) for k, v in self.events.items()}

    def calls(self):
        return {k: len(v) for k, v in self.events.items()}


_timer: Optional[StageTimer] = None


def set_stage_timer(timer: Optional[StageTimer]) -> None:
    global _timer
    _timer = timer


def _run(what: str, fn, *args) -> None:
    if _timer is not None and (_timer.only is None or what in _timer.only):
        e0, e1 = _timer.bracket(what)
        e0.record()
        rc = fn(*args)
        e1.record()
    else:
        rc = fn(*args)
    if rc != 0:
        raise RuntimeError(f"libgsr_hip {what} failed ({rc}): {_lib.gsr_last_error().decode()}")


def _run_blend(what: str, fn, fn_timed, *args) -> None:
    """A blend launch; ``args`` without the trailing stream.  Kernel-exact events when the installed timer asks."""
    if _timer is not None and what in _timer.exact:
        start, stop = _timer.kernel_pair(what)
        rc = fn_timed(*args, start, stop, _stream())
        if rc != 0:
            raise RuntimeError(f"libgsr_hip {what} failed ({rc}): {_lib.gsr_last_error().decode()}")
    else:
        _run(what, fn, *args, _stream())


def _ptr(t: Optional[torch.Tensor]):
    return None if t is None else C.c_void_p(t.data_ptr())


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


# Host-side helpers a forward keeps between calls, one set per (host thread, device): two threads that
# render on one device (each on its own stream) never share a landing buffer, a side stream or a
# status word.  The library itself keeps no state.
_local = threading.local()
TOTALS_HEAD = 8            # words the host reads back: N, E1, E2, V, status, too many instances, 2 spare (gsr_totals_words)
STATUS_SORT, STATUS_TOTALS = 1, 2


def _per_device(dev: torch.device) -> dict:
    slots = getattr(_local, "slots", None)
    if slots is None:
        slots = _local.slots = {}
    key = dev.index if dev.index is not None else torch.cuda.current_device()
    d = slots.get(key)
    if d is None:
        d = slots[key] = {}
    return d


def _side_stream(dev: torch.device) -> torch.cuda.Stream:
    """A second stream (default priority) for work off the critical path."""
    d = _per_device(dev)
    if "side" not in d:
        d["side"] = torch.cuda.Stream(device=dev)
    return d["side"]


def _pinned_totals(dev: torch.device) -> torch.Tensor:
    """Page-locked landing buffer for the list sizes and the status word.  Every forward waits
    for its own copy before reading it, so the successive calls of one thread can share it."""
    d = _per_device(dev)
    if "pinned" not in d:
        d["pinned"] = torch.empty((TOTALS_HEAD,), dtype=torch.int32, pin_memory=True)
    return d["pinned"]


def _status_word(dev: torch.device) -> torch.Tensor:
    """The device-side status word of this thread's forwards on ``dev`` (see gsr_bin_lists in gsr.h): a
    list builder that refuses to run (its depth sort gave up, or it was handed stale list sizes) sets a
    bit here, and the next forward's read-back of the list sizes brings it to the host."""
    d = _per_device(dev)
    if "status" not in d:
        d["status"] = torch.zeros((4,), dtype=torch.int32, device=dev)
    return d["status"]


class RasterizerFault(RuntimeError):
    """A forward on this thread and device did not build its tile lists (status bits in ``.status``);
    its image was the background colour and its backward added nothing."""

    def __init__(self, status: int):
        why = []
        if status & STATUS_SORT:
            why.append("its depth sort gave up waiting (GSR_E_SORT)")
        if status & STATUS_TOTALS:
            why.append("it was handed list sizes that were not its own")
        super().__init__("libgsr_hip: an earlier forward built no tile lists: " + "; ".join(why or [hex(status)]))
        self.status = status


def _f32(t: Optional[torch.Tensor], name: str) -> Optional[torch.Tensor]:
    if t is None or t.numel() == 0:
        return None
    if t.dtype != torch.float32:
        raise TypeError(f"{name} must be float32, got {t.dtype}")
    if not t.is_cuda:
        raise ValueError(f"{name} must be on the GPU")
    return t.contiguous()


class GaussianRasterizationSettings(NamedTuple):
    image_height: int
    image_width: int
    tanfovx: float
    tanfovy: float
    bg: torch.Tensor
    scale_modifier: float
    viewmatrix: torch.Tensor
    projmatrix: torch.Tensor
    sh_degree: int
    campos: torch.Tensor
    prefiltered: bool
    debug: bool


class ForwardState:
    """Everything the forward leaves behind: op outputs, what backward needs,
    and the intermediate buffers the parity tests inspect."""
    __slots__ = ("P", "M", "D", "W", "H", "records", "radii", "depth_keys", "sorted_vals",
                 "num_rendered", "num_visible", "dsplat", "zeroed_grads", "visible", "point_list", "ranges", "tile_ids_sorted", "color", "final_T", "n_contrib",
                 "view", "proj", "campos", "bg", "segments", "segments_block")

    # views into the 64-byte records, named as the oracle names them
    @property
    def means2D(self):
        return self.records.view(torch.float32)[:, 0:2]

    @property
    def conic_opacity(self):
        return self.records.view(torch.float32)[:, 4:8]

    @property
    def rgb(self):
        return self.records.view(torch.float32)[:, 8:11]

    @property
    def depths(self):
        return self.records.view(torch.float32)[:, 12]

    @property
    def rects(self):
        r = self.records[:, 13:15]
        return torch.stack([r[:, 0] & 0xFFFF, r[:, 0] >> 16, r[:, 1] & 0xFFFF, r[:, 1] >> 16], dim=1)

    @property
    def tiles_touched(self):
        """Tiles in each Gaussian's rectangle (0 for culled ones), from the records."""
        r = self.rects.to(torch.int64)
        n = (r[:, 2] - r[:, 0]) * (r[:, 3] - r[:, 1])
        return torch.where(self.radii > 0, n, torch.zeros_like(n)).to(torch.int32)

    @property
    def clamped(self):
        c = self.records[:, 11]
        return torch.stack([c & 1, (c >> 1) & 1, (c >> 2) & 1], dim=1).to(torch.uint8)

    def sort_keys(self):
        """The 64-bit (tile << 32 | depth bits) key of every sorted instance."""
        d = self.depth_keys.to(torch.int64)[self.point_list.to(torch.int64)]
        return (self.tile_ids_sorted.to(torch.int64) << 32) | d


def _read_back_event(dev: torch.device):
    """The event a single-call forward marks its read-back with, one per (host thread, device)."""
    d = _per_device(dev)
    if "event" not in d:
        d["event"] = _lib.gsr_event_create_untimed()
        if not d["event"]:
            raise RuntimeError("libgsr_hip: " + _lib.gsr_last_error().decode())
    return d["event"]


def _gradient_shapes(P: int, raw: bool, n_rest: int = 0, M: int = 0, colors: bool = False, cov3D: bool = False) -> dict:
    """Name -> shape of the gradient tensors gsr_preprocess_bwd[_raw] writes for this combination of inputs."""
    if raw:
        shapes = dict(xyz=(P, 3), means2D=(P, 3), opacity=(P, 1), features_dc=(P, 1, 3), scaling=(P, 3), rotation=(P, 4))
        if n_rest:
            shapes["features_rest"] = (P, n_rest, 3)
        return shapes
    shapes = dict(means3D=(P, 3), means2D=(P, 3), opacities=(P, 1))
    shapes["colors_precomp" if colors else "shs"] = (P, 3) if colors else (P, M, 3)
    if cov3D:
        shapes["cov3D_precomp"] = (P, 6)
    else:
        shapes["scales"], shapes["rotations"] = (P, 3), (P, 4)
    return shapes


class _GradBlock:
    """Gradient tensors carved from one allocation made in the forward: `views` (name -> tensor, shapes as in `shapes`)
    lie in the `nbytes` bytes at address `at`, which the blend backward clears (`cleared`) before the per-Gaussian
    backward may take them as zeroed.  `touched` (P bytes, cleared by the forward): the blend backward marks the Gaussians
    it may have sums for, the per-Gaussian backward reads the rows of those only."""
    __slots__ = ("shapes", "views", "touched", "at", "nbytes", "cleared")

    def __init__(self, shapes, views, touched, at, nbytes):
        self.shapes, self.views, self.touched, self.at, self.nbytes = shapes, views, touched, at, nbytes
        self.cleared = False


_carve_plans: dict = {}


def _carve_zero_block(P: int, shapes: dict, dev):
    """Everything the backward wants zeroed before it starts, as TWO allocations: the (P,16) accumulator of the
    blend backward with the flags of the Gaussians it touches, and the gradient tensors of the per-Gaussian
    backward (which then writes only the rows of Gaussians that have gradients).  Two, because autograd installs the
    gradient views as ``param.grad``: in one block with them the accumulator (64 bytes per Gaussian) would stay
    allocated through the optimiser step.  Returns (accumulator block, accumulator, flags, gradient block,
    {name: tensor}); the pieces start on 256-byte boundaries.  One as_strided per piece, from a plan kept per
    (P, shapes): this runs every step, and at a hundred thousand Gaussians the step is short enough for a dozen
    extra tensor operations to show."""
    key = (P, tuple(shapes.items()))
    plan = _carve_plans.get(key)
    if plan is None:
        def pad(n):
            return (n + 63) // 64 * 64

        def strides(shape):
            out, run = [], 1
            for d in reversed(shape):
                out.append(run)
                run *= d
            return tuple(reversed(out))
        flags_at = pad(P * GRAD_WORDS)          # P bytes of "touched" flags behind the accumulator
        first_total = flags_at + pad((P + 3) // 4)
        at, pieces = 0, []
        for k, v in shapes.items():
            pieces.append((k, tuple(v), strides(v), at))
            at += pad(int(torch.Size(v).numel()))
        if len(_carve_plans) > 64:
            _carve_plans.clear()
        plan = _carve_plans[key] = (first_total, at, pieces, flags_at)
    first_total, second_total, pieces, flags_at = plan
    first = torch.empty((first_total,), dtype=torch.float32, device=dev)
    second = torch.empty((max(second_total, 1),), dtype=torch.float32, device=dev)
    views = {k: torch.as_strided(second, shape, stride, off) for k, shape, stride, off in pieces}
    touched = first[flags_at:flags_at + (P + 3) // 4].view(torch.uint8)
    return first, torch.as_strided(first, (P, GRAD_WORDS), (GRAD_WORDS, 1), 0), touched, second, views


def _backward_block(st: ForwardState, raw, shs, cov3D_precomp, dev):
    """The accumulator of the blend backward (st.dsplat) with the flags of the Gaussians it touches, which the
    forward clears, and the gradient tensors of the per-Gaussian backward (st.zeroed_grads), which the blend backward
    clears; the per-Gaussian backward then writes only the rows of Gaussians that have gradients.  Returns the
    accumulator block's (address, bytes)."""
    if raw is not None:
        rest = raw["features_rest"]
        shapes = _gradient_shapes(st.P, True, n_rest=0 if rest is None else rest.shape[1])
    else:
        shapes = _gradient_shapes(st.P, False, M=st.M, colors=shs is None, cov3D=cov3D_precomp is not None)
    first, st.dsplat, touched, second, views = _carve_zero_block(st.P, shapes, dev)
    st.zeroed_grads = _GradBlock(shapes, views, touched, second.data_ptr(), second.numel() * 4)
    return first.data_ptr(), first.numel() * 4


def _thread_callbacks():
    """The resize callback and the two hooks gsr_forward is given, one set per host thread for the life of
    the thread.  They forward to whatever the current call has put into their slots.  A ctypes callback
    object made per call would sit in a reference cycle, and the tensors its function holds (the tile lists:
    a hundred megabytes a call) would stay allocated until Python's cyclic collector happened to run."""
    cbs = getattr(_local, "callbacks", None)
    if cbs is None:
        slot = {"resize": None, "after_preprocess": None, "before_blend": None}
        cbs = _local.callbacks = {
            "slot": slot,
            "resize": _RESIZE_FN(lambda _ctx, bytes_: slot["resize"](bytes_)),
            "after_preprocess": _HOOK_FN(lambda _ctx: slot["after_preprocess"]()),
            "before_blend": _HOOK_FN(lambda _ctx: slot["before_blend"]()),
        }
    return cbs


def _forward_single_call(s, st, dev, means3D, opacities, shs, colors_precomp, scales, rotations, cov3D_precomp,
                         keep_tile_ids, raw, prepare_backward, ref_lists) -> None:
    """gsr_forward: fills ``st`` like the staged path does.  Everything but the user-visible outputs lives in
    one workspace tensor; the tile lists come from the resize callback."""
    P, W, H = st.P, st.W, st.H
    u8 = dict(dtype=torch.uint8, device=dev)
    st.radii = torch.empty((P,), dtype=torch.int32, device=dev)
    st.visible = torch.empty((P,), dtype=torch.bool, device=dev)
    st.color = torch.empty((3, H, W), dtype=torch.float32, device=dev)
    nbytes = _lib.gsr_forward_workspace_bytes(P, W, H)
    ws = torch.empty((nbytes,), **u8)
    keep = {}

    def resize(bytes_):
        keep["lists"] = torch.empty((bytes_,), **u8)
        return keep["lists"].data_ptr()

    hooks = None
    if prepare_backward and _OVERLAP and P > 0:
        # The accumulator of the blend backward is cleared on a second stream beside the depth sort and the
        # list builder (short dependent launches that leave HBM idle): the library calls back once the
        # preprocess is enqueued, and again before it enqueues the blend, which then waits for the fill.
        main, side = torch.cuda.current_stream(dev), _side_stream(dev)
        st.dsplat = torch.empty((P, GRAD_WORDS), dtype=torch.float32, device=dev)
        cleared = torch.cuda.Event()

        def start_fill():
            handed_out = torch.cuda.Event()
            handed_out.record(main)
            side.wait_event(handed_out)
            with torch.cuda.stream(side):
                st.dsplat.zero_()
                cleared.record(side)

        hooks = (start_fill, lambda: main.wait_event(cleared))

    a = _ForwardArgs()
    a.P, a.D, a.W, a.H = P, st.D, W, H
    p = lambda t: None if t is None else t.data_ptr()
    if prepare_backward and hooks is None and P > 0 and _CARRIED_FILLS:
        a.zero_fill, a.zero_fill_bytes = _backward_block(st, raw, shs, cov3D_precomp, dev)
    if raw is not None:
        r = {k: _f32(v, k) for k, v in raw.items()}
        n_rest = 0 if r["features_rest"] is None else r["features_rest"].shape[1]
        st.M = 1 + n_rest
        a.raw, a.M = 1, st.M
        a.scales, a.rotations, a.opacities = p(r["scaling"]), p(r["rotation"]), p(r["opacity"])
        a.shs, a.shs_rest = p(r["features_dc"]), p(r["features_rest"])
        held = r
    else:
        a.raw, a.M = 0, st.M
        a.scales, a.rotations, a.opacities = p(scales), p(rotations), p(opacities)
        a.shs, a.colors_precomp, a.cov3D_precomp = p(shs), p(colors_precomp), p(cov3D_precomp)
        held = None
    a.means3D = p(means3D)
    a.viewmatrix, a.projmatrix, a.campos, a.bg = p(st.view), p(st.proj), p(st.campos), p(st.bg)
    a.scale_modifier, a.tanfovx, a.tanfovy = float(s.scale_modifier), float(s.tanfovx), float(s.tanfovy)
    a.prefiltered, a.reference_lists, a.debug, a.keep_tile_ids = int(bool(s.prefiltered)), ref_lists, int(bool(s.debug)), int(keep_tile_ids)
    a.out_color, a.radii, a.visible = p(st.color), p(st.radii), p(st.visible)
    a.workspace, a.workspace_bytes = ws.data_ptr(), nbytes
    status = _status_word(dev)
    a.h_totals, a.event, a.status = _pinned_totals(dev).data_ptr(), _read_back_event(dev), status.data_ptr()
    cbs = _thread_callbacks()
    slot = cbs["slot"]
    slot["resize"] = resize
    if hooks is not None:
        slot["after_preprocess"], slot["before_blend"] = hooks
        a.after_preprocess = C.cast(cbs["after_preprocess"], C.c_void_p)
        a.before_blend = C.cast(cbs["before_blend"], C.c_void_p)
    if _timer is not None and "blend_fwd" in _timer.exact:
        a.blend_start_event, a.blend_stop_event = _timer.kernel_pair("blend_fwd")
    out = _ForwardOut()
    try:
        rc = _lib.gsr_forward(C.byref(a), cbs["resize"], None, C.byref(out), _stream())
    finally:
        slot["resize"] = slot["after_preprocess"] = slot["before_blend"] = None
    del held
    if rc == GSR_E_EARLIER:
        status.zero_()
        raise RasterizerFault(int(out.carried_status))
    if rc != 0:
        raise RuntimeError(f"libgsr_hip forward failed ({rc}): {_lib.gsr_last_error().decode()}")
    N = int(out.N)
    st.num_rendered, st.num_visible = N, int(out.V)
    base, lists = ws.data_ptr(), keep["lists"]

    width = {torch.int32: 4, torch.float32: 4, torch.int64: 8}

    def view(buf, origin, ptr, count, dtype, shape):
        off = ptr - origin
        return buf[off:off + count * width[dtype]].view(dtype).view(shape)

    tiles = ((W + TILE - 1) // TILE) * ((H + TILE - 1) // TILE)
    st.records = view(ws, base, out.records, P * REC_WORDS, torch.int32, (P, REC_WORDS))
    st.depth_keys = view(ws, base, out.depth_keys, P, torch.int32, (P,))
    st.sorted_vals = view(ws, base, out.sorted_vals, P, torch.int64, (P,))
    st.ranges = view(ws, base, out.ranges, tiles * 2, torch.int32, (tiles, 2))
    st.final_T = view(ws, base, out.final_T, H * W, torch.float32, (H, W))
    st.n_contrib = view(ws, base, out.n_contrib, H * W, torch.int32, (H, W))
    st.point_list = view(lists, lists.data_ptr(), out.point_list, N, torch.int32, (N,))
    st.tile_ids_sorted = (view(lists, lists.data_ptr(), out.tile_ids_sorted, N, torch.int32, (N,))
                          if keep_tile_ids else None)
    st.segments = out.segments            # inside ws, which the views above keep alive


def rasterize_forward(s: GaussianRasterizationSettings, means3D, opacities, shs=None, colors_precomp=None,
                      scales=None, rotations=None, cov3D_precomp=None, keep_tile_ids: bool = False,
                      raw: Optional[dict] = None, prepare_backward: bool = False,
                      reference_lists: Optional[bool] = None, single_call: Optional[bool] = None) -> ForwardState:
    """preprocess -> bin -> blend on the current stream; one host read-back
    (the instance count) in the middle, as in the module it replaces.
    With ``raw`` (features_dc, features_rest, scaling, rotation, opacity: the model's raw
    parameters) the activations run inside the preprocess kernel and the activated
    arguments are ignored.  ``prepare_backward`` also clears the accumulator of the blend backward
    during the forward (off the critical path).  ``reference_lists`` (default: the module-level
    REFERENCE_LISTS) asks for the replaced module's own tile lists instead of the shorter ones.
    ``single_call`` (default: the module-level SINGLE_CALL) runs the forward as one gsr_forward call
    instead of the sequence of stage calls; the results are the same bit for bit."""
    means3D = _f32(means3D, "means3D")
    dev = means3D.device if means3D is not None else s.bg.device
    st = ForwardState()
    P = 0 if means3D is None else means3D.shape[0]
    W, H = int(s.image_width), int(s.image_height)
    st.P, st.W, st.H, st.D = P, W, H, int(s.sh_degree)
    opacities, shs, colors_precomp = _f32(opacities, "opacities"), _f32(shs, "shs"), _f32(colors_precomp, "colors_precomp")
    scales, rotations, cov3D_precomp = _f32(scales, "scales"), _f32(rotations, "rotations"), _f32(cov3D_precomp, "cov3D_precomp")
    st.M = 0 if shs is None else shs.shape[1]
    st.view, st.proj = _f32(s.viewmatrix, "viewmatrix"), _f32(s.projmatrix, "projmatrix")
    st.campos, st.bg = _f32(s.campos, "campos"), _f32(s.bg, "bg")
    st.dsplat = None
    st.zeroed_grads = None
    st.segments = st.segments_block = None
    ref_lists = int(REFERENCE_LISTS if reference_lists is None else reference_lists)
    # a timer that brackets stages with events needs the stage calls; one that only wants the blend kernels' own
    # events does not
    if (SINGLE_CALL if single_call is None else single_call) and (_timer is None or _timer.only == frozenset()):
        _forward_single_call(s, st, dev, means3D, opacities, shs, colors_precomp, scales, rotations, cov3D_precomp,
                             keep_tile_ids, raw, prepare_backward, ref_lists)
        return st
    stream = _stream()
    u32 = dict(dtype=torch.int32, device=dev)   # uint32 payloads carried in int32 tensors
    st.records = torch.empty((P, REC_WORDS), **u32)
    st.radii = torch.empty((P,), **u32)
    st.depth_keys = torch.empty((P,), **u32)
    st.sorted_vals = torch.empty((P,), dtype=torch.int64, device=dev)   # packed (index, rectangle), front to back
    st.visible = torch.empty((P,), dtype=torch.bool, device=dev)      # radii > 0, written by the preprocess kernel
    gx, gy = (W + TILE - 1) // TILE, (H + TILE - 1) // TILE
    st.ranges = torch.empty((gx * gy, 2), **u32)
    st.color = torch.empty((3, H, W), dtype=torch.float32, device=dev)
    st.final_T = torch.empty((H, W), dtype=torch.float32, device=dev)
    st.n_contrib = torch.empty((H, W), **u32)
    st.tile_ids_sorted = None

    N = E1 = E2 = V = 0
    if P > 0:
        totals = torch.empty((_lib.gsr_totals_words(P),), **u32)
        sort_vals = torch.empty((P,), dtype=torch.int64, device=dev)
        status = _status_word(dev)
        outs = (_ptr(st.records), _ptr(st.radii), None, _ptr(st.depth_keys), None,
                _ptr(sort_vals), _ptr(totals), _ptr(status), _ptr(st.visible), stream)
        if raw is not None:
            r = {k: _f32(v, k) for k, v in raw.items()}
            n_rest = 0 if r["features_rest"] is None else r["features_rest"].shape[1]
            st.M = 1 + n_rest
            _run("preprocess_fwd", _lib.gsr_preprocess_fwd_raw, P, st.D, n_rest, _ptr(means3D), _ptr(r["scaling"]),
                 float(s.scale_modifier), _ptr(r["rotation"]), _ptr(r["opacity"]), _ptr(r["features_dc"]),
                 _ptr(r["features_rest"]), _ptr(st.view), _ptr(st.proj), _ptr(st.campos), W, H, float(s.tanfovx),
                 float(s.tanfovy), ref_lists, *outs)
        else:
            _run("preprocess_fwd", _lib.gsr_preprocess_fwd, P, st.D, st.M, _ptr(means3D), _ptr(scales),
                 float(s.scale_modifier), _ptr(rotations), _ptr(opacities), _ptr(shs), _ptr(colors_precomp),
                 _ptr(cov3D_precomp), _ptr(st.view), _ptr(st.proj), _ptr(st.campos), W, H, float(s.tanfovx),
                 float(s.tanfovy), int(bool(s.prefiltered)), ref_lists, *outs)

        # The blend backward accumulates into a zeroed (P,16) buffer.  The depth sort and the
        # partitions are short dependent launches that leave HBM idle: the fill runs beside them on
        # a second stream.  The buffer itself comes from the caller's stream like every other tensor
        # here (a block handed out on the side stream and marked as used by this one would make the
        # caching allocator poll events, and fall back to hipMalloc whenever a long backward has not
        # finished by the next forward); the two events order the fill after the block's previous use
        # and before its next one.
        if prepare_backward and _OVERLAP:
            main, side = torch.cuda.current_stream(dev), _side_stream(dev)
            st.dsplat = torch.empty((P, GRAD_WORDS), dtype=torch.float32, device=dev)
            handed_out = torch.cuda.Event()
            handed_out.record(main)
            side.wait_event(handed_out)
            with torch.cuda.stream(side):
                st.dsplat.zero_()
                cleared = torch.cuda.Event()
                cleared.record(side)
        elif prepare_backward and _CARRIED_FILLS:
            _backward_block(st, raw, shs, cov3D_precomp, dev)
            st.dsplat.zero_()                                 # the single call has the blend kernel do this
            st.zeroed_grads.touched.zero_()

        # The list sizes go to the host (the pipeline's one read-back) while the GPU sorts by depth:
        # the copy is queued before the sort and the host waits for the copy alone.
        host = _pinned_totals(dev)
        host.copy_(totals[:TOTALS_HEAD], non_blocking=True)
        copied = torch.cuda.Event()
        copied.record()
        nb = _lib.gsr_bin_sort_bytes(P)
        sort_scratch = torch.empty((nb,), dtype=torch.uint8, device=dev)
        _run("bin_sort", _lib.gsr_bin_sort, P, _ptr(st.depth_keys), _ptr(sort_vals), _ptr(st.sorted_vals),
             _ptr(sort_scratch), nb, stream)
        sort_error = C.c_void_p(_lib.gsr_bin_sort_error_word(_ptr(sort_scratch)))
        if s.debug:
            # a synchronising check, as the replaced module synchronises after every stage in this mode
            _run("bin_sort_status", _lib.gsr_bin_sort_status, P, _ptr(sort_scratch), nb, stream)
        copied.synchronize()
        N, E1, E2, V, carried, too_many = (int(v) & 0xFFFFFFFF for v in host.tolist()[:6])
        if too_many:
            # the sizes above have wrapped; nothing has been sized by them yet (gsr_forward refuses the same way)
            raise RuntimeError(f"libgsr_hip forward failed ({GSR_E_TOO_LARGE}): more than {MAX_INSTANCES} tile instances in "
                               f"one frame ({P} Gaussians on {W}x{H} pixels): the lists' sizes are 32-bit")
        if carried:
            # an earlier forward of this thread refused to build its lists; this one has not been hurt
            # (it is abandoned here, before its own lists are built)
            status.zero_()
            raise RasterizerFault(carried)
    st.num_rendered, st.num_visible = N, V
    st.point_list = torch.empty((max(N, 1),), **u32)
    if keep_tile_ids:
        st.tile_ids_sorted = torch.empty((max(N, 1),), **u32)
    nb = _lib.gsr_bin_lists_bytes(V, E1, E2, W, H)
    scratch = torch.empty((nb,), dtype=torch.uint8, device=dev)
    guards = (sort_error, _ptr(totals), _ptr(status)) if P > 0 else (None, None, None)
    _run("bin_lists", _lib.gsr_bin_lists, P, V, N, E1, E2, W, H, _ptr(st.sorted_vals), _ptr(st.point_list),
         _ptr(st.ranges), _ptr(st.tile_ids_sorted), *guards, _ptr(scratch), nb, stream)
    if P > 0 and prepare_backward and _OVERLAP:
        torch.cuda.current_stream(dev).wait_event(cleared)
    blend_args = (W, H, _ptr(st.ranges), _ptr(st.point_list), _ptr(st.records), _ptr(st.bg), _ptr(st.color), _ptr(st.final_T),
                  _ptr(st.n_contrib))
    if prepare_backward and N > 0 and _SEGMENTS:
        # depth segments for the backward's long walks (gsr_blend_bwd_segments); the block's last 256 bytes hold its counters
        nb = _lib.gsr_segments_bytes(W, H)
        st.segments_block = torch.empty((nb,), dtype=torch.uint8, device=dev)
        st.segments_block[nb - 256:].zero_()
        st.segments = st.segments_block.data_ptr()
        seg = (_ptr(st.segments_block), nb, N)
        _run_blend("blend_fwd", lambda *a: _lib.gsr_blend_fwd_segments(*a[:-1], None, None, a[-1]), _lib.gsr_blend_fwd_segments,
                   *blend_args, *seg)
    else:
        _run_blend("blend_fwd", _lib.gsr_blend_fwd, _lib.gsr_blend_fwd_timed, *blend_args)
    st.point_list = st.point_list[:N]
    if keep_tile_ids:
        st.tile_ids_sorted = st.tile_ids_sorted[:N]
    return st


def blend_backward(st: ForwardState, grad_color: torch.Tensor) -> torch.Tensor:
    """dL/dpixels (3,H,W) -> per-Gaussian sums (P,16); see GSR_GRAD_WORDS in gsr.h."""
    grad_color = _f32(grad_color, "grad_color")
    dsplat, st.dsplat = st.dsplat, None                     # cleared during the forward, good for one backward
    if dsplat is None:
        dsplat = torch.zeros((st.P, GRAD_WORDS), dtype=torch.float32, device=grad_color.device)
    if st.P > 0 and st.num_rendered > 0:
        # the gradient tensors that came with the accumulator are cleared by this kernel's workgroups
        held = st.zeroed_grads
        fill = (None, None, 0) if held is None or held.cleared else (_ptr(held.touched), C.c_void_p(held.at), held.nbytes)
        args = (st.W, st.H, _ptr(st.ranges), _ptr(st.point_list), _ptr(st.records), _ptr(st.bg), _ptr(st.final_T),
                _ptr(st.n_contrib), _ptr(grad_color), _ptr(dsplat), *fill, C.c_void_p(st.segments))
        _run_blend("blend_bwd", lambda *a: _lib.gsr_blend_bwd_segments(*a[:-1], None, None, a[-1]), _lib.gsr_blend_bwd_segments, *args)
        if held is not None:
            held.cleared = True
    return dsplat


def splat_sums_to_grads(st: ForwardState, dsplat: torch.Tensor) -> dict:
    """The linear map gsr_preprocess_bwd applies to the (P,16) moment sums of the blend
    backward, in PyTorch: screen-space gradients per Gaussian, for inspection and tests."""
    co = st.conic_opacity
    A, B, Cc, o = co[:, 0], co[:, 1], co[:, 2], co[:, 3]
    m = dsplat
    vis = st.radii > 0
    z = torch.zeros_like(A)
    mean2D = torch.stack([-(A * m[:, 1] + B * m[:, 2]) * (0.5 * st.W), -(Cc * m[:, 2] + B * m[:, 1]) * (0.5 * st.H)], 1)
    conic = torch.stack([-0.5 * m[:, 3], -m[:, 4], -0.5 * m[:, 5]], 1)
    return dict(mean2D=torch.where(vis[:, None], mean2D, z[:, None]), conic=torch.where(vis[:, None], conic, z[:, None]),
                opacity=torch.where(vis & (o > 0), m[:, 0] / o, z), color=torch.where(vis[:, None], m[:, 6:9], z[:, None]))


def rasterize_backward(s: GaussianRasterizationSettings, st: ForwardState, grad_color, means3D, shs=None,
                       scales=None, rotations=None, cov3D_precomp=None, dsplat=None):
    """Gradients of every differentiable input, as a dict."""
    means3D = _f32(means3D, "means3D")
    shs, scales = _f32(shs, "shs"), _f32(scales, "scales")
    rotations, cov3D_precomp = _f32(rotations, "rotations"), _f32(cov3D_precomp, "cov3D_precomp")
    dev = grad_color.device
    P = st.P
    if dsplat is None:
        dsplat = blend_backward(st, grad_color)
    shapes = _gradient_shapes(P, False, M=st.M, colors=shs is None, cov3D=cov3D_precomp is not None)
    g = dict(shs=None, colors_precomp=None, scales=None, rotations=None, cov3D_precomp=None)
    held = _take_zeroed_grads(st, shapes)
    g.update(held.views if held is not None else {k: torch.empty(v, dtype=torch.float32, device=dev) for k, v in shapes.items()})
    if P > 0:
        _run("preprocess_bwd", _lib.gsr_preprocess_bwd, P, st.D, st.M, _ptr(means3D), _ptr(scales), float(s.scale_modifier), _ptr(rotations), _ptr(shs),
            _ptr(cov3D_precomp), _ptr(st.view), _ptr(st.proj), _ptr(st.campos), st.W, st.H, float(s.tanfovx),
            float(s.tanfovy), _ptr(st.records), _ptr(dsplat), _ptr(g["means3D"]), _ptr(g["means2D"]),
            _ptr(g["opacities"]), _ptr(g["shs"]), _ptr(g["colors_precomp"]), _ptr(g["scales"]), _ptr(g["rotations"]),
            _ptr(g["cov3D_precomp"]), int(held is not None), None if held is None else _ptr(held.touched), _stream())
    return g


def _take_zeroed_grads(st: ForwardState, shapes: dict):
    """The gradient tensors the forward had cleared, if they fit this backward (same tensors, same shapes) and
    have not been handed out yet; good for one backward, like the accumulator."""
    held, st.zeroed_grads = st.zeroed_grads, None
    if held is None or not held.cleared or held.shapes != shapes:
        return None
    return held


def rasterize_backward_raw(s: GaussianRasterizationSettings, st: ForwardState, grad_color, xyz, raw: dict,
                           dsplat=None):
    """Gradients w.r.t. the model's raw parameters (activations' chain rule in-kernel)."""
    xyz = _f32(xyz, "xyz")
    r = {k: _f32(v, k) for k, v in raw.items()}
    dev = grad_color.device
    P = st.P
    if dsplat is None:
        dsplat = blend_backward(st, grad_color)
    n_rest = 0 if r["features_rest"] is None else r["features_rest"].shape[1]
    shapes = _gradient_shapes(P, True, n_rest=n_rest)
    held = _take_zeroed_grads(st, shapes)
    g = dict(features_rest=None)
    g.update(held.views if held is not None else {k: torch.empty(v, dtype=torch.float32, device=dev) for k, v in shapes.items()})
    if P > 0:
        _run("preprocess_bwd", _lib.gsr_preprocess_bwd_raw, P, st.D, n_rest, _ptr(xyz), _ptr(r["scaling"]),
             float(s.scale_modifier), _ptr(r["rotation"]), _ptr(r["opacity"]), _ptr(r["features_dc"]),
             _ptr(r["features_rest"]), _ptr(st.view), _ptr(st.proj), _ptr(st.campos), st.W, st.H, float(s.tanfovx),
             float(s.tanfovy), _ptr(st.records), _ptr(dsplat), _ptr(g["xyz"]), _ptr(g["means2D"]), _ptr(g["opacity"]),
             _ptr(g["features_dc"]), _ptr(g["features_rest"]), _ptr(g["scaling"]), _ptr(g["rotation"]),
             int(held is not None), None if held is None else _ptr(held.touched), _stream())
    return g


def _detach_image(ctx, st: ForwardState) -> torch.Tensor:
    """Takes the image out of the forward state that the autograd node keeps.  The returned tensor gets that
    node as its grad_fn; left on the state it would close a reference cycle (node -> state -> image -> node),
    and every step's lists and records (hundreds of megabytes) would live until Python's cyclic collector
    runs, which then stalls the training loop for tens of milliseconds."""
    color, st.color = st.color, None
    ctx.image_shape = tuple(color.shape)
    return color


class _RasterizeGaussiansRaw(torch.autograd.Function):
    """Same operator on the model's raw parameters: exp / sigmoid / normalize / dc|rest
    concatenation and their backward run inside the per-Gaussian kernels."""

    @staticmethod
    def forward(ctx, xyz, means2D, features_dc, features_rest, opacity_raw, scaling_raw, rotation_raw,
                raster_settings):
        raw = dict(features_dc=features_dc, features_rest=_empty_to_none(features_rest), scaling=scaling_raw,
                   rotation=rotation_raw, opacity=opacity_raw)
        st = rasterize_forward(raster_settings, xyz, None, raw=raw, prepare_backward=any(ctx.needs_input_grad))
        if raster_settings.debug:
            torch.cuda.synchronize()
        ctx.raster_settings = raster_settings
        ctx.state = st
        ctx.save_for_backward(xyz, features_dc, features_rest, opacity_raw, scaling_raw, rotation_raw)
        ctx.mark_non_differentiable(st.radii, st.visible)
        ctx.set_materialize_grads(False)        # no zero tensors for the integer outputs' "gradients"
        return _detach_image(ctx, st), st.radii, st.visible

    @staticmethod
    def backward(ctx, grad_out_color, _grad_radii, _grad_visible):
        xyz, features_dc, features_rest, opacity_raw, scaling_raw, rotation_raw = ctx.saved_tensors
        if grad_out_color is None:
            grad_out_color = torch.zeros(ctx.image_shape, dtype=torch.float32, device=xyz.device)
        raw = dict(features_dc=features_dc, features_rest=_empty_to_none(features_rest), scaling=scaling_raw,
                   rotation=rotation_raw, opacity=opacity_raw)
        g = rasterize_backward_raw(ctx.raster_settings, ctx.state, grad_out_color, xyz, raw)
        if ctx.raster_settings.debug:
            torch.cuda.synchronize()
        return (g["xyz"], g["means2D"], g["features_dc"], g["features_rest"], g["opacity"], g["scaling"],
                g["rotation"], None)


def rasterize_gaussians_raw(xyz, means2D, features_dc, features_rest, opacity_raw, scaling_raw, rotation_raw,
                            raster_settings):
    """(color, radii, radii > 0) from raw model parameters; an extension, not part of the reference's API."""
    if features_rest is None:
        features_rest = torch.Tensor([])
    return _RasterizeGaussiansRaw.apply(xyz, means2D, features_dc, features_rest, opacity_raw, scaling_raw,
                                        rotation_raw, raster_settings)


def _empty_to_none(t):
    return None if t is None or t.numel() == 0 else t


class _RasterizeGaussians(torch.autograd.Function):
    @staticmethod
    def forward(ctx, means3D, means2D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp,
                raster_settings):
        s = raster_settings
        args = dict(shs=_empty_to_none(sh), colors_precomp=_empty_to_none(colors_precomp),
                    scales=_empty_to_none(scales), rotations=_empty_to_none(rotations),
                    cov3D_precomp=_empty_to_none(cov3Ds_precomp))
        if s.debug:
            try:
                st = rasterize_forward(s, means3D, opacities, prepare_backward=any(ctx.needs_input_grad), **args)
                torch.cuda.synchronize()
            except Exception:
                torch.save(dict(settings=tuple(s), means3D=means3D, opacities=opacities, **args), "snapshot_fw.dump")
                print("\nAn error occured in forward. Please forward snapshot_fw.dump for debugging.")
                raise
        else:
            st = rasterize_forward(s, means3D, opacities, prepare_backward=any(ctx.needs_input_grad), **args)
        ctx.raster_settings = s
        ctx.state = st
        ctx.save_for_backward(means3D, sh, scales, rotations, cov3Ds_precomp)
        ctx.mark_non_differentiable(st.radii)
        ctx.set_materialize_grads(False)        # no zero tensor for the radii's "gradient"
        return _detach_image(ctx, st), st.radii

    @staticmethod
    def backward(ctx, grad_out_color, _grad_radii):
        s, st = ctx.raster_settings, ctx.state
        means3D, sh, scales, rotations, cov3Ds_precomp = ctx.saved_tensors
        if grad_out_color is None:
            grad_out_color = torch.zeros(ctx.image_shape, dtype=torch.float32, device=means3D.device)
        kw = dict(shs=_empty_to_none(sh), scales=_empty_to_none(scales), rotations=_empty_to_none(rotations),
                  cov3D_precomp=_empty_to_none(cov3Ds_precomp))
        if s.debug:
            try:
                g = rasterize_backward(s, st, grad_out_color, means3D, **kw)
                torch.cuda.synchronize()
            except Exception:
                torch.save(dict(settings=tuple(s), means3D=means3D, grad=grad_out_color, **kw), "snapshot_bw.dump")
                print("\nAn error occured in backward. Writing snapshot_bw.dump for debugging.\n")
                raise
        else:
            g = rasterize_backward(s, st, grad_out_color, means3D, **kw)
        return (g["means3D"], g["means2D"], g["shs"], g["colors_precomp"], g["opacities"], g["scales"],
                g["rotations"], g["cov3D_precomp"], None)


def rasterize_gaussians(means3D, means2D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp,
                        raster_settings):
    return _RasterizeGaussians.apply(means3D, means2D, sh, colors_precomp, opacities, scales, rotations,
                                     cov3Ds_precomp, raster_settings)


class GaussianRasterizer(nn.Module):
    def __init__(self, raster_settings: GaussianRasterizationSettings):
        super().__init__()
        self.raster_settings = raster_settings

    def markVisible(self, positions: torch.Tensor) -> torch.Tensor:
        """Boolean mask of points in front of the near-cull plane."""
        with torch.no_grad():
            pos = _f32(positions, "positions")
            P = 0 if pos is None else pos.shape[0]
            out = torch.empty((P,), dtype=torch.uint8, device=positions.device)
            if P:
                view = _f32(self.raster_settings.viewmatrix, "viewmatrix")
                _run("mark_visible", _lib.gsr_mark_visible, P, _ptr(pos), _ptr(view), _ptr(out), _stream())
            return out.bool()

    def forward(self, means3D, means2D, opacities, shs=None, colors_precomp=None, scales=None, rotations=None,
                cov3D_precomp=None):
        if (shs is None and colors_precomp is None) or (shs is not None and colors_precomp is not None):
            raise Exception("Please provide excatly one of either SHs or precomputed colors!")
        if ((scales is None or rotations is None) and cov3D_precomp is None) or \
                ((scales is not None or rotations is not None) and cov3D_precomp is not None):
            raise Exception("Please provide exactly one of either scale/rotation pair or precomputed 3D covariance!")
        empty = torch.Tensor([])
        return rasterize_gaussians(
            means3D, means2D,
            empty if shs is None else shs, empty if colors_precomp is None else colors_precomp, opacities,
            empty if scales is None else scales, empty if rotations is None else rotations,
            empty if cov3D_precomp is None else cov3D_precomp, self.raster_settings)
