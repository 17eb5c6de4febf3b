"""The C-ABI library loads (no GPU needed for that) and exports every entry
point that include/gsr.h declares; the drop-in module exposes the reference's
operator surface.  No compute calls here."""
import ctypes
import inspect
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "gaussian-splatting-own_amd", "diff_gaussian_rasterization", "libgsr_hip.so")
HEADER = os.path.join(ROOT, "include", "gsr.h")


def declared_symbols():
    text = open(HEADER).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(gsr_[a-z0-9_]+)\s*\(", text)))


def test_header_declares_the_expected_entry_points():
    syms = declared_symbols()
    for s in ("gsr_version", "gsr_last_error", "gsr_preprocess_fwd", "gsr_mark_visible", "gsr_totals_words",
              "gsr_bin_sort_bytes", "gsr_bin_sort", "gsr_bin_sort_status", "gsr_debug_sort_spin_limit",
              "gsr_bin_sort_error_word", "gsr_bin_lists_bytes", "gsr_bin_lists", "gsr_blend_fwd", "gsr_blend_bwd",
              "gsr_preprocess_bwd"):
        assert s in syms


def test_library_exports_every_declared_symbol():
    if not os.path.exists(LIB):
        pytest.fail(f"{LIB} is not built: run `make -C gaussian-splatting-own_amd/csrc`")
    lib = ctypes.CDLL(LIB)
    for s in declared_symbols():
        assert hasattr(lib, s), s
    lib.gsr_version.restype = ctypes.c_int
    assert lib.gsr_version() == 6
    lib.gsr_last_error.restype = ctypes.c_char_p
    assert isinstance(lib.gsr_last_error(), bytes)


def test_argument_errors_are_reported_without_touching_the_gpu():
    lib = ctypes.CDLL(LIB)
    lib.gsr_last_error.restype = ctypes.c_char_p
    rc = lib.gsr_blend_fwd(0, 0, None, None, None, None, None, None, None, None)
    assert rc == -1 and b"bad image size" in lib.gsr_last_error()
    rc = lib.gsr_mark_visible(5, None, None, None, None)
    assert rc == -1 and b"null pointer" in lib.gsr_last_error()


def test_module_surface_matches_the_reference_call_sites():
    import diff_gaussian_rasterization as dgr
    # gaussian_renderer/__init__.py:36-49 constructs the settings with exactly these keywords
    assert dgr.GaussianRasterizationSettings._fields == (
        "image_height", "image_width", "tanfovx", "tanfovy", "bg", "scale_modifier", "viewmatrix", "projmatrix",
        "sh_degree", "campos", "prefiltered", "debug")
    # :51 GaussianRasterizer(raster_settings=...), :85-93 keyword call
    assert list(inspect.signature(dgr.GaussianRasterizer.__init__).parameters)[1:] == ["raster_settings"]
    params = list(inspect.signature(dgr.GaussianRasterizer.forward).parameters)[1:]
    assert set(params) == {"means3D", "means2D", "opacities", "shs", "colors_precomp", "scales", "rotations",
                           "cov3D_precomp"}
    assert hasattr(dgr.GaussianRasterizer, "markVisible")


def test_scratch_and_size_checks_come_before_any_launch():
    """Sizes are queried without a GPU; wrong scratch, sizes or totals are refused with a message
    and no kernel is launched (these calls run on a machine without a GPU)."""
    lib = ctypes.CDLL(LIB)
    lib.gsr_last_error.restype = ctypes.c_char_p
    lib.gsr_bin_sort_bytes.restype = ctypes.c_size_t
    lib.gsr_bin_lists_bytes.restype = ctypes.c_size_t
    lib.gsr_totals_words.restype = ctypes.c_size_t
    vp, sz, u32 = ctypes.c_void_p, ctypes.c_size_t, ctypes.c_uint32
    lib.gsr_bin_sort.argtypes = [ctypes.c_int, vp, vp, vp, vp, sz, vp]
    lib.gsr_bin_lists.argtypes = [ctypes.c_int, u32, u32, u32, u32, ctypes.c_int, ctypes.c_int, vp, vp, vp, vp, vp, vp, vp,
                                  vp, sz, vp]

    sizes = [lib.gsr_bin_sort_bytes(p) for p in (0, 1, 4096, 4097, 1_000_000, 6_000_000)]
    assert all(s % 256 == 0 for s in sizes) and sizes == sorted(sizes)
    assert sizes[4] >= 24 * 1_000_000                      # two key and two value buffers
    assert lib.gsr_totals_words(0) == 8 and lib.gsr_totals_words(257) == 8 + 4 * 2
    assert lib.gsr_bin_lists_bytes(0, 0, 0, 1920, 1080) == 256
    assert lib.gsr_bin_lists_bytes(1000, 1200, 3000, 1920, 1080) % 256 == 0

    fake = vp(0x1000)                                      # never dereferenced: the checks fail first
    assert lib.gsr_bin_sort(-1, fake, fake, fake, fake, 1 << 20, None) == -1
    assert lib.gsr_bin_sort(1 << 30, fake, fake, fake, fake, 1 << 20, None) == -1
    assert lib.gsr_bin_sort(1000, None, fake, fake, fake, 1 << 20, None) == -1 and b"null" in lib.gsr_last_error()
    rc = lib.gsr_bin_sort(1000, fake, fake, fake, fake, 16, None)
    assert rc != 0 and b"scratch too small" in lib.gsr_last_error()
    rc = lib.gsr_bin_sort(1000, fake, fake, fake, vp(0x1010), 1 << 20, None)
    assert rc != 0 and b"256-byte aligned" in lib.gsr_last_error()
    assert lib.gsr_bin_sort(0, None, None, None, None, 0, None) == 0          # nothing to do
    lib.gsr_bin_sort_status.argtypes = [ctypes.c_int, vp, sz, vp]
    assert lib.gsr_bin_sort_status(-1, fake, 1 << 20, None) == -1
    assert lib.gsr_bin_sort_status(1000, None, 1 << 20, None) == -1 and b"null" in lib.gsr_last_error()
    assert lib.gsr_bin_sort_status(1000, fake, 16, None) == -3
    assert lib.gsr_bin_sort_status(0, None, 0, None) == 0
    lib.gsr_bin_sort_error_word.argtypes, lib.gsr_bin_sort_error_word.restype = [vp], vp
    assert lib.gsr_bin_sort_error_word(None) is None
    word = lib.gsr_bin_sort_error_word(vp(0x100000))       # pointer arithmetic only
    assert 0x100000 < word < 0x100000 + lib.gsr_bin_sort_bytes(1) and word % 4 == 0

    rc = lib.gsr_bin_lists(1000, 900, 5000, 800, 3000, 640, 480, fake, fake, fake, None, None, None, None, fake, 1 << 24, None)
    assert rc == -1 and b"inconsistent totals" in lib.gsr_last_error()      # E1 < V
    rc = lib.gsr_bin_lists(1000, 900, 5000, 950, 3000, 640, 480, fake, fake, fake, None, None, None, None, fake, 16, None)
    assert rc != 0 and b"scratch too small" in lib.gsr_last_error()
    rc = lib.gsr_bin_lists(1000, 900, 5000, 950, 3000, 70000, 480, fake, fake, fake, None, None, None, None, fake, 1 << 24, None)
    assert rc == -1 and b"too large" in lib.gsr_last_error()


def test_the_backward_block_is_laid_out_as_the_library_expects():
    """Host logic of the binding, no GPU: a prepared forward makes two allocations for its backward.  The first holds
    the (P,16) accumulator and, behind it, the P `touched` bytes: what the forward clears.  The second holds every
    gradient tensor: what the blend backward clears, and all that stays alive once autograd has installed the
    gradients (the accumulator's 64 bytes per Gaussian are not pinned by them).  Every piece starts on a 256-byte
    boundary and none overlaps another."""
    import torch
    import diff_gaussian_rasterization as dgr
    for P, shapes in ((1, dgr._gradient_shapes(1, True, n_rest=15)), (1001, dgr._gradient_shapes(1001, True, n_rest=0)),
                      (777, dgr._gradient_shapes(777, False, M=16)), (64, dgr._gradient_shapes(64, False, colors=True, cov3D=True))):
        first, dsplat, touched, second, views = dgr._carve_zero_block(P, shapes, torch.device("cpu"))
        base, end = first.data_ptr(), first.data_ptr() + first.numel() * 4
        assert dsplat.data_ptr() == base and dsplat.shape == (P, 16)
        assert touched.dtype == torch.uint8 and touched.numel() >= P
        assert (touched.data_ptr() - base) % 256 == 0 and dsplat.data_ptr() + dsplat.numel() * 4 <= touched.data_ptr()
        assert touched.data_ptr() + touched.numel() <= end and (end - base) % 16 == 0
        gbase, gend = second.data_ptr(), second.data_ptr() + second.numel() * 4
        assert gend <= base or gbase >= end                          # two allocations
        assert set(views) == set(shapes)
        spans = []
        for name, v in views.items():
            assert tuple(v.shape) == tuple(shapes[name]) and v.is_contiguous() and v.dtype == torch.float32
            assert v.untyped_storage().data_ptr() == second.untyped_storage().data_ptr()
            spans.append((v.data_ptr(), v.numel() * 4))
        spans.sort()
        for (a, n), (b, _) in zip(spans, spans[1:]):
            assert (a - gbase) % 256 == 0 and a + n <= b
        assert spans[0][0] == gbase and spans[-1][0] + spans[-1][1] <= gend and (gend - gbase) % 16 == 0
        # writing one piece leaves the others alone
        first.zero_()
        second.zero_()
        name0 = next(iter(views))
        views[name0].fill_(1.0)
        assert float(dsplat.abs().max()) == 0.0 and int(touched.max()) == 0
        assert all(float(v.abs().max()) == 0.0 for k, v in views.items() if k != name0)


def test_static_sort_scratch_is_as_large_as_the_layout_at_its_capacity():
    """The device array behind gsr_bin_sort_static is declared with a literal size; the sort's layout at the
    capacity (and at the key counts where the keys per thread change) must fit it."""
    lib = ctypes.CDLL(LIB)
    lib.gsr_bin_sort_static_bytes.restype = ctypes.c_size_t
    lib.gsr_bin_sort_bytes.restype = ctypes.c_size_t
    have, need = lib.gsr_bin_sort_static_bytes(0), lib.gsr_bin_sort_static_bytes(1)
    assert need <= have and have - need < 4096                     # no slack to speak of either
    assert need == lib.gsr_bin_sort_bytes(1 << 20)
    for p in (1, 4096, 199_999, 200_000, 200_001, 999_999, 1 << 20):
        assert lib.gsr_bin_sort_bytes(p) <= have, p


def test_exclusive_device_switch_is_exported_and_callable_without_a_gpu():
    lib = ctypes.CDLL(LIB)
    lib.gsr_assume_exclusive_device.restype = None
    lib.gsr_assume_exclusive_device(1)
    lib.gsr_assume_exclusive_device(0)
