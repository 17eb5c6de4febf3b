"""``distCUDA2`` on MI355X: same module path and call as the reference's native extension
(``from simple_knn._C import distCUDA2``, ``scene/gaussian_model.py:20``; called once, at
``:134``, to initialise the scales).  The arithmetic runs in ``libgsr_hip.so``
(``gsr_knn3_mean_dist2``); there is no CPU path.
"""
from __future__ import annotations

import ctypes as C

import torch

import diff_gaussian_rasterization as _dgr

_lib = _dgr._lib
_lib.gsr_knn3_bytes.restype = C.c_size_t
_lib.gsr_knn3_bytes.argtypes = [C.c_int]
_lib.gsr_knn3_mean_dist2.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]


def distCUDA2(points: torch.Tensor) -> torch.Tensor:
    """(P,3) float32 GPU tensor -> (P,) mean squared distance to the three nearest other points."""
    if points.dim() != 2 or points.shape[1] != 3:
        raise ValueError(f"expected (P,3) points, got {tuple(points.shape)}")
    pts = _dgr._f32(points, "points")
    P = points.shape[0]
    out = torch.empty((P,), dtype=torch.float32, device=points.device)
    if P == 0:
        return out
    nb = _lib.gsr_knn3_bytes(P)
    scratch = torch.empty((nb,), dtype=torch.uint8, device=points.device)
    _dgr._run("knn3", _lib.gsr_knn3_mean_dist2, P, _dgr._ptr(pts), _dgr._ptr(out), _dgr._ptr(scratch), nb,
              _dgr._stream())
    return out
