"""CPU oracle of the Gaussian rasterizer: ctypes front end of ``gsr_oracle.c``.

TEST INFRASTRUCTURE ONLY.  Nothing under ``gaussian-splatting-own_amd/`` may
import this package; it exists so that ``tests/``, the smoke check and the
benchmark's ``cpu_baseline`` leg have something independent to compare the HIP
kernels with.

PARITY UNPINNED at the rasterizer boundary (the reference holds neither source
nor tests of ``diff_gaussian_rasterization``, see the header of
``gsr_oracle.c``); the SH, covariance and camera stages are pinned to the
reference's Python by ``tests/golden``.

The stage functions mirror the call sequence behind
``GaussianRasterizer.forward`` (reference call site
``gaussian_renderer/__init__.py:85-93``) and take / return numpy arrays.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from dataclasses import dataclass

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_BUILD = os.path.join(_HERE, "_build")

TILE = 16


def build(force: bool = False) -> None:
    """Compile both precisions of the oracle with gcc (seconds)."""
    if force:
        subprocess.check_call(["make", "-C", _HERE, "clean"], stdout=subprocess.DEVNULL)
    subprocess.check_call(["make", "-C", _HERE], stdout=subprocess.DEVNULL)


def _cam_struct(real):
    class Cam(C.Structure):
        _fields_ = [
            ("P", C.c_int32), ("D", C.c_int32), ("M", C.c_int32),
            ("W", C.c_int32), ("H", C.c_int32), ("prefiltered", C.c_int32),
            ("tanfovx", real), ("tanfovy", real), ("scale_modifier", real),
            ("view", real * 16), ("proj", real * 16), ("campos", real * 3), ("bg", real * 3),
        ]
    return Cam


@dataclass
class Settings:
    """The 12 fields of GaussianRasterizationSettings
    (gaussian_renderer/__init__.py:36-49) as plain Python / numpy values."""
    image_height: int
    image_width: int
    tanfovx: float
    tanfovy: float
    bg: np.ndarray
    scale_modifier: float
    viewmatrix: np.ndarray
    projmatrix: np.ndarray
    sh_degree: int
    campos: np.ndarray
    prefiltered: bool = False
    debug: bool = False


class Oracle:
    """One precision of the oracle.  ``Oracle(np.float32)`` is the parity
    oracle, ``Oracle(np.float64)`` the gradient-check build."""

    def __init__(self, dtype=np.float32, threads: int | None = None, kernel_order: bool = False):
        """``kernel_order`` (fp32 only): the build that evaluates the Gaussian's exponent in the HIP blend
        kernels' operation order (log2 domain, fused multiply-adds) instead of the published expression's; it
        exists to attribute the difference between the two roundings, the default build stays the oracle."""
        self.dtype = np.dtype(dtype)
        if kernel_order and self.dtype != np.float32:
            raise ValueError("the kernel-order build exists in fp32 only")
        self.kernel_order = bool(kernel_order)
        name = ("liboracle_f32k.so" if kernel_order else "liboracle_f32.so") if self.dtype == np.float32 else "liboracle_f64.so"
        path = os.path.join(_BUILD, name)
        if not os.path.exists(path):
            build()
        if threads is not None:
            os.environ["OMP_NUM_THREADS"] = str(threads)
        self.lib = C.CDLL(path)
        self.real = C.c_float if self.dtype == np.float32 else C.c_double
        assert self.lib.gso_real_bytes() == self.dtype.itemsize
        assert self.lib.gso_kernel_order() == int(self.kernel_order)
        self.Cam = _cam_struct(self.real)
        self.lib.gso_bin.restype = C.c_int

    # -- helpers ---------------------------------------------------------
    def _a(self, x, dtype=None):
        if x is None:
            return None
        return np.ascontiguousarray(np.asarray(x), dtype=dtype or self.dtype)

    @staticmethod
    def _p(x):
        return None if x is None else x.ctypes.data_as(C.c_void_p)

    def cam(self, s: Settings, P: int, M: int):
        c = self.Cam()
        c.P, c.D, c.M, c.W, c.H = P, int(s.sh_degree), M, int(s.image_width), int(s.image_height)
        c.prefiltered = int(bool(s.prefiltered))
        c.tanfovx, c.tanfovy, c.scale_modifier = float(s.tanfovx), float(s.tanfovy), float(s.scale_modifier)
        for i, v in enumerate(np.asarray(s.viewmatrix, dtype=self.dtype).reshape(-1)):
            c.view[i] = v
        for i, v in enumerate(np.asarray(s.projmatrix, dtype=self.dtype).reshape(-1)):
            c.proj[i] = v
        for i, v in enumerate(np.asarray(s.campos, dtype=self.dtype).reshape(-1)):
            c.campos[i] = v
        for i, v in enumerate(np.asarray(s.bg, dtype=self.dtype).reshape(-1)):
            c.bg[i] = v
        return c

    # -- stages ----------------------------------------------------------
    def preprocess(self, s: Settings, means3D, opacities, scales=None, rotations=None, shs=None,
                   colors_precomp=None, cov3D_precomp=None):
        means3D = self._a(means3D)
        P = means3D.shape[0]
        shs = self._a(shs)
        M = 0 if shs is None else shs.shape[1]
        cam = self.cam(s, P, M)
        scales, rotations = self._a(scales), self._a(rotations)
        opacities = self._a(opacities).reshape(-1)
        colors_precomp, cov3D_precomp = self._a(colors_precomp), self._a(cov3D_precomp)
        dt = self.dtype
        out = dict(
            radii=np.zeros(P, np.int32), means2D=np.zeros((P, 2), dt), depths=np.zeros(P, dt),
            cov3D=np.zeros((P, 6), dt), rgb=np.zeros((P, 3), dt), conic_opacity=np.zeros((P, 4), dt),
            tiles_touched=np.zeros(P, np.uint32), clamped=np.zeros((P, 3), np.uint8),
            rects=np.zeros((P, 4), np.int32),
        )
        self.lib.gso_preprocess_fwd(
            C.byref(cam), self._p(means3D), self._p(scales), self._p(rotations), self._p(opacities),
            self._p(shs), self._p(colors_precomp), self._p(cov3D_precomp), self._p(out["radii"]),
            self._p(out["means2D"]), self._p(out["depths"]), self._p(out["cov3D"]), self._p(out["rgb"]),
            self._p(out["conic_opacity"]), self._p(out["tiles_touched"]), self._p(out["clamped"]),
            self._p(out["rects"]))
        out["_cam"] = cam
        return out

    def mark_visible(self, s: Settings, means3D):
        means3D = self._a(means3D)
        cam = self.cam(s, means3D.shape[0], 0)
        present = np.zeros(means3D.shape[0], np.uint8)
        self.lib.gso_mark_visible(C.byref(cam), self._p(means3D), self._p(present))
        return present.astype(bool)

    def bin(self, s: Settings, pre):
        cam = pre["_cam"]
        offsets = np.cumsum(pre["tiles_touched"], dtype=np.uint64)
        n_dup = int(offsets[-1]) if offsets.size else 0
        assert n_dup < 2 ** 32
        offsets32 = offsets.astype(np.uint32)
        gx, gy = (s.image_width + TILE - 1) // TILE, (s.image_height + TILE - 1) // TILE
        out = dict(
            offsets=offsets32, n_dup=n_dup,
            keys_unsorted=np.zeros(n_dup, np.uint64), vals_unsorted=np.zeros(n_dup, np.uint32),
            keys=np.zeros(n_dup, np.uint64), point_list=np.zeros(n_dup, np.uint32),
            ranges=np.zeros((gx * gy, 2), np.uint32),
        )
        depths32 = np.ascontiguousarray(pre["depths"], dtype=np.float32)
        rc = self.lib.gso_bin(
            C.byref(cam), self._p(pre["radii"]), self._p(pre["means2D"]), self._p(depths32),
            self._p(offsets32), C.c_uint64(n_dup), self._p(out["keys_unsorted"]),
            self._p(out["vals_unsorted"]), self._p(out["keys"]), self._p(out["point_list"]),
            self._p(out["ranges"]))
        if rc != 0:
            raise RuntimeError(f"oracle bin failed: {rc}")
        return out

    def blend_fwd(self, s: Settings, pre, binned, colors=None, want_margin=True):
        cam = pre["_cam"]
        H, W = s.image_height, s.image_width
        dt = self.dtype
        colors = pre["rgb"] if colors is None else self._a(colors)
        out = dict(color=np.zeros((3, H, W), dt), final_T=np.zeros((H, W), dt),
                   n_contrib=np.zeros((H, W), np.uint32),
                   margin=np.zeros((H, W), np.float32) if want_margin else None)
        n_eval = C.c_uint64(0)
        self.lib.gso_blend_fwd(
            C.byref(cam), self._p(binned["ranges"]), self._p(binned["point_list"]),
            self._p(pre["means2D"]), self._p(colors), self._p(pre["conic_opacity"]),
            self._p(out["color"]), self._p(out["final_T"]), self._p(out["n_contrib"]),
            self._p(out["margin"]), C.byref(n_eval))
        out["n_evaluated"] = n_eval.value
        return out

    def blend_bwd(self, s: Settings, pre, binned, fwd, dL_dpix, colors=None):
        cam = pre["_cam"]
        P = cam.P
        dt = self.dtype
        colors = pre["rgb"] if colors is None else self._a(colors)
        dL_dpix = self._a(dL_dpix)
        out = dict(dL_dmean2D=np.zeros((P, 2), dt), dL_dconic=np.zeros((P, 3), dt),
                   dL_dopacity=np.zeros(P, dt), dL_dcolor=np.zeros((P, 3), dt))
        self.lib.gso_blend_bwd(
            C.byref(cam), self._p(binned["ranges"]), self._p(binned["point_list"]),
            self._p(pre["means2D"]), self._p(colors), self._p(pre["conic_opacity"]),
            self._p(fwd["final_T"]), self._p(fwd["n_contrib"]), self._p(dL_dpix),
            self._p(out["dL_dmean2D"]), self._p(out["dL_dconic"]), self._p(out["dL_dopacity"]),
            self._p(out["dL_dcolor"]))
        return out

    def preprocess_bwd(self, s: Settings, pre, bb, means3D, scales=None, rotations=None, shs=None):
        cam = pre["_cam"]
        P = cam.P
        dt = self.dtype
        means3D, scales, rotations, shs = self._a(means3D), self._a(scales), self._a(rotations), self._a(shs)
        out = dict(dL_dmeans3D=np.zeros((P, 3), dt), dL_dcov3D=np.zeros((P, 6), dt),
                   dL_dshs=None if shs is None else np.zeros(shs.shape, dt),
                   dL_dscales=None if scales is None else np.zeros((P, 3), dt),
                   dL_drotations=None if scales is None else np.zeros((P, 4), dt))
        self.lib.gso_preprocess_bwd(
            C.byref(cam), self._p(pre["radii"]), self._p(means3D), self._p(scales), self._p(rotations),
            self._p(shs), self._p(pre["clamped"]), self._p(pre["cov3D"]), self._p(bb["dL_dmean2D"]),
            self._p(bb["dL_dconic"]), self._p(bb["dL_dcolor"]), self._p(out["dL_dmeans3D"]),
            self._p(out["dL_dcov3D"]), self._p(out["dL_dshs"]), self._p(out["dL_dscales"]),
            self._p(out["dL_drotations"]))
        return out

    # -- fragments pinned by the golden vectors ------------------------------
    def sh_eval(self, deg, shs, dirs):
        shs, dirs = self._a(shs), self._a(dirs)
        out = np.zeros((shs.shape[0], 3), self.dtype)
        self.lib.gso_sh_eval(int(deg), int(shs.shape[1]), int(shs.shape[0]), self._p(dirs), self._p(shs), self._p(out))
        return out

    def cov3d(self, scales, mod, rots):
        scales, rots = self._a(scales), self._a(rots)
        out = np.zeros((scales.shape[0], 6), self.dtype)
        self.lib.gso_cov3d(int(scales.shape[0]), self._p(scales), self.real(mod), self._p(rots), self._p(out))
        return out

    # -- whole op ----------------------------------------------------------
    def forward(self, s: Settings, means3D, opacities, scales=None, rotations=None, shs=None,
                colors_precomp=None, cov3D_precomp=None, want_margin=True):
        """(color (3,H,W), radii (P,)) plus every intermediate, as a dict."""
        pre = self.preprocess(s, means3D, opacities, scales, rotations, shs, colors_precomp, cov3D_precomp)
        binned = self.bin(s, pre)
        fwd = self.blend_fwd(s, pre, binned, want_margin=want_margin)
        return dict(pre=pre, bin=binned, fwd=fwd, color=fwd["color"], radii=pre["radii"])

    def backward(self, s: Settings, state, dL_dpix, means3D, scales=None, rotations=None, shs=None):
        """Gradients in the order the autograd Function returns them."""
        pre, binned, fwd = state["pre"], state["bin"], state["fwd"]
        bb = self.blend_bwd(s, pre, binned, fwd, dL_dpix)
        pb = self.preprocess_bwd(s, pre, bb, means3D, scales, rotations, shs)
        P = pre["_cam"].P
        g2 = np.zeros((P, 3), self.dtype)
        g2[:, :2] = bb["dL_dmean2D"]
        return dict(means3D=pb["dL_dmeans3D"], means2D=g2, shs=pb["dL_dshs"], colors_precomp=bb["dL_dcolor"],
                    opacities=bb["dL_dopacity"].reshape(P, 1), scales=pb["dL_dscales"],
                    rotations=pb["dL_drotations"], cov3D_precomp=pb["dL_dcov3D"], blend=bb)
