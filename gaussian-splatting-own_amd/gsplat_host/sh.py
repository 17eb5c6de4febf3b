"""Real spherical harmonics up to degree 3, host side (PyTorch).

``eval_sh`` has the signature and semantics of the reference's
(``utils/sh_utils.py:57-112``): ``sh`` is ``[..., C, K]`` with K >= (deg+1)^2,
``dirs`` are unit vectors ``[..., 3]``, the result is ``[..., C]``.  It is
the Python-side alternative to the in-kernel SH evaluation that the reference
selects with ``pipe.convert_SHs_python`` (``gaussian_renderer/__init__.py:72-78``).
Written as basis-vector times coefficients rather than term by term.
"""
from __future__ import annotations

import torch

C0 = 0.28209479177387814
C1 = 0.4886025119029199
C2 = (1.0925484305920792, -1.0925484305920792, 0.31539156525252005, -1.0925484305920792, 0.5462742152960396)
C3 = (-0.5900435899266435, 2.890611442640554, -0.4570457994644658, 0.3731763325901154, -0.4570457994644658,
      1.445305721320277, -0.5900435899266435)


def sh_basis(deg: int, dirs: torch.Tensor) -> torch.Tensor:
    """[..., (deg+1)^2] basis values; basis[k] multiplies coefficient k."""
    if not 0 <= deg <= 3:
        raise ValueError("SH degree must be 0..3")
    x, y, z = dirs[..., 0], dirs[..., 1], dirs[..., 2]
    b = [torch.full_like(x, C0)]
    if deg > 0:
        b += [-C1 * y, C1 * z, -C1 * x]
    if deg > 1:
        xx, yy, zz, xy, yz, xz = x * x, y * y, z * z, x * y, y * z, x * z
        b += [C2[0] * xy, C2[1] * yz, C2[2] * (2.0 * zz - xx - yy), C2[3] * xz, C2[4] * (xx - yy)]
    if deg > 2:
        b += [C3[0] * y * (3 * xx - yy), C3[1] * xy * z, C3[2] * y * (4 * zz - xx - yy),
              C3[3] * z * (2 * zz - 3 * xx - 3 * yy), C3[4] * x * (4 * zz - xx - yy), C3[5] * z * (xx - yy),
              C3[6] * x * (xx - 3 * yy)]
    return torch.stack(b, dim=-1)


def eval_sh(deg: int, sh: torch.Tensor, dirs: torch.Tensor) -> torch.Tensor:
    basis = sh_basis(deg, dirs)                       # [..., K']
    k = basis.shape[-1]
    return (sh[..., :k] * basis[..., None, :]).sum(dim=-1)


def RGB2SH(rgb):
    """utils/sh_utils.py:114-115."""
    return (rgb - 0.5) / C0


def SH2RGB(sh):
    """utils/sh_utils.py:117-118."""
    return sh * C0 + 0.5
