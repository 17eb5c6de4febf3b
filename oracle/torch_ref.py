"""Dense, differentiable PyTorch (fp64, CPU) formulation of the rasterizer.

TEST INFRASTRUCTURE ONLY.  Independent of ``gsr_oracle.c``: every pixel is
composited against every Gaussian with masks instead of tiles and lists, and
gradients come from autograd instead of hand-derived formulas.  It exists to
check the C oracle's explicit backward; it is O(P*H*W) memory, so keep P in
the hundreds and images tiny.

The two places where the rasterizer module's published backward is NOT the
derivative of its forward are reproduced on purpose, so that autograd yields
the module's gradients:
  * alpha = min(0.99, opacity*G): the cap is transparent to gradients;
  * the clamp of x/z, y/z to +-1.3 tan(fov/2) before the EWA Jacobian: where
    active, the clamped coordinate is a constant.
The 1e-7 added to det^2 in the conic backward is NOT reproduced (autograd
differentiates 1/det exactly); compare with a relative tolerance.
"""
from __future__ import annotations

import math

import torch

import numpy as np


def _f(v):
    """Constants rounded to fp32 first: the C oracle's fp64 build keeps the
    fp32 values of include/gsr_constants.h."""
    return float(np.float32(v))


TILE = 16
SH_C0 = _f(0.28209479177387814)
SH_C1 = _f(0.4886025119029199)
SH_C2 = [_f(v) for v in (1.0925484305920792, -1.0925484305920792, 0.31539156525252005,
                         -1.0925484305920792, 0.5462742152960396)]
SH_C3 = [_f(v) for v in (-0.5900435899266435, 2.890611442640554, -0.4570457994644658, 0.3731763325901154,
                         -0.4570457994644658, 1.445305721320277, -0.5900435899266435)]
NEAR, W_EPS, CLAMP, DILATE, LAM_FLOOR = _f(0.2), _f(1e-7), _f(1.3), _f(0.3), _f(0.1)
A_MAX, A_MIN, T_STOP = _f(0.99), _f(1.0 / 255.0), _f(1e-4)


def sh_to_rgb(deg, sh, dirs):
    """sh (P,M,3) coefficient-major; dirs (P,3) unit.  Same polynomial as the
    reference's eval_sh (utils/sh_utils.py:57-112), written on (P,M,3)."""
    x, y, z = dirs[:, 0:1], dirs[:, 1:2], dirs[:, 2:3]
    res = SH_C0 * sh[:, 0]
    if deg > 0:
        res = res - SH_C1 * y * sh[:, 1] + SH_C1 * z * sh[:, 2] - SH_C1 * x * sh[:, 3]
    if deg > 1:
        xx, yy, zz, xy, yz, xz = x * x, y * y, z * z, x * y, y * z, x * z
        res = (res + SH_C2[0] * xy * sh[:, 4] + SH_C2[1] * yz * sh[:, 5]
               + SH_C2[2] * (2.0 * zz - xx - yy) * sh[:, 6] + SH_C2[3] * xz * sh[:, 7]
               + SH_C2[4] * (xx - yy) * sh[:, 8])
    if deg > 2:
        res = (res + SH_C3[0] * y * (3 * xx - yy) * sh[:, 9] + SH_C3[1] * xy * z * sh[:, 10]
               + SH_C3[2] * y * (4 * zz - xx - yy) * sh[:, 11]
               + SH_C3[3] * z * (2 * zz - 3 * xx - 3 * yy) * sh[:, 12]
               + SH_C3[4] * x * (4 * zz - xx - yy) * sh[:, 13] + SH_C3[5] * z * (xx - yy) * sh[:, 14]
               + SH_C3[6] * x * (xx - 3 * yy) * sh[:, 15])
    return res


def quat_to_rot(q):
    r, x, y, z = q[:, 0], q[:, 1], q[:, 2], q[:, 3]
    R = torch.stack([
        1 - 2 * (y * y + z * z), 2 * (x * y - r * z), 2 * (x * z + r * y),
        2 * (x * y + r * z), 1 - 2 * (x * x + z * z), 2 * (y * z - r * x),
        2 * (x * z - r * y), 2 * (y * z + r * x), 1 - 2 * (x * x + y * y)], dim=1)
    return R.reshape(-1, 3, 3)


def cov3d(scales, mod, q):
    L = quat_to_rot(q) * (mod * scales)[:, None, :]
    S = L @ L.transpose(1, 2)
    return torch.stack([S[:, 0, 0], S[:, 0, 1], S[:, 0, 2], S[:, 1, 1], S[:, 1, 2], S[:, 2, 2]], dim=1)


def render_dense(W, H, tanfovx, tanfovy, bg, scale_modifier, view, proj, sh_degree, campos,
                 means3D, opacities, scales=None, rotations=None, shs=None, colors_precomp=None,
                 cov3D_precomp=None):
    """Returns (color (3,H,W), radii (P,), aux) with aux['pix'] the (P,2)
    pixel-space means (retain_grad'ed when differentiable)."""
    dt = means3D.dtype
    P = means3D.shape[0]
    ones = torch.ones(P, 1, dtype=dt)
    hom = torch.cat([means3D, ones], dim=1)
    pv = (hom @ view)[:, :3]
    ph = hom @ proj
    pw = 1.0 / (ph[:, 3] + W_EPS)
    ndc = ph[:, :2] * pw[:, None]
    c6 = cov3D_precomp if cov3D_precomp is not None else cov3d(scales, scale_modifier, rotations)

    fx, fy = W / (2.0 * tanfovx), H / (2.0 * tanfovy)
    limx, limy = CLAMP * tanfovx, CLAMP * tanfovy
    tz = pv[:, 2]
    safe_tz = torch.where(tz > NEAR, tz, torch.ones_like(tz))
    txtz, tytz = pv[:, 0] / safe_tz, pv[:, 1] / safe_tz
    cx, cy = (txtz < -limx) | (txtz > limx), (tytz < -limy) | (tytz > limy)
    tx = torch.where(cx, (txtz.clamp(-limx, limx) * safe_tz).detach(), pv[:, 0])
    ty = torch.where(cy, (tytz.clamp(-limy, limy) * safe_tz).detach(), pv[:, 1])
    J = torch.zeros(P, 2, 3, dtype=dt)
    J[:, 0, 0] = fx / safe_tz
    J[:, 0, 2] = -(fx * tx) / (safe_tz * safe_tz)
    J[:, 1, 1] = fy / safe_tz
    J[:, 1, 2] = -(fy * ty) / (safe_tz * safe_tz)
    Wm = view[:3, :3].transpose(0, 1)      # W[k][j] = view[j][k]
    Mx = J @ Wm
    Sig = torch.stack([c6[:, 0], c6[:, 1], c6[:, 2], c6[:, 1], c6[:, 3], c6[:, 4],
                       c6[:, 2], c6[:, 4], c6[:, 5]], dim=1).reshape(P, 3, 3)
    c2 = Mx @ Sig @ Mx.transpose(1, 2)
    a, b, c = c2[:, 0, 0] + DILATE, c2[:, 0, 1], c2[:, 1, 1] + DILATE
    det = a * c - b * b
    ok = (tz > NEAR) & (det != 0)
    safe_det = torch.where(ok, det, torch.ones_like(det))
    conA, conB, conC = c / safe_det, -b / safe_det, a / safe_det
    with torch.no_grad():
        mid = 0.5 * (a + c)
        lam = mid + torch.sqrt(torch.clamp_min(mid * mid - det, LAM_FLOOR))
        radius = torch.ceil(3.0 * torch.sqrt(lam))
    pix = torch.stack([((ndc[:, 0] + 1.0) * W - 1.0) * 0.5, ((ndc[:, 1] + 1.0) * H - 1.0) * 0.5], dim=1)
    if pix.requires_grad:
        pix.retain_grad()
    gx, gy = (W + TILE - 1) // TILE, (H + TILE - 1) // TILE
    with torch.no_grad():
        x0 = torch.clamp((pix[:, 0] - radius) / TILE, 0, gx).to(torch.int64)
        x1 = torch.clamp((pix[:, 0] + radius + TILE - 1) / TILE, 0, gx).to(torch.int64)
        y0 = torch.clamp((pix[:, 1] - radius) / TILE, 0, gy).to(torch.int64)
        y1 = torch.clamp((pix[:, 1] + radius + TILE - 1) / TILE, 0, gy).to(torch.int64)
        vis = ok & ((x1 - x0) * (y1 - y0) > 0)
        radii = torch.where(vis, radius, torch.zeros_like(radius)).to(torch.int32)

    if colors_precomp is not None:
        rgb = colors_precomp
    else:
        d = means3D - campos[None, :]
        d = d / d.norm(dim=1, keepdim=True)
        rgb = torch.clamp_min(sh_to_rgb(sh_degree, shs, d) + 0.5, 0.0)

    idx = torch.nonzero(vis).reshape(-1)
    order = torch.argsort(pv[idx, 2].detach().to(torch.float32), stable=True)
    idx = idx[order]
    ys, xs = torch.meshgrid(torch.arange(H, dtype=dt), torch.arange(W, dtype=dt), indexing="ij")
    tyi, txi = (ys / TILE).floor().to(torch.int64), (xs / TILE).floor().to(torch.int64)
    dx = pix[idx, 0][:, None, None] - xs[None]
    dy = pix[idx, 1][:, None, None] - ys[None]
    A, B, Cc = conA[idx][:, None, None], conB[idx][:, None, None], conC[idx][:, None, None]
    power = -0.5 * (A * dx * dx + Cc * dy * dy) - B * dx * dy
    araw = opacities.reshape(-1)[idx][:, None, None] * torch.exp(torch.clamp_max(power, 0.0))
    alpha = araw + (torch.clamp_max(araw, A_MAX) - araw).detach()
    inrect = ((txi[None] >= x0[idx][:, None, None]) & (txi[None] < x1[idx][:, None, None])
              & (tyi[None] >= y0[idx][:, None, None]) & (tyi[None] < y1[idx][:, None, None]))
    valid = inrect & (power <= 0) & (alpha >= A_MIN)
    aeff = torch.where(valid, alpha, torch.zeros_like(alpha))
    one_minus = 1.0 - aeff
    Tafter = torch.cumprod(one_minus, dim=0)
    Tbefore = torch.cat([torch.ones(1, H, W, dtype=dt), Tafter[:-1]], dim=0)
    include = valid & (Tafter >= T_STOP)
    ainc = torch.where(include, alpha, torch.zeros_like(alpha))
    wgt = ainc * Tbefore
    color = torch.einsum("nhw,nc->chw", wgt, rgb[idx])
    final_T = torch.prod(1.0 - ainc, dim=0)
    out = color + final_T[None] * bg[:, None, None]
    n_contrib = torch.zeros(H, W, dtype=torch.int64)
    if idx.numel():
        # 1-based position inside the pixel's TILE list (Gaussians whose rect covers the tile)
        pos = torch.cumsum(inrect.to(torch.int64), dim=0) * include
        n_contrib = pos.max(dim=0).values
    return out, radii, dict(pix=pix, final_T=final_T, n_contrib=n_contrib, order=idx, rgb=rgb,
                            conic=torch.stack([conA, conB, conC], dim=1), cov3D=c6)
