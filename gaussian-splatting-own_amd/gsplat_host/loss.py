"""Photometric loss of the reference's training step on the GPU (train.py:90-92):
``(1 - lambda) * L1 + lambda * (1 - SSIM)`` with lambda_dssim = 0.2
(arguments/__init__.py:85), SSIM as utils/loss_utils.py:33-63.

One fused HIP kernel per direction (``csrc/loss.hip``) behind the C ABI
(``gsr_loss_fwd`` / ``gsr_loss_bwd``); no PyTorch convolutions.  There is no
CPU path here.
"""
from __future__ import annotations

import ctypes as C

import torch

import diff_gaussian_rasterization as _dgr

LAMBDA_DSSIM = 0.2

_lib = _dgr._lib
_lib.gsr_loss_partial_count.restype = C.c_size_t
_lib.gsr_loss_partial_count.argtypes = [C.c_int, C.c_int, C.c_int]
_lib.gsr_loss_fwd.argtypes = [C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_float, C.c_void_p, C.c_void_p,
                              C.c_void_p, C.c_void_p]
_lib.gsr_loss_bwd.argtypes = [C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_float,
                              C.c_void_p, C.c_void_p]


class _PhotometricLoss(torch.autograd.Function):
    @staticmethod
    def forward(ctx, image, gt, lambda_dssim):
        image = _dgr._f32(image, "image")
        gt = _dgr._f32(gt, "gt")
        if image.shape != gt.shape or image.dim() != 3 or image.shape[0] != 3:
            raise ValueError(f"expected two (3,H,W) images, got {tuple(image.shape)} and {tuple(gt.shape)}")
        ch, H, W = image.shape
        dev = image.device
        dmaps = torch.empty((3, ch, H, W), dtype=torch.float32, device=dev)
        partial = torch.empty((_lib.gsr_loss_partial_count(ch, H, W), 2), dtype=torch.float32, device=dev)
        out = torch.empty((3,), dtype=torch.float32, device=dev)
        _dgr._run("loss_fwd", _lib.gsr_loss_fwd, ch, H, W, _dgr._ptr(image), _dgr._ptr(gt), float(lambda_dssim),
                  _dgr._ptr(dmaps), _dgr._ptr(partial), _dgr._ptr(out), _dgr._stream())
        ctx.save_for_backward(image, gt, dmaps)
        ctx.lambda_dssim = float(lambda_dssim)
        ctx.parts = out
        return out[0]

    @staticmethod
    def backward(ctx, grad_out):
        image, gt, dmaps = ctx.saved_tensors
        ch, H, W = image.shape
        grad_out = grad_out.to(torch.float32).reshape(1).contiguous()
        grad = torch.empty_like(image)
        _dgr._run("loss_bwd", _lib.gsr_loss_bwd, ch, H, W, _dgr._ptr(image), _dgr._ptr(gt), _dgr._ptr(dmaps),
                  _dgr._ptr(grad_out), ctx.lambda_dssim, _dgr._ptr(grad), _dgr._stream())
        return grad, None, None


def photometric_loss(image: torch.Tensor, gt: torch.Tensor, lambda_dssim: float = LAMBDA_DSSIM) -> torch.Tensor:
    """Scalar loss tensor on the GPU; differentiable w.r.t. ``image``."""
    return _PhotometricLoss.apply(image, gt, lambda_dssim)


def loss_terms(image: torch.Tensor, gt: torch.Tensor, lambda_dssim: float = LAMBDA_DSSIM) -> torch.Tensor:
    """(loss, L1 mean, SSIM mean) as one 3-element tensor, no gradient."""
    with torch.no_grad():
        ctx = type("Ctx", (), {"save_for_backward": lambda self, *a: None})()
        _PhotometricLoss.forward(ctx, image, gt, lambda_dssim)
        return ctx.parts


def psnr(img1: torch.Tensor, img2: torch.Tensor) -> torch.Tensor:
    """utils/image_utils.py:17-19."""
    mse = ((img1 - img2) ** 2).reshape(img1.shape[0], -1).mean(1, keepdim=True)
    return 20 * torch.log10(1.0 / torch.sqrt(mse))
