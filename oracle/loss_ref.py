"""CPU restatement (PyTorch) of the photometric loss of the reference's training
step -- TEST INFRASTRUCTURE, pinned by tests/golden (values and gradients of
the reference's own l1_loss / ssim).  The product path is the fused HIP kernel
behind gsplat_host.loss.

Loss (train.py:90-92):
``(1 - lambda) * L1 + lambda * (1 - SSIM)`` with lambda_dssim = 0.2
(arguments/__init__.py:85).

SSIM follows utils/loss_utils.py:33-63: 11x11 Gaussian window, sigma 1.5,
zero padding, per-channel (grouped) filtering, C1 = 0.01^2, C2 = 0.03^2, mean
over all elements.  The window is the outer product of a normalised 1-D
Gaussian (loss_utils.py:23-31), so it is applied here as two 1-D passes, which
is the same sum in a different association.
"""
from __future__ import annotations

import math

import torch
import torch.nn.functional as F

WINDOW = 11
SIGMA = 1.5
LAMBDA_DSSIM = 0.2


def l1_loss(x: torch.Tensor, gt: torch.Tensor) -> torch.Tensor:
    return (x - gt).abs().mean()


def _gauss1d(dtype, device) -> torch.Tensor:
    g = torch.tensor([math.exp(-((i - WINDOW // 2) ** 2) / (2.0 * SIGMA ** 2)) for i in range(WINDOW)])
    return (g / g.sum()).to(dtype=dtype, device=device)


def _blur(x: torch.Tensor, g: torch.Tensor) -> torch.Tensor:
    c = x.shape[-3]
    kh = g.view(1, 1, WINDOW, 1).expand(c, 1, WINDOW, 1)
    kw = g.view(1, 1, 1, WINDOW).expand(c, 1, 1, WINDOW)
    x = F.conv2d(x, kh, padding=(WINDOW // 2, 0), groups=c)
    return F.conv2d(x, kw, padding=(0, WINDOW // 2), groups=c)


def ssim(img1: torch.Tensor, img2: torch.Tensor) -> torch.Tensor:
    squeeze = img1.dim() == 3
    if squeeze:
        img1, img2 = img1[None], img2[None]
    g = _gauss1d(img1.dtype, img1.device)
    mu1, mu2 = _blur(img1, g), _blur(img2, g)
    mu1_sq, mu2_sq, mu12 = mu1 * mu1, mu2 * mu2, mu1 * mu2
    s1 = _blur(img1 * img1, g) - mu1_sq
    s2 = _blur(img2 * img2, g) - mu2_sq
    s12 = _blur(img1 * img2, g) - mu12
    c1, c2 = 0.01 ** 2, 0.03 ** 2
    m = ((2 * mu12 + c1) * (2 * s12 + c2)) / ((mu1_sq + mu2_sq + c1) * (s1 + s2 + c2))
    return m.mean()


def photometric_loss(image: torch.Tensor, gt: torch.Tensor, lambda_dssim: float = LAMBDA_DSSIM) -> torch.Tensor:
    """train.py:91-92."""
    return (1.0 - lambda_dssim) * l1_loss(image, gt) + lambda_dssim * (1.0 - ssim(image, gt))


def psnr(img1: torch.Tensor, img2: torch.Tensor) -> torch.Tensor:
    """utils/image_utils.py:17-19."""
    mse = ((img1 - img2) ** 2).reshape(img1.shape[0], -1).mean(1, keepdim=True)
    return 20 * torch.log10(1.0 / torch.sqrt(mse))
