"""Fused L1 + SSIM loss kernels (through the C ABI) against the golden vectors
made from the reference's l1_loss / ssim, and against the pinned CPU
restatement at sizes the goldens do not cover."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

G = np.load(os.path.join(os.path.dirname(__file__), "golden", "reference_fragments.npz"))


@pytest.mark.parametrize("tag", ["a", "b"])
def test_fused_loss_matches_reference_golden(tag):
    from gsplat_host import loss
    img = torch.from_numpy(G[f"loss_{tag}_img"]).cuda().requires_grad_(True)
    gt = torch.from_numpy(G[f"loss_{tag}_gt"]).cuda()
    terms = loss.loss_terms(img.detach(), gt).cpu().numpy()
    assert abs(terms[1] - float(G[f"loss_{tag}_l1"])) < 1e-6
    assert abs(terms[2] - float(G[f"loss_{tag}_ssim"])) < 2e-6
    total = loss.photometric_loss(img, gt)
    assert abs(total.item() - float(G[f"loss_{tag}_total"])) < 2e-6
    (3.0 * total).backward()
    ref = 3.0 * G[f"loss_{tag}_grad"]
    err = np.abs(img.grad.cpu().numpy() - ref).max() / np.abs(ref).max()
    print(tag, "fused loss grad rel err %.2e" % err, flush=True)
    assert err < 1e-5


@pytest.mark.parametrize("shape", [(37, 53), (270, 480), (1080, 1920)])
def test_fused_loss_matches_cpu_restatement(shape):
    from gsplat_host import loss
    from oracle import loss_ref
    H, W = shape
    g = torch.Generator().manual_seed(H)
    img = torch.rand(3, H, W, generator=g)
    gt = (img + 0.3 * torch.randn(3, H, W, generator=g)).clamp(0, 1)
    a = img.clone().requires_grad_(True)
    want = loss_ref.photometric_loss(a, gt)
    want.backward()
    b = img.cuda().requires_grad_(True)
    got = loss.photometric_loss(b, gt.cuda())
    got.backward()
    assert abs(got.item() - want.item()) < 3e-6
    ref = a.grad.numpy()
    err = np.abs(b.grad.cpu().numpy() - ref).max() / np.abs(ref).max()
    print(shape, "loss %.6f grad rel err %.2e" % (got.item(), err), flush=True)
    assert err < 2e-5
    # identical images: SSIM = 1, L1 = 0
    t = loss.loss_terms(gt.cuda(), gt.cuda()).cpu().numpy()
    assert abs(t[0]) < 1e-6 and abs(t[2] - 1.0) < 1e-6
