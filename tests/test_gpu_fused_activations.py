"""The raw-parameter entry points (activations folded into the per-Gaussian
kernels) against the reference's way of doing it: PyTorch activations
(scene/gaussian_model.py:95-118) feeding the standard operator, gradients by
autograd through both."""
import numpy as np
import pytest
import torch

from scenes import make_scene

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("P,W,H,deg,max_deg", [(20_000, 320, 200, 3, 3), (5_000, 128, 128, 1, 3), (3_000, 96, 64, 0, 0)])
def test_fused_activations_match_pytorch_activations(P, W, H, deg, max_deg):
    from gsplat_host.model import GaussianParams
    from gsplat_host.render import PipelineParams, render
    cloud, cam, bg, _ = make_scene(P, W, H, sh_degree=deg, seed=21, bg=(0.1, 0.3, 0.2), coeffs=(max_deg + 1) ** 2)
    cam.to("cuda")
    bg = bg.cuda()
    w = torch.randn(3, H, W, generator=torch.Generator().manual_seed(2)).cuda()
    results = []
    for fused in (False, True):
        model = GaussianParams(cloud, max_sh_degree=max_deg, device="cuda")
        with torch.no_grad():       # un-normalised quaternions, so that normalize's backward matters
            model._rotation.mul_(torch.linspace(0.5, 2.0, P, device="cuda")[:, None])
        out = render(cam, model, PipelineParams(fused_activations=fused), bg)
        (out["render"] * w).sum().backward()
        grads = {n: p.grad.clone() for n, p in model.named_parameters() if p.numel() and p.grad is not None}
        assert set(grads) >= {"_xyz", "_features_dc", "_scaling", "_rotation", "_opacity"}
        results.append((out["render"].detach(), out["radii"], out["viewspace_points"].grad.clone(), grads))
    (img_a, radii_a, vs_a, g_a), (img_b, radii_b, vs_b, g_b) = results
    assert float((radii_a != radii_b).float().mean()) < 1e-3          # exp/sigmoid differ by ulps
    same = (radii_a == radii_b)
    diff = (img_a - img_b).abs()
    print("fused vs unfused: image max diff %.2e mean %.2e" % (float(diff.max()), float(diff.mean())), flush=True)
    assert float(diff.mean()) < 1e-5 and float((diff > 1e-3).float().mean()) < 1e-3
    for n in g_a:
        a, b = g_a[n], g_b[n]
        scale = float(a.abs().max()) + 1e-20
        err = float((a - b).abs().max()) / scale
        print("fused vs unfused grad", n, "%.2e" % err, flush=True)
        assert err < 2e-3, (n, err)
        assert float((a - b).abs().mean()) / (float(a.abs().mean()) + 1e-20) < 1e-4, n
    err = float((vs_a - vs_b).abs().max()) / (float(vs_a.abs().max()) + 1e-20)
    assert err < 2e-3
    assert float(vs_b[:, 2].abs().max()) == 0.0
