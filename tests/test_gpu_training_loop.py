"""End to end, as train.py strings the pieces together (train.py:76-128): render -> 0.8 L1 + 0.2 (1 - SSIM)
-> backward -> Adam with the reference's learning rates.  A perturbed cloud must move back towards
the images its unperturbed self renders, on both list modes and both activation paths."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _fit(fused: bool, short_lists: bool, steps: int = 60):
    import diff_gaussian_rasterization as dgr
    from gsplat_host.cameras import orbit_camera
    from gsplat_host.loss import photometric_loss
    from gsplat_host.model import GaussianParams
    from gsplat_host.optim import GaussianAdam
    from gsplat_host.render import PipelineParams, render
    from gsplat_host.synthetic import make_cloud
    dev = torch.device("cuda", 0)
    old = dgr.REFERENCE_LISTS
    dgr.REFERENCE_LISTS = not short_lists
    try:
        truth = GaussianParams(make_cloud(4_000, sh_degree=1, seed=21, scale_mult=1.5), max_sh_degree=1, device=dev)
        cams = [orbit_camera(160, 112, angle_deg=60.0 * k, device=dev) for k in range(6)]
        bg = torch.zeros(3, device=dev)
        pipe = PipelineParams(fused_activations=fused)
        with torch.no_grad():
            targets = [render(c, truth, pipe, bg)["render"].clone() for c in cams]
        model = GaussianParams(make_cloud(4_000, sh_degree=1, seed=21, scale_mult=1.5), max_sh_degree=1, device=dev)
        with torch.no_grad():
            g = torch.Generator(device=dev).manual_seed(5)
            model._features_dc.add_(0.3 * torch.randn(model._features_dc.shape, device=dev, generator=g))
            model._opacity.add_(0.5 * torch.randn(model._opacity.shape, device=dev, generator=g))
            model._scaling.add_(0.1 * torch.randn(model._scaling.shape, device=dev, generator=g))
        opt = GaussianAdam.for_model(model)
        losses = []
        for it in range(steps):
            k = it % len(cams)
            out = render(cams[k], model, pipe, bg)
            loss = photometric_loss(out["render"], targets[k])
            loss.backward()
            assert out["viewspace_points"].grad is not None and bool(torch.isfinite(out["viewspace_points"].grad).all())
            opt.update_learning_rate(it + 1)
            opt.step()
            opt.zero_grad()
            losses.append(float(loss.detach()))
        for p in model.parameters():
            assert bool(torch.isfinite(p).all())
        first, last = sum(losses[:6]) / 6, sum(losses[-6:]) / 6
        return first, last
    finally:
        dgr.REFERENCE_LISTS = old


@pytest.mark.parametrize("fused,short_lists", [(False, False), (True, False), (True, True)])
def test_loss_goes_down(fused, short_lists):
    first, last = _fit(fused, short_lists)
    print(f"fused={fused} short_lists={short_lists}: mean loss over the six views {first:.5f} -> {last:.5f}", flush=True)
    assert last < 0.7 * first
