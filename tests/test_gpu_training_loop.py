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


def test_training_through_densification_and_opacity_reset():
    """The reference's schedule in miniature (train.py:109-128): statistics from the screen-space gradients,
    densify_and_prune, reset_opacity, with the fused Adam's state following the changing number of Gaussians."""
    from gsplat_host.cameras import orbit_camera
    from gsplat_host.loss import photometric_loss
    from gsplat_host.model import GaussianParams
    from gsplat_host.optim import GaussianAdam
    from gsplat_host.render import PipelineParams, render
    from gsplat_host.synthetic import make_cloud
    dev = torch.device("cuda", 0)
    truth = GaussianParams(make_cloud(6_000, sh_degree=1, seed=33, scale_mult=1.2), max_sh_degree=1, device=dev)
    cams = [orbit_camera(192, 128, angle_deg=45.0 * k, device=dev) for k in range(8)]
    bg = torch.zeros(3, device=dev)
    pipe = PipelineParams(fused_activations=True)
    with torch.no_grad():
        targets = [render(c, truth, pipe, bg)["render"].clone() for c in cams]
    # a coarser model of the same scene: a third of the Gaussians, larger
    model = GaussianParams(make_cloud(2_000, sh_degree=1, seed=34, scale_mult=2.0), max_sh_degree=1, device=dev)
    model.active_sh_degree = 0
    opt = GaussianAdam.for_model(model)
    model.training_setup(opt, percent_dense=0.01)
    counts, losses = [model._xyz.shape[0]], []
    gen = torch.Generator(device=dev).manual_seed(1)
    for it in range(1, 181):
        opt.update_learning_rate(it)
        if it % 60 == 0:
            model.oneupSHdegree()
        k = it % len(cams)
        out = render(cams[k], model, pipe, bg)
        loss = photometric_loss(out["render"], targets[k])
        loss.backward()
        losses.append(float(loss.detach()))
        with torch.no_grad():
            vis, radii = out["visibility_filter"], out["radii"]
            model.max_radii2D[vis] = torch.max(model.max_radii2D[vis], radii[vis])          # train.py:115
            model.add_densification_stats(out["viewspace_points"], vis)                       # train.py:116
            if it in (60, 120):
                model.densify_and_prune(0.00005, 0.005, 2.6, 40, generator=gen)               # train.py:118-120
                counts.append(model._xyz.shape[0])
            if it == 140:
                model.reset_opacity()                                                         # train.py:122-123
        opt.step()
        opt.zero_grad()
        n = model._xyz.shape[0]
        for name, p in opt.params.items():
            assert p.shape[0] == n and opt.exp_avg[name].shape == p.shape and bool(torch.isfinite(p).all()), name
    print("Gaussians", counts, "loss %.5f -> %.5f" % (sum(losses[:8]) / 8, sum(losses[-8:]) / 8), flush=True)
    assert counts[1] != counts[0] and model.active_sh_degree == 1
    assert opt.step_count == 180 and model.denom.shape[0] == model._xyz.shape[0]
    assert sum(losses[120:128]) / 8 < 0.95 * sum(losses[:8]) / 8         # it has learnt something by the second densification
    assert all(l == l and l < 1.0 for l in losses)


@pytest.mark.parametrize("fused", [True, False])
def test_a_train_step_leaves_nothing_for_the_cyclic_collector(fused):
    """Reference counting alone must free a step's tensors (tile lists, records, accumulators: hundreds of
    megabytes at 1M Gaussians).  With the cyclic collector switched off, the allocated device memory after
    each of several steps is the same, for the raw-parameter path and for the reference's operator API."""
    import gc
    from gsplat_host.cameras import orbit_camera
    from gsplat_host.loss import photometric_loss
    from gsplat_host.model import GaussianParams
    from gsplat_host.render import PipelineParams, render
    from gsplat_host.synthetic import make_cloud
    dev = torch.device("cuda", 0)
    model = GaussianParams(make_cloud(20_000, sh_degree=3, seed=31), max_sh_degree=3, device=dev)
    cam = orbit_camera(320, 200, device=dev)
    bg = torch.zeros(3, device=dev)
    gt = torch.rand((3, 200, 320), device=dev)
    pipe = PipelineParams(fused_activations=fused)

    def step():
        for p in model.parameters():
            p.grad = None
        out = render(cam, model, pipe, bg)
        photometric_loss(out["render"], gt).backward()

    step()
    step()
    torch.cuda.synchronize()
    gc.collect()
    gc.disable()
    try:
        held = []
        for _ in range(4):
            step()
            torch.cuda.synchronize()
            held.append(torch.cuda.memory_allocated())
    finally:
        gc.enable()
    assert len(set(held)) == 1, held


def test_every_render_gets_its_own_gradient_sink():
    """render() hands out a new (P,3) leaf of zeros per call (gaussian_renderer/__init__.py:26-30) on storage that is
    cleared once: two renders in flight keep separate .grad tensors, the values stay zero, and a cloud of another
    size gets a sink of its own size."""
    from gsplat_host.cameras import orbit_camera
    from gsplat_host.model import GaussianParams
    from gsplat_host.render import PipelineParams, render
    from gsplat_host.synthetic import make_cloud
    dev = torch.device("cuda", 0)
    bg = torch.zeros(3, device=dev)
    pipe = PipelineParams(fused_activations=True)
    model = GaussianParams(make_cloud(3_000, sh_degree=1, seed=3), max_sh_degree=1, device=dev)
    a = render(orbit_camera(128, 96, angle_deg=0.0, device=dev), model, pipe, bg)
    b = render(orbit_camera(128, 96, angle_deg=90.0, device=dev), model, pipe, bg)
    sa, sb = a["viewspace_points"], b["viewspace_points"]
    assert sa is not sb and sa.is_leaf and sb.is_leaf and sa.requires_grad and sb.requires_grad
    a["render"].sum().backward()
    assert sa.grad is not None and sb.grad is None
    (2.0 * b["render"]).sum().backward()
    assert sb.grad is not None and sa.grad.data_ptr() != sb.grad.data_ptr()
    assert float(sa.abs().max()) == 0.0 and float(sb.abs().max()) == 0.0
    assert float(sa.grad.abs().max()) > 0.0 and not torch.equal(sa.grad, sb.grad)
    small = GaussianParams(make_cloud(1_000, sh_degree=1, seed=4), max_sh_degree=1, device=dev)
    c = render(orbit_camera(128, 96, angle_deg=0.0, device=dev), small, pipe, bg)
    assert c["viewspace_points"].shape == (1_000, 3) and float(c["viewspace_points"].abs().max()) == 0.0
