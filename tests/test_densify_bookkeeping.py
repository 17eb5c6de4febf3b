"""Host bookkeeping that changes the number of Gaussians between optimiser steps, against the reference's
own code path: torch.optim.Adam with the optimiser surgery of scene/gaussian_model.py:258-327 restated
literally on a plain torch optimiser.  Runs on the CPU (the fused step itself is a GPU kernel and is tested
there); what is checked here is that parameters, moments, step count and statistics follow each other through
prune / append / replace, and that the state dictionary is the one torch.optim.Adam exchanges."""
import pytest
import torch

from gsplat_host.model import GaussianParams
from gsplat_host.optim import GaussianAdam
from gsplat_host.synthetic import make_cloud

NAMES = ("xyz", "f_dc", "f_rest", "opacity", "scaling", "rotation")


def _model(P=50, deg=2, seed=3):
    m = GaussianParams(make_cloud(P, sh_degree=deg, seed=seed), max_sh_degree=deg, device="cpu")
    opt = GaussianAdam.for_model(m)
    m.training_setup(opt, percent_dense=0.01)
    return m, opt


def _fake_moments(opt, step=7):
    g = torch.Generator().manual_seed(1)
    for n in opt.params:
        opt.exp_avg[n] = torch.randn(opt.params[n].shape, generator=g)
        opt.exp_avg_sq[n] = torch.rand(opt.params[n].shape, generator=g)
    opt.step_count = step


def test_prune_append_replace_keep_rows_aligned():
    m, opt = _model()
    _fake_moments(opt)
    before = {n: (opt.params[n].detach().clone(), opt.exp_avg[n].clone(), opt.exp_avg_sq[n].clone()) for n in NAMES}
    mask = torch.zeros(50, dtype=torch.bool)
    mask[::3] = True
    m.max_radii2D = torch.arange(50.0)
    m.prune_points(mask)
    keep = ~mask
    assert m._xyz.shape[0] == int(keep.sum()) and m._xyz is opt.params["xyz"] and m._features_rest is opt.params["f_rest"]
    for n in NAMES:
        p0, a0, b0 = before[n]
        assert torch.equal(opt.params[n], p0[keep]) and torch.equal(opt.exp_avg[n], a0[keep])
        assert torch.equal(opt.exp_avg_sq[n], b0[keep]) and opt.params[n].requires_grad and opt.params[n].is_leaf
    assert torch.equal(m.max_radii2D, torch.arange(50.0)[keep]) and m.denom.shape == (int(keep.sum()), 1)
    assert opt.step_count == 7

    n0 = m._xyz.shape[0]
    ext = {n: torch.full((4,) + tuple(opt.params[n].shape[1:]), 0.5) for n in NAMES}
    m.densification_postfix(ext["xyz"], ext["f_dc"], ext["f_rest"], ext["opacity"], ext["scaling"], ext["rotation"])
    for n in NAMES:
        assert opt.params[n].shape[0] == n0 + 4
        assert float(opt.exp_avg[n][n0:].abs().max()) == 0.0 and float(opt.exp_avg_sq[n][n0:].abs().max()) == 0.0
        assert torch.equal(opt.exp_avg[n][:n0], before[n][1][keep])
        assert float((opt.params[n].detach()[n0:] - 0.5).abs().max()) == 0.0
    assert float(m.xyz_gradient_accum.abs().max()) == 0.0 and m.max_radii2D.shape == (n0 + 4,)

    m.reset_opacity()
    assert float(torch.sigmoid(m._opacity).max()) <= 0.01 + 1e-6 and m._opacity is opt.params["opacity"]
    assert float(opt.exp_avg["opacity"].abs().max()) == 0.0 and opt.step_count == 7
    assert torch.equal(opt.exp_avg["xyz"][:n0], before["xyz"][1][keep])           # the other groups are untouched


def test_state_dict_is_torch_adams_and_round_trips():
    m, opt = _model()
    _fake_moments(opt, step=12)
    sd = opt.state_dict()
    # a torch.optim.Adam set up as the reference does accepts it, and hands back the same thing
    ref_params = [torch.nn.Parameter(opt.params[n].detach().clone()) for n in NAMES]
    ref = torch.optim.Adam([{"params": [p], "lr": opt.lrs[n], "name": n} for p, n in zip(ref_params, NAMES)],
                           lr=0.0, eps=1e-15)
    ref.load_state_dict(sd)
    for i, n in enumerate(NAMES):
        st = ref.state[ref_params[i]]
        assert torch.equal(st["exp_avg"], opt.exp_avg[n]) and float(st["step"]) == 12.0
        assert ref.param_groups[i]["name"] == n and ref.param_groups[i]["lr"] == pytest.approx(opt.lrs[n])
    back = ref.state_dict()
    m2, opt2 = _model()
    opt2.load_state_dict(back)
    assert opt2.step_count == 12 and opt2.eps == 1e-15
    for n in NAMES:
        assert torch.equal(opt2.exp_avg[n], opt.exp_avg[n]) and torch.equal(opt2.exp_avg_sq[n], opt.exp_avg_sq[n])

    # checkpoint tuple of the model (scene/gaussian_model.py:61-93), through torch.save / weights_only load
    m.xyz_gradient_accum += 1.5
    args = m.capture()
    m3 = GaussianParams(None, max_sh_degree=2, device="cpu")
    m3.restore(args, lambda mm: GaussianAdam.for_model(mm))
    assert m3.active_sh_degree == m.active_sh_degree and torch.equal(m3._scaling, m._scaling)
    assert torch.equal(m3.optimizer.exp_avg_sq["rotation"], opt.exp_avg_sq["rotation"]) and m3.optimizer.step_count == 12
    assert float(m3.xyz_gradient_accum.min()) == 1.5
    with pytest.raises(ValueError):
        opt2.load_state_dict({"state": {}, "param_groups": sd["param_groups"][:3]})


def test_densify_and_prune_matches_the_reference_recipe():
    """Clone the small, split the large, prune the faint: counts and contents against the rules of
    scene/gaussian_model.py:349-403 evaluated independently."""
    m, opt = _model(P=200, deg=1, seed=5)
    _fake_moments(opt, step=3)
    g = torch.Generator().manual_seed(0)
    m.xyz_gradient_accum = torch.rand((200, 1), generator=g) * 0.001
    m.denom = torch.ones((200, 1))
    m.denom[:5] = 0.0                                       # never seen: 0 / 0 -> nan -> 0
    with torch.no_grad():
        m._opacity[10:20] = -8.0                            # faint ones: pruned at the end
    extent, max_grad, min_opacity = 22.0, 0.0005, 0.005
    grads = (m.xyz_gradient_accum / m.denom).nan_to_num(0.0).squeeze()
    scale_max = m.get_scaling.detach().max(dim=1).values
    big = scale_max > m.percent_dense * extent
    clone = (grads >= max_grad) & ~big
    split = (grads >= max_grad) & big
    n_clone, n_split = int(clone.sum()), int(split.sum())
    assert n_clone > 0 and n_split > 0
    faint = torch.sigmoid(m._opacity.detach()).squeeze() < min_opacity
    xyz0, scal0, ea0 = m._xyz.detach().clone(), m._scaling.detach().clone(), opt.exp_avg["xyz"].clone()
    m.densify_and_prune(max_grad, min_opacity, extent, max_screen_size=None, generator=torch.Generator().manual_seed(9))
    # after clone: 200 + n_clone; split appends 2 per selected and removes the selected; then the faint go
    survivors = ~split & ~faint
    n_faint_clones = int((clone & faint).sum())
    n_faint_children = 2 * int((split & faint).sum())
    want = int(survivors.sum()) + (n_clone - n_faint_clones) + (2 * n_split - n_faint_children)
    assert m._xyz.shape[0] == want == opt.params["xyz"].shape[0] == opt.exp_avg["f_rest"].shape[0]
    k = int(survivors.sum())
    assert torch.equal(m._xyz.detach()[:k], xyz0[survivors]) and torch.equal(opt.exp_avg["xyz"][:k], ea0[survivors])
    assert float(opt.exp_avg["xyz"][k:].abs().max()) == 0.0          # every new Gaussian starts with zero moments
    # the children of a split are smaller by 0.8 * 2
    children = m._scaling.detach()[m._xyz.shape[0] - (2 * n_split - n_faint_children):]
    parents = scal0[split & ~faint].repeat(2, 1)
    assert torch.allclose(torch.exp(children), torch.exp(parents) / 1.6, rtol=1e-5)
    assert opt.step_count == 3 and m.denom.shape[0] == want and float(m.denom.abs().max()) == 0.0


def test_the_fused_step_refuses_cpu_tensors():
    m, opt = _model()
    with pytest.raises(ValueError, match="GPU only"):
        opt.step(grads={n: torch.zeros_like(p) for n, p in opt.params.items()})
