"""Fused Adam step (through the C ABI) against torch.optim.Adam configured as the reference
configures it (scene/gaussian_model.py:154-163), over several steps, six tensors of the model's
shapes with the reference's learning rates; plus ragged sizes and the gradient scale."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

SHAPES = {"xyz": (5003, 3), "f_dc": (5003, 1, 3), "f_rest": (5003, 15, 3), "opacity": (5003, 1), "scaling": (5003, 3),
          "rotation": (5003, 4)}
LRS = {"xyz": 0.00016, "f_dc": 0.0025, "f_rest": 0.0025 / 20.0, "opacity": 0.05, "scaling": 0.005, "rotation": 0.001}


def _rel(a, b):
    return float((a - b).abs().max() / (b.abs().max() + 1e-30))


@pytest.mark.parametrize("grad_scale", [1.0, 0.125])
def test_fused_adam_matches_torch_adam(grad_scale):
    from gsplat_host.optim import GaussianAdam
    g = torch.Generator().manual_seed(7)
    init = {k: torch.randn(*s, generator=g) for k, s in SHAPES.items()}
    ours = {k: v.clone().cuda() for k, v in init.items()}
    ref = {k: torch.nn.Parameter(v.clone().cuda()) for k, v in init.items()}
    opt_ref = torch.optim.Adam([{"params": [ref[k]], "lr": LRS[k], "name": k} for k in SHAPES], lr=0.0, eps=1e-15)
    opt = GaussianAdam(ours, LRS)
    for step in range(1, 8):
        grads = {k: (torch.randn(*s, generator=g) * 10.0 ** (-(step % 5))).cuda() for k, s in SHAPES.items()}
        if step == 3:
            grads["opacity"].zero_()                      # zero gradients: the update is m / (sqrt(v) + 1e-15)
        for k in SHAPES:
            ref[k].grad = grads[k] * grad_scale
        opt_ref.step()
        opt.step(grads, grad_scale=grad_scale)
        torch.cuda.synchronize()
        for k in SHAPES:
            st = opt_ref.state[ref[k]]
            assert _rel(opt.exp_avg[k], st["exp_avg"]) < 2e-6, (step, k)
            assert _rel(opt.exp_avg_sq[k], st["exp_avg_sq"]) < 2e-6, (step, k)
            err = float((ours[k] - ref[k].detach()).abs().max())
            assert err <= 4e-6 * max(1.0, LRS[k] / 1e-3), (step, k, err)
    assert opt.step_count == 7


def test_fused_adam_against_cpu_restatement_and_errors():
    from gsplat_host.optim import GaussianAdam
    from oracle import adam_ref
    g = torch.Generator().manual_seed(1)
    p0, gr = torch.randn(1237, generator=g), torch.randn(1237, generator=g)
    ours = {"a": p0.clone().cuda()}
    opt = GaussianAdam(ours, {"a": 0.01})
    p, m, v = p0.numpy().copy(), np.zeros(1237, np.float32), np.zeros(1237, np.float32)
    for step in range(1, 4):
        opt.step({"a": gr.cuda()})
        adam_ref.adam_step(p, gr.numpy(), m, v, 0.01, step)
    torch.cuda.synchronize()
    assert np.abs(ours["a"].cpu().numpy() - p).max() < 2e-6
    cpu = GaussianAdam({"a": p0.clone()}, {"a": 0.01})         # a CPU tensor can be kept (state bookkeeping) ...
    with pytest.raises(ValueError, match="GPU only"):
        cpu.step({"a": torch.zeros_like(p0)})                   # ... but not stepped: no fallback
    with pytest.raises(ValueError):
        opt.step({"a": gr[:10].cuda()})


def test_grad_arena_feeds_backward_and_step():
    """Gradients of a render land in the arena's flat buffer through .grad views; one fused step
    over them equals torch.optim.Adam on the same gradients."""
    from gsplat_host.cameras import orbit_camera
    from gsplat_host.loss import photometric_loss
    from gsplat_host.model import GaussianParams
    from gsplat_host.optim import GaussianAdam, GradArena
    from gsplat_host.render import PipelineParams, render
    from gsplat_host.synthetic import make_cloud
    dev = torch.device("cuda")
    model = GaussianParams(make_cloud(3000, sh_degree=3, seed=4), device=dev)
    cam = orbit_camera(160, 96, device=dev)
    bg = torch.zeros(3, device=dev)
    gt = torch.rand(3, 96, 160, generator=torch.Generator().manual_seed(2)).to(dev)
    opt = GaussianAdam.for_model(model)
    arena = GradArena(opt.params)
    arena.attach(opt.params)
    out = render(cam, model, PipelineParams(fused_activations=True), bg)
    photometric_loss(out["render"], gt).backward()
    for k, p in opt.params.items():
        assert p.grad.data_ptr() == arena.views[k].data_ptr()      # accumulated in place
        assert float(p.grad.abs().max()) > 0 or k == "f_rest"
    ref = {k: torch.nn.Parameter(p.detach().clone()) for k, p in opt.params.items()}
    for k in ref:
        ref[k].grad = opt.params[k].grad.clone()
    lr = opt.update_learning_rate(1)
    opt_ref = torch.optim.Adam([{"params": [ref[k]], "lr": opt.lrs[k]} for k in ref], lr=0.0, eps=1e-15)
    assert abs(lr - 1.59975441e-04) < 1e-10
    scale = arena.all_reduce()                                   # no process group: identity
    assert scale == 1.0
    opt.step(grad_scale=scale)
    opt_ref.step()
    torch.cuda.synchronize()
    for k in ref:
        assert float((opt.params[k] - ref[k]).abs().max()) <= 4e-6 * max(1.0, opt.lrs[k] / 1e-3), k
