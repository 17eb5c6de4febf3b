"""The numpy restatement of Adam (oracle/adam_ref.py) against torch.optim.Adam configured as the
reference configures it (scene/gaussian_model.py:163), and the learning-rate schedule against the
golden vector made from the reference's get_expon_lr_func."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G = np.load(os.path.join(ROOT, "tests", "golden", "reference_fragments.npz"))


def test_adam_restatement_matches_torch():
    from oracle import adam_ref
    g = torch.Generator().manual_seed(3)
    p0 = torch.randn(1000, 3, generator=g)
    grads = [torch.randn(1000, 3, generator=g) * (10.0 ** float(torch.randint(-6, 1, (1,), generator=g))) for _ in range(6)]
    tp = torch.nn.Parameter(p0.clone())
    opt = torch.optim.Adam([{"params": [tp], "lr": 0.0025}], lr=0.0, eps=1e-15)
    p, m, v = p0.numpy().copy(), np.zeros((1000, 3), np.float32), np.zeros((1000, 3), np.float32)
    for k, gr in enumerate(grads):
        tp.grad = gr.clone()
        opt.step()
        adam_ref.adam_step(p, gr.numpy(), m, v, 0.0025, k + 1)
        st = opt.state[tp]
        assert np.abs(m - st["exp_avg"].numpy()).max() <= 1e-6 * np.abs(m).max()
        assert np.abs(v - st["exp_avg_sq"].numpy()).max() <= 1e-6 * np.abs(v).max()
        assert np.abs(p - tp.detach().numpy()).max() <= 2e-6


def test_lr_schedule_matches_reference_golden():
    sys.path.insert(0, os.path.join(ROOT, "gaussian-splatting-own_amd"))
    # the schedule is plain Python: import the function without the GPU-only module around it
    import ast
    src = open(os.path.join(ROOT, "gaussian-splatting-own_amd", "gsplat_host", "optim.py")).read()
    fn = next(n for n in ast.parse(src).body if isinstance(n, ast.FunctionDef) and n.name == "expon_lr")
    ns = {"math": __import__("math")}
    exec(compile(ast.Module([fn], []), "optim.py", "exec"), ns)
    for s, want in zip(G["lr_steps"], G["lr_values"]):
        got = ns["expon_lr"](int(s), 1.6e-4, 1.6e-6, lr_delay_mult=0.01, max_steps=30000)
        assert abs(got - want) <= 1e-12 * max(1.0, abs(want)) + 1e-18, (s, got, want)
