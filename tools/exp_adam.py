#!/usr/bin/env python3
"""The fused Adam step alone on a model of P Gaussians (median of 20 event-bracketed steps):
    [GSR_HIP_LIB=...] python3 tools/exp_adam.py [P]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "gaussian-splatting-own_amd"))
import torch  # noqa: E402
from gsplat_host.model import GaussianParams  # noqa: E402
from gsplat_host.optim import GaussianAdam  # noqa: E402
from gsplat_host.synthetic import make_cloud  # noqa: E402

P = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
dev = torch.device("cuda", 0)
model = GaussianParams(make_cloud(P, sh_degree=3, seed=0), max_sh_degree=3, device=dev)
opt = GaussianAdam.for_model(model)
for p in model.parameters():
    p.grad = torch.randn_like(p)
for _ in range(3):
    opt.step()
ts = []
for _ in range(20):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(); opt.step(); b.record()
    torch.cuda.synchronize()
    ts.append(a.elapsed_time(b))
ts.sort()
n = sum(p.numel() for p in model.parameters())
print(f"P={P}: adam step median {1e3 * ts[10]:.1f} us  = {28 * n / (ts[10] * 1e-3) / 1e12:.2f} TB/s  checksum {float(model._xyz.detach().sum()):.6f}", flush=True)
