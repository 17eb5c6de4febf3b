#!/usr/bin/env python3
"""How much of a training step is host time: wall-clock per step with and without the per-stage
event brackets, and the host-side time (no device wait) spent inside each C-ABI call and between them."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "gaussian-splatting-own_amd"))
import bench  # noqa: E402
import diff_gaussian_rasterization as dgr  # noqa: E402
from gsplat_host.cameras import orbit_camera  # noqa: E402
from gsplat_host.loss import photometric_loss  # noqa: E402
from gsplat_host.model import GaussianParams  # noqa: E402
from gsplat_host.render import PipelineParams, render  # noqa: E402
from gsplat_host.synthetic import make_cloud  # noqa: E402


def main(workload="c3", steps=30):
    dev = torch.device("cuda", 0)
    P, W, H, deg, skew = bench.WORKLOADS[workload]
    model = GaussianParams(make_cloud(P, sh_degree=deg, seed=0, skew_fraction=skew), max_sh_degree=3, device=dev)
    cam = orbit_camera(W, H, angle_deg=0.0, device=dev)
    bg = torch.zeros(3, device=dev)
    gt = torch.rand(3, H, W, generator=torch.Generator().manual_seed(1)).to(dev)
    pipe = PipelineParams(fused_activations=True)
    params = list(model.parameters())

    def step():
        for p in params:
            p.grad = None
        out = render(cam, model, pipe, bg)
        loss = photometric_loss(out["render"], gt)
        loss.backward()

    def run(n):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            step()
        torch.cuda.synchronize()
        return 1e3 * (time.perf_counter() - t0) / n

    run(5)
    print(f"plain            {run(steps):.4f} ms/step", flush=True)
    dgr.set_stage_timer(dgr.StageTimer())
    print(f"stage brackets   {run(steps):.4f} ms/step", flush=True)
    dgr.set_stage_timer(None)
    print(f"plain again      {run(steps):.4f} ms/step", flush=True)

    # host time of every C-ABI call and of the Python between them
    log = []
    orig = dgr._run

    def logged(what, fn, *a):
        t0 = time.perf_counter()
        orig(what, fn, *a)
        log.append((what, t0, time.perf_counter()))
    dgr._run = logged
    import gsplat_host.loss as L
    if hasattr(L, "_run"):
        L._run = logged
    torch.cuda.synchronize()
    marks = []
    for _ in range(steps):
        marks.append(len(log))
        log.append(("step_begin", time.perf_counter(), time.perf_counter()))
        step()
    torch.cuda.synchronize()
    dgr._run = orig
    names = [w for w, _, _ in log[marks[-1]:]]
    acc = {}
    for m0, m1 in zip(marks[5:], marks[6:] + [len(log)]):
        seg = log[m0:m1]
        for k, (w, a, b) in enumerate(seg):
            nxt = seg[k + 1][1] if k + 1 < len(seg) else None
            d = acc.setdefault((k, w), [0.0, 0.0, 0])
            d[0] += b - a
            d[1] += (nxt - b) if nxt is not None else 0.0
            d[2] += 1
    print(f"{'call':24s} {'inside_us':>10s} {'until_next_us':>14s}")
    for (k, w), (ins, gap, n) in sorted(acc.items()):
        print(f"{w:24s} {1e6 * ins / n:10.1f} {1e6 * gap / n:14.1f}")


if __name__ == "__main__":
    main(*(sys.argv[1:2]))
