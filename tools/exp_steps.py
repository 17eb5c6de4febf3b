#!/usr/bin/env python3
"""Wall time of consecutive train steps of a benchmark scene, in groups of 10 (is the step time steady?):
    python3 tools/exp_steps.py [workload] [steps] [--no-gc]"""
import gc
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "gaussian-splatting-own_amd"))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import bench  # noqa: E402
import diff_gaussian_rasterization as dgr  # noqa: E402,F401
from gsplat_host.loss import photometric_loss  # noqa: E402
from gsplat_host.render import PipelineParams, render  # noqa: E402

wl = sys.argv[1] if len(sys.argv) > 1 and not sys.argv[1].startswith("-") else "c3"
steps = int(sys.argv[2]) if len(sys.argv) > 2 and sys.argv[2].isdigit() else 300
if "--no-gc" in sys.argv:
    gc.disable()
dev = torch.device("cuda", 0)
sc = bench.Scene(wl, dev, 0)
pipe = PipelineParams(fused_activations=True)


def step():
    for p in sc.params:
        p.grad = None
    out = render(sc.cam, sc.model, pipe, sc.bg)
    loss = photometric_loss(out["render"], sc.gt)
    loss.backward()


for _ in range(10):
    step()
torch.cuda.synchronize()
marks = [time.perf_counter()]
for i in range(steps):
    step()
    if i % 10 == 9:
        torch.cuda.synchronize()
        marks.append(time.perf_counter())
groups = [1e3 * (b - a) / 10 for a, b in zip(marks, marks[1:])]
print(wl, "ms per step in groups of 10:", " ".join(f"{g:.3f}" for g in groups), flush=True)
print("allocator:", {k: v for k, v in torch.cuda.memory_stats().items() if k in ("num_alloc_retries", "num_device_alloc", "num_device_free", "reserved_bytes.all.peak")}, flush=True)
