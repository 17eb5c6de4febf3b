#!/usr/bin/env python3
"""Which objects does a train step leave in reference cycles (freed only by Python's cyclic collector)?
    python3 tools/exp_cycles.py"""
import collections
import gc
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "gaussian-splatting-own_amd"))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import bench  # noqa: E402
from gsplat_host.loss import photometric_loss  # noqa: E402
from gsplat_host.render import PipelineParams, render  # noqa: E402

dev = torch.device("cuda", 0)
sc = bench.Scene("c2", dev, 0)
pipe = PipelineParams(fused_activations="--unfused" not in sys.argv)


def step():
    for p in sc.params:
        p.grad = None
    out = render(sc.cam, sc.model, pipe, sc.bg)
    loss = photometric_loss(out["render"], sc.gt)
    loss.backward()


for _ in range(3):
    step()
torch.cuda.synchronize()
gc.collect()
gc.disable()
before = torch.cuda.memory_allocated()
gc.collect()
for _ in range(5):
    step()
torch.cuda.synchronize()
held = torch.cuda.memory_allocated() - before
gc.set_debug(gc.DEBUG_SAVEALL)
n = gc.collect()
kinds = collections.Counter(type(o).__name__ for o in gc.garbage)
print(f"5 steps without the collector hold {held / 2**20:.1f} MiB; the collector then finds {n} objects:", flush=True)
for k, v in kinds.most_common(25):
    print(f"  {v:5d} {k}")
for o in gc.garbage:
    name = type(o).__name__
    if name in ("function", "cell", "method") or (name not in ("tuple", "dict", "list", "Tensor", "Parameter") and not name.startswith("_")):
        print("   ", name, repr(o)[:160].replace("\n", " "))
