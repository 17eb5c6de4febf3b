#!/usr/bin/env python3
"""Wall time per training step with the forward as one call and as stage calls, and where the host time of a
step goes (cProfile, device idle: the GPU work of a tiny scene is negligible).   python3 tools/exp_host.py [c2 c3]"""
import cProfile
import os
import pstats
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "gaussian-splatting-own_amd"))
import bench  # noqa: E402
import diff_gaussian_rasterization as dgr  # noqa: E402
from gsplat_host.loss import photometric_loss  # noqa: E402
from gsplat_host.render import PipelineParams, render  # noqa: E402


def main():
    dev = torch.device("cuda", 0)
    for wl in (sys.argv[1:] or ["c2", "c3"]):
        sc = bench.Scene(wl, dev, 0)
        pipe = PipelineParams(fused_activations=True)

        def step():
            for p in sc.params:
                p.grad = None
            out = render(sc.cam, sc.model, pipe, sc.bg)
            photometric_loss(out["render"], sc.gt).backward()

        def run(n):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(n):
                step()
            torch.cuda.synchronize()
            return 1e3 * (time.perf_counter() - t0) / n

        for single in (True, False, True, False, True, False):
            dgr.SINGLE_CALL = single
            run(5)
            print(f"{wl} {'single call' if single else 'stage calls'}: {run(40):.4f} ms/step", flush=True)
        if wl == "c2":
            for single in (True, False):
                dgr.SINGLE_CALL = single
                pr = cProfile.Profile()
                torch.cuda.synchronize()
                pr.enable()
                for _ in range(50):
                    step()
                torch.cuda.synchronize()
                pr.disable()
                print("---- host profile,", "single call" if single else "stage calls", flush=True)
                pstats.Stats(pr).sort_stats("cumulative").print_stats(22)


if __name__ == "__main__":
    main()
