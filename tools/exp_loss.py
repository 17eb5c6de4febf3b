#!/usr/bin/env python3
"""Time of the fused loss kernels alone (forward + finalize, backward) on random images, 20 calls between two events:
    [GSR_HIP_LIB=...] python3 tools/exp_loss.py [H W]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "gaussian-splatting-own_amd"))
import torch  # noqa: E402
import diff_gaussian_rasterization as dgr  # noqa: E402
import gsplat_host.loss  # noqa: E402,F401  (declares the argument types of the loss entry points)

H, W = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (1080, 1920)
dev = torch.device("cuda", 0)
lib, ptr = dgr._lib, dgr._ptr
g = torch.Generator(device=dev).manual_seed(1)
img = torch.rand((3, H, W), device=dev, generator=g)
gt = torch.rand((3, H, W), device=dev, generator=g)
dmaps = torch.empty((3, 3, H, W), device=dev)
partial = torch.empty((2 * lib.gsr_loss_partial_count(3, H, W),), device=dev)
out3 = torch.empty((3,), device=dev)
go = torch.ones((1,), device=dev)
grad = torch.empty_like(img)


def fwd():
    assert lib.gsr_loss_fwd(3, H, W, ptr(img), ptr(gt), 0.2, ptr(dmaps), ptr(partial), ptr(out3), dgr._stream()) == 0


def bwd():
    assert lib.gsr_loss_bwd(3, H, W, ptr(img), ptr(gt), ptr(dmaps), ptr(go), 0.2, ptr(grad), dgr._stream()) == 0


for name, fn in (("loss_fwd + finalize", fwd), ("loss_bwd", bwd)):
    for _ in range(5):
        fn()
    ts = []
    for _ in range(15):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(20):
            fn()
        b.record()
        torch.cuda.synchronize()
        ts.append(a.elapsed_time(b) / 20)
    ts.sort()
    print(f"{H}x{W} {name}: median {1e3 * ts[7]:.1f} us  min {1e3 * ts[0]:.1f} us   loss {out3[0].item():.6f}", flush=True)
