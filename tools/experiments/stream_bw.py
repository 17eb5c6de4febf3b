"""What HBM bandwidth do plain streaming kernels of this step's sizes reach?  (torch copy / fill, event-timed)"""
import torch
dev = torch.device("cuda", 0)
def t(fn, reps=30):
    for _ in range(5): fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e-3
for mb in (64, 160, 320, 640, 1280, 2560):
    n = mb * 1024 * 1024 // 4
    x = torch.empty(n, device=dev); y = torch.empty(n, device=dev)
    c = t(lambda: y.copy_(x)); f = t(lambda: y.zero_()); r = t(lambda: x.sum())
    print(f"{mb:5d} MB  copy {2 * mb / 1024 / c / 1e0:7.0f} GB/s ({c * 1e6:6.1f} us)   fill {mb / 1024 / f:7.0f} GB/s ({f * 1e6:6.1f} us)   "
          f"sum {mb / 1024 / r:7.0f} GB/s ({r * 1e6:6.1f} us)", flush=True)
