#!/usr/bin/env python3
"""Event-bracketed time of the depth sort alone on the keys of a benchmark scene (median of 40):
    [GSR_HIP_LIB=...] python3 tools/exp_sort.py [c3 c5]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "gaussian-splatting-own_amd"))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import bench  # noqa: E402
import diff_gaussian_rasterization as dgr  # noqa: E402
from tools_common import scene_state  # noqa: E402

dev = torch.device("cuda", 0)
for wl in (sys.argv[1:] or ["c3"]):
    P, keys, sv = scene_state(wl, dev)
    lib, ptr = dgr._lib, dgr._ptr
    out = torch.empty((P,), dtype=torch.int64, device=dev)
    nb = lib.gsr_bin_sort_bytes(P)
    scr = torch.empty((nb,), dtype=torch.uint8, device=dev)
    call = lambda: lib.gsr_bin_sort(P, ptr(keys), ptr(sv), ptr(out), ptr(scr), nb, dgr._stream())
    for _ in range(5):
        call()
    ev = []
    for _ in range(40):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); call(); b.record()
        ev.append((a, b))
    torch.cuda.synchronize()
    ts = sorted(a.elapsed_time(b) for a, b in ev)
    print(f"{wl}: depth sort median {1e3 * ts[20]:.1f} us  min {1e3 * ts[0]:.1f} us  checksum {int(out[:1000].sum())}", flush=True)
