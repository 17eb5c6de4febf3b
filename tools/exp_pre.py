#!/usr/bin/env python3
"""Event-bracketed time of the raw-parameter preprocess forward alone on a benchmark scene (mean of 30 calls):
    [GSR_HIP_LIB=...] python3 tools/exp_pre.py [c3 c5]"""
import math
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "gaussian-splatting-own_amd"))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import bench  # noqa: E402
import diff_gaussian_rasterization as dgr  # noqa: E402

dev = torch.device("cuda", 0)
for wl in (sys.argv[1:] or ["c3"]):
    sc = bench.Scene(wl, dev, 0)
    m, cam = sc.model, sc.cam
    P, W, H = sc.P, sc.W, sc.H
    lib, ptr = dgr._lib, dgr._ptr
    u32 = dict(dtype=torch.int32, device=dev)
    totals = torch.empty((lib.gsr_totals_words(P),), **u32)
    rec, radii, keys = torch.empty((P, 16), **u32), torch.empty((P,), **u32), torch.empty((P,), **u32)
    sv = torch.empty((P,), dtype=torch.int64, device=dev)
    vis = torch.empty((P,), dtype=torch.bool, device=dev)

    def call():
        rc = lib.gsr_preprocess_fwd_raw(P, sc.deg, 15, ptr(m._xyz), ptr(m._scaling), 1.0, ptr(m._rotation), ptr(m._opacity),
                                        ptr(m._features_dc), ptr(m._features_rest), ptr(cam.world_view_transform),
                                        ptr(cam.full_proj_transform), ptr(cam.camera_center), W, H,
                                        math.tan(cam.FoVx * 0.5), math.tan(cam.FoVy * 0.5), 1, ptr(rec), ptr(radii), None,
                                        ptr(keys), None, ptr(sv), ptr(totals), None, ptr(vis), dgr._stream())
        assert rc == 0, lib.gsr_last_error()
    for _ in range(5):
        call()
    ev = []
    for _ in range(30):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); call(); b.record()
        ev.append((a, b))
    torch.cuda.synchronize()
    ts = sorted(a.elapsed_time(b) for a, b in ev)
    print(f"{wl}: preprocess_fwd_raw (+totals) median {1e3 * ts[15]:.1f} us  min {1e3 * ts[0]:.1f} us", flush=True)
