#!/usr/bin/env python3
"""Does an HBM-streaming kernel on a second stream hide beside the depth sort and the list builder (short
dependent launches)?  Times sort + lists of the C3 scene alone, and with a 220 MB read-mostly pass (a torch
sum over 55M floats, or a 110 MB copy) running beside them on another stream."""
import math
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "gaussian-splatting-own_amd"))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import bench  # noqa: E402
import diff_gaussian_rasterization as dgr  # noqa: E402

dev = torch.device("cuda", 0)
sc = bench.Scene("c3", dev, 0)
m, cam = sc.model, sc.cam
s = dgr.GaussianRasterizationSettings(
    image_height=sc.H, image_width=sc.W, tanfovx=math.tan(cam.FoVx * 0.5), tanfovy=math.tan(cam.FoVy * 0.5),
    bg=sc.bg, scale_modifier=1.0, viewmatrix=cam.world_view_transform, projmatrix=cam.full_proj_transform,
    sh_degree=3, campos=cam.camera_center, prefiltered=False, debug=False)
with torch.no_grad():
    args = (m.get_xyz, m.get_opacity)
    kw = dict(shs=m.get_features, scales=m.get_scaling, rotations=m.get_rotation)
st = dgr.rasterize_forward(s, *args, single_call=False, **kw)
torch.cuda.synchronize()
lib, ptr = dgr._lib, dgr._ptr
P, W, H = sc.P, sc.W, sc.H
V, N = st.num_visible, st.num_rendered
# recover E1, E2 through a preprocess call
u32 = dict(dtype=torch.int32, device=dev)
totals = torch.empty((lib.gsr_totals_words(P),), **u32)
rec, radii, keys = torch.empty((P, 16), **u32), torch.empty((P,), **u32), torch.empty((P,), **u32)
sv = torch.empty((P,), dtype=torch.int64, device=dev)
rc = lib.gsr_preprocess_fwd(P, 3, 16, ptr(args[0]), ptr(kw["scales"]), 1.0, ptr(kw["rotations"]), ptr(args[1]), ptr(kw["shs"]),
                            None, None, ptr(st.view), ptr(st.proj), ptr(st.campos), W, H, float(s.tanfovx), float(s.tanfovy), 0, 1,
                            ptr(rec), ptr(radii), None, ptr(keys), None, ptr(sv), ptr(totals), None, None, dgr._stream())
assert rc == 0
n, e1, e2, v = (int(x) for x in totals[:4].tolist())
sorted_vals = torch.empty((P,), dtype=torch.int64, device=dev)
nb = lib.gsr_bin_sort_bytes(P)
scr = torch.empty((nb,), dtype=torch.uint8, device=dev)
nb2 = lib.gsr_bin_lists_bytes(v, e1, e2, W, H)
scr2 = torch.empty((nb2,), dtype=torch.uint8, device=dev)
plist = torch.empty((n,), **u32)
ranges = torch.empty((st.ranges.shape[0], 2), **u32)
big = torch.rand((55_000_000,), device=dev)
dst = torch.empty((27_500_000,), device=dev)
side = torch.cuda.Stream()


def binning():
    lib.gsr_bin_sort(P, ptr(keys), ptr(sv), ptr(sorted_vals), ptr(scr), nb, dgr._stream())
    lib.gsr_bin_lists(P, v, n, e1, e2, W, H, ptr(sorted_vals), ptr(plist), ptr(ranges), None, None, None, None, ptr(scr2), nb2,
                      dgr._stream())


def run(mode, reps=30):
    ts, tside = [], []
    for _ in range(reps):
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        c, d = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        if mode != "alone":
            side.wait_event(a)
            with torch.cuda.stream(side):
                c.record(side)
                if mode == "sum":
                    big.sum()
                else:
                    dst.copy_(big[:27_500_000])
                d.record(side)
        binning()
        b.record()
        torch.cuda.synchronize()
        ts.append(a.elapsed_time(b))
        if mode != "alone":
            tside.append(c.elapsed_time(d))
    ts.sort(); tside.sort()
    print(f"{mode:6s}: sort + lists median {1e3 * ts[len(ts) // 2]:.1f} us" +
          (f"   side kernel median {1e3 * tside[len(tside) // 2]:.1f} us" if tside else ""), flush=True)


for mode in ("alone", "sum", "copy", "alone", "sum", "copy"):
    run(mode)
# the side kernels alone
for mode in ("sum", "copy"):
    ts = []
    for _ in range(20):
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        big.sum() if mode == "sum" else dst.copy_(big[:27_500_000])
        b.record()
        torch.cuda.synchronize()
        ts.append(a.elapsed_time(b))
    ts.sort()
    print(f"{mode} alone: {1e3 * ts[10]:.1f} us", flush=True)
assert torch.equal(plist, st.point_list) and torch.equal(ranges, st.ranges)
