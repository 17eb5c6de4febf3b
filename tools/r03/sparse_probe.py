#!/usr/bin/env python3
"""Where the sparse per-Gaussian backward spends its time: the C3 workload's real flags and sums, then the same call with
the sums zeroed (look phase only: flags and the rows of flagged Gaussians are read, nothing is computed) and with the
flags zeroed too (flags only).   python3 tools/r03/sparse_probe.py [c3 c3w]"""
import math
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "gaussian-splatting-own_amd"))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import bench  # noqa: E402
import diff_gaussian_rasterization as dgr  # noqa: E402

dev = torch.device("cuda", 0)


def timed(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


for wl in sys.argv[1:] or ["c3"]:
    scene = bench.Scene(wl, dev, 0)
    cam, model = scene.cam, scene.model
    s = dgr.GaussianRasterizationSettings(
        image_height=cam.image_height, image_width=cam.image_width, tanfovx=math.tan(cam.FoVx * 0.5),
        tanfovy=math.tan(cam.FoVy * 0.5), bg=scene.bg, scale_modifier=1.0, viewmatrix=cam.world_view_transform,
        projmatrix=cam.full_proj_transform, sh_degree=3, campos=cam.camera_center, prefiltered=False, debug=False)
    H, W = cam.image_height, cam.image_width
    d = torch.randn((3, H, W), device=dev, generator=torch.Generator(device=dev).manual_seed(1)) / (3 * H * W)
    with torch.no_grad():
        xyz, opac = model.get_xyz, model.get_opacity
        kw = dict(shs=model.get_features, scales=model.get_scaling, rotations=model.get_rotation)
        st = dgr.rasterize_forward(s, xyz, opac, prepare_backward=True, **kw)
        held = st.zeroed_grads
        dsplat = dgr.blend_backward(st, d)
        touched = held.touched
        P = st.P
        active = int((dsplat[:, :9] != 0).any(dim=1).sum())
        print(f"{wl}: P {P}, touched {int(touched.sum())}, with sums {active}", flush=True)
        lib, p = dgr._lib, dgr._ptr
        g = held.views

        def call(ds, tc):
            rc = lib.gsr_preprocess_bwd(P, st.D, st.M, p(xyz), p(kw["scales"]), 1.0, p(kw["rotations"]), p(kw["shs"]), None,
                                        p(st.view), p(st.proj), p(st.campos), W, H, float(s.tanfovx), float(s.tanfovy),
                                        p(st.records), p(ds), p(g["means3D"]), p(g["means2D"]), p(g["opacities"]), p(g["shs"]),
                                        None, p(g["scales"]), p(g["rotations"]), None, 1, p(tc), dgr._stream())
            assert rc == 0
        zeros = torch.zeros_like(dsplat)
        none = torch.zeros_like(touched)
        print(f"  real flags and sums      {timed(lambda: call(dsplat, touched)):7.1f} us")
        print(f"  real flags, zero sums    {timed(lambda: call(zeros, touched)):7.1f} us")
        print(f"  no flags                 {timed(lambda: call(zeros, none)):7.1f} us", flush=True)
    del scene, st, model
    torch.cuda.empty_cache()
