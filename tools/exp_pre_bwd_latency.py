#!/usr/bin/env python3
"""How long does the per-Gaussian backward take when the GPU is nearly empty (the latency of one wave's chain)?
    python3 tools/exp_pre_bwd_latency.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "gaussian-splatting-own_amd"))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import diff_gaussian_rasterization as dgr  # noqa: E402
from gsplat_host.cameras import orbit_camera  # noqa: E402
import math  # noqa: E402
from gsplat_host.synthetic import make_cloud  # noqa: E402

for P in (64, 256, 4096, 65536):
    c = make_cloud(P, sh_degree=3, seed=1).to("cuda")
    cam = orbit_camera(640, 360, device="cuda")
    s = dgr.GaussianRasterizationSettings(
        image_height=360, image_width=640, tanfovx=math.tan(cam.FoVx * 0.5), tanfovy=math.tan(cam.FoVy * 0.5),
        bg=torch.zeros(3, device="cuda"), scale_modifier=1.0, viewmatrix=cam.world_view_transform,
        projmatrix=cam.full_proj_transform, sh_degree=3, campos=cam.camera_center, prefiltered=False, debug=False)
    st = dgr.rasterize_forward(s, c.xyz, c.opacity, shs=c.features, scales=c.scaling, rotations=c.rotation)
    ds = torch.randn((P, 16), device="cuda")          # every Gaussian has sums: all take the long path
    call = lambda: dgr.rasterize_backward(s, st, torch.zeros_like(st.color), c.xyz, shs=c.features, scales=c.scaling,
                                          rotations=c.rotation, dsplat=ds)
    for _ in range(3):
        call()
    ts = []
    for _ in range(20):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        a.record(); call(); b.record()
        torch.cuda.synchronize()
        ts.append(a.elapsed_time(b))
    ts.sort()
    print(f"P={P}: backward call (allocations + one kernel) median {1e3 * ts[10]:.1f} us, min {1e3 * ts[0]:.1f} us", flush=True)
