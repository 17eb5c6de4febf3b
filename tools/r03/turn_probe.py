#!/usr/bin/env python3
"""What the (wave, splat) turns of the blend backward come to, counted by a library built with
-DGSR_EXPERIMENT_COUNT from blend.hip with tools/experiments/blend_bwd_turn_counters.patch applied (GSR_HIP_LIB points
at it):   python3 tools/r03/turn_probe.py c3 c3w c5"""
import ctypes as C
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
lib = dgr._lib
out = (C.c_ulonglong * 8)()
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
        st = dgr.rasterize_forward(s, model.get_xyz, model.get_opacity, shs=model.get_features,
                                   scales=model.get_scaling, rotations=model.get_rotation)
        lib.gsr_debug_counters(out, 1)
        dgr.blend_backward(st, d)
        lib.gsr_debug_counters(out, 1)
    turns, passed, hits, active, near, fast = (int(out[i]) for i in range(6))
    print(f"{wl}: {turns / 1e6:.2f} M turns, {passed / turns:.1%} pass the vote, {near / turns:.1%} reach a pixel of the "
          f"half tile (active or not); per turn {active / turns:.1f} lanes with an active pixel, {hits / max(passed, 1):.1f} "
          f"lanes hit per passing turn; {fast / turns:.1%} of the turns in the copy without the end checks", flush=True)
    del scene, st, model
    torch.cuda.empty_cache()
