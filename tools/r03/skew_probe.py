#!/usr/bin/env python3
"""How unevenly the tile lists are walked when nothing saturates early: the 6M cloud with half of its Gaussians in a
hot spot, at several constant opacities.  Prints the list statistics bench.py reports and the blend kernels' own
times:   python3 tools/r03/skew_probe.py 0.1 0.02 0.005"""
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
for op in [float(a) for a in sys.argv[1:]] or [0.1]:
    bench.WORKLOADS["probe"] = (6_000_000, 1920, 1080, 3, 0.5, op)
    scene = bench.Scene("probe", dev, 0)
    lists = bench._list_stats(dgr, scene)
    cam, model = scene.cam, scene.model
    s = dgr.GaussianRasterizationSettings(
        image_height=cam.image_height, image_width=cam.image_width, tanfovx=math.tan(cam.FoVx * 0.5),
        tanfovy=math.tan(cam.FoVy * 0.5), bg=scene.bg, scale_modifier=1.0, viewmatrix=cam.world_view_transform,
        projmatrix=cam.full_proj_transform, sh_degree=3, campos=cam.camera_center, prefiltered=False, debug=False)
    d = torch.randn((3, 1080, 1920), device=dev, generator=torch.Generator(device=dev).manual_seed(1)) / (3 * 1080 * 1920)
    with torch.no_grad():
        kw = dict(shs=model.get_features, scales=model.get_scaling, rotations=model.get_rotation)
        xyz, opac = model.get_xyz, model.get_opacity
        for _ in range(2):
            st = dgr.rasterize_forward(s, xyz, opac, **kw)
            dgr.blend_backward(st, d)
        timer = dgr.StageTimer(only=(), exact=dgr.StageTimer.EXACT_STAGES)
        dgr.set_stage_timer(timer)
        for _ in range(5):
            st = dgr.rasterize_forward(s, xyz, opac, **kw)
            dgr.blend_backward(st, d)
        dgr.set_stage_timer(None)
        ex = timer.exact_ms()
        timer.close()
    asked, length, per_tile = dgr.segment_counts(st)
    cut = f"depth segments of {length}: {asked} dumps, {int(per_tile.max())} in one tile  " if length else "whole walks  "
    print(f"opacity {op}: lists {lists}  " + cut + "  ".join(f"{k} {v[0] * 1e3:.0f} us" for k, v in sorted(ex.items())), flush=True)
    del scene, st, model
    torch.cuda.empty_cache()
