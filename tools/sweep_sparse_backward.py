#!/usr/bin/env python3
"""Random scenes through the single call with its prepared backward: the per-Gaussian backward into cleared
gradients (only flagged Gaussians looked at, only ones with sums computed and written) must equal the backward
that writes every row, bit for bit, and the forward's lists and image those of the stage calls.
    python3 tools/sweep_sparse_backward.py [scenes]"""
import os
import random
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "gaussian-splatting-own_amd"))
import math  # noqa: E402

import torch  # noqa: E402
import diff_gaussian_rasterization as dgr  # noqa: E402
from gsplat_host.cameras import orbit_camera  # noqa: E402
from gsplat_host.model import GaussianParams  # noqa: E402
from gsplat_host.synthetic import make_cloud  # noqa: E402

scenes = int(sys.argv[1]) if len(sys.argv) > 1 else 30
dev = torch.device("cuda", 0)
rng = random.Random(7)
for k in range(scenes):
    P = rng.choice([1, 63, 257, 1_000, 5_000, 33_333, 120_000, 300_001])
    W, H = rng.choice([(64, 48), (333, 222), (640, 360), (1280, 720), (1921, 1081)])
    deg = rng.choice([0, 1, 2, 3])
    raw = rng.random() < 0.5
    opacity = rng.choice([None, None, 0.1, 0.9])
    model = GaussianParams(make_cloud(P, sh_degree=deg, seed=100 + k, skew_fraction=rng.choice([0.0, 0.5]), opacity_const=opacity),
                           max_sh_degree=deg, device=dev)
    cam = orbit_camera(W, H, angle_deg=rng.uniform(0, 360), device=dev)
    s = dgr.GaussianRasterizationSettings(
        image_height=int(cam.image_height), image_width=int(cam.image_width), tanfovx=math.tan(cam.FoVx * 0.5),
        tanfovy=math.tan(cam.FoVy * 0.5), bg=torch.rand(3, device=dev), scale_modifier=1.0, viewmatrix=cam.world_view_transform,
        projmatrix=cam.full_proj_transform, sh_degree=deg, campos=cam.camera_center, prefiltered=False, debug=False)
    with torch.no_grad():
        xyz = model.get_xyz.contiguous()
        if raw:
            rawd = dict(features_dc=model._features_dc.detach(), features_rest=model._features_rest.detach() if deg > 0 else None,
                        scaling=model._scaling.detach(), rotation=model._rotation.detach(), opacity=model._opacity.detach())
            fwd = lambda **kw: dgr.rasterize_forward(s, xyz, None, raw=rawd, **kw)
            bwd = lambda st, d, sums: dgr.rasterize_backward_raw(s, st, d, xyz, rawd, dsplat=sums)
        else:
            inp = dict(shs=model.get_features.contiguous(), scales=model.get_scaling.contiguous(),
                       rotations=model.get_rotation.contiguous())
            op = model.get_opacity.contiguous()
            fwd = lambda **kw: dgr.rasterize_forward(s, xyz, op, **inp, **kw)
            bwd = lambda st, d, sums: dgr.rasterize_backward(s, st, d, xyz, dsplat=sums, **inp)
        prepared = fwd(single_call=True, prepare_backward=True)
        staged = fwd(single_call=False)
        assert prepared.num_rendered == staged.num_rendered and torch.equal(prepared.color, staged.color), (k, "forward")
        assert torch.equal(prepared.point_list, staged.point_list) and torch.equal(prepared.ranges, staged.ranges), (k, "lists")
        d = torch.randn_like(prepared.color)
        sums = dgr.blend_backward(prepared, d)
        sparse = bwd(prepared, d, sums)
        dense = bwd(staged, d, sums)
        for name, v in dense.items():
            if v is None:
                assert sparse[name] is None, (k, name)
            else:
                assert torch.equal(sparse[name], v), (k, name, P, W, H, deg, raw)
    torch.cuda.synchronize()
    print(f"scene {k}: P={P} {W}x{H} degree {deg} {'raw' if raw else 'api'} opacity {opacity}: "
          f"{prepared.num_rendered} instances, equal", flush=True)
print("all equal")
