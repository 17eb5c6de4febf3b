"""How many tile instances would a tile rectangle cut down to the bounding box of the alpha >= 1/255
ellipse (the half extents in record words 2, 3) leave, against the reference's 3-sigma rectangle?"""
import math
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "gaussian-splatting-own_amd"))
import bench  # noqa: E402
import diff_gaussian_rasterization as dgr  # noqa: E402
from gsplat_host.cameras import orbit_camera  # noqa: E402
from gsplat_host.model import GaussianParams  # noqa: E402
from gsplat_host.synthetic import make_cloud  # noqa: E402

for wl in sys.argv[1:] or ["c3"]:
    dev = torch.device("cuda", 0)
    P, W, H, deg, skew = bench.WORKLOADS[wl]
    model = GaussianParams(make_cloud(P, sh_degree=deg, seed=0, skew_fraction=skew), max_sh_degree=3, device=dev)
    cam = orbit_camera(W, H, angle_deg=0.0, device=dev)
    with torch.no_grad():
        s = dgr.GaussianRasterizationSettings(
            image_height=H, image_width=W, tanfovx=math.tan(cam.FoVx * 0.5), tanfovy=math.tan(cam.FoVy * 0.5),
            bg=torch.zeros(3, device=dev), scale_modifier=1.0, viewmatrix=cam.world_view_transform,
            projmatrix=cam.full_proj_transform, sh_degree=model.active_sh_degree, campos=cam.camera_center,
            prefiltered=False, debug=False)
        st = dgr.rasterize_forward(s, model.get_xyz, model.get_opacity, shs=model.get_features,
                                   scales=model.get_scaling, rotations=model.get_rotation)
        rec = st.records.view(torch.float32)
        vis = st.radii > 0
        x, y, hx, hy = rec[vis, 0], rec[vis, 1], rec[vis, 2], rec[vis, 3]
        r = st.rects[vis].to(torch.float32)
        never = hx < 0
        tx0 = torch.clamp(torch.floor((x - hx) / 16), min=0)
        tx1 = torch.floor((x + hx) / 16) + 1
        ty0 = torch.clamp(torch.floor((y - hy) / 16), min=0)
        ty1 = torch.floor((y + hy) / 16) + 1
        ax0, ax1 = torch.maximum(tx0, r[:, 0]), torch.minimum(tx1, r[:, 2])
        ay0, ay1 = torch.maximum(ty0, r[:, 1]), torch.minimum(ty1, r[:, 3])
        n_tight = (torch.clamp(ax1 - ax0, min=0) * torch.clamp(ay1 - ay0, min=0))
        n_tight = torch.where(never, torch.zeros_like(n_tight), n_tight)
        n_full = (r[:, 2] - r[:, 0]) * (r[:, 3] - r[:, 1])
        print(f"{wl}: visible {int(vis.sum())}  instances {int(n_full.sum())} -> {int(n_tight.sum())} "
              f"({100 * float(n_tight.sum()) / float(n_full.sum()):.1f} %), never-visible splats {int(never.sum())}, "
              f"splats with empty tight rect {int((n_tight == 0).sum())}", flush=True)
