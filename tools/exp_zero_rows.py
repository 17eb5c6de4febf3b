#!/usr/bin/env python3
"""How many visible Gaussians receive no gradient at all from the blend backward (an all-zero row of sums)?
    python3 tools/exp_zero_rows.py [c3 c5 c3w c2]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "gaussian-splatting-own_amd"))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import bench  # noqa: E402
import diff_gaussian_rasterization as dgr  # noqa: E402
from gsplat_host.cameras import orbit_camera  # noqa: E402
from gsplat_host.synthetic import make_cloud  # noqa: E402
sys.path.insert(0, os.path.join(ROOT, "tests"))

dev = torch.device("cuda", 0)
for wl in (sys.argv[1:] or ["c3"]):
    sc = bench.Scene(wl, dev, 0)
    from gsplat_host.render import PipelineParams, render
    out = render(sc.cam, sc.model, PipelineParams(fused_activations=True), sc.bg)
    img = out["render"]
    # the state of the forward sits on the autograd node; run the blend backward by hand on a random image gradient
    node = img.grad_fn
    st = node.state
    d = torch.randn_like(img)
    ds = dgr.blend_backward(st, d)
    torch.cuda.synchronize()
    vis = out["radii"] > 0
    zero = (ds[:, :9].abs().sum(dim=1) == 0)
    print(f"{wl}: P {sc.P}  visible {int(vis.sum())}  visible with an all-zero row {int((zero & vis).sum())} "
          f"({100.0 * float((zero & vis).sum()) / float(vis.sum()):.1f} % of the visible)", flush=True)
