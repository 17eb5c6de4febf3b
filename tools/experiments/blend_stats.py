"""Runs one train step per workload against a library instrumented with iteration counters
(GSR_HIP_LIB must point at it) and prints how many splat iterations the blend kernels start,
how many survive the wave-wide vote, and how many lanes take part in those."""
import ctypes as C
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "gaussian-splatting-own_amd"))
import bench  # noqa: E402
import diff_gaussian_rasterization as dgr  # noqa: E402
from gsplat_host.cameras import orbit_camera  # noqa: E402
from gsplat_host.loss import photometric_loss  # noqa: E402
from gsplat_host.model import GaussianParams  # noqa: E402
from gsplat_host.render import PipelineParams, render  # noqa: E402
from gsplat_host.synthetic import make_cloud  # noqa: E402

for wl in sys.argv[1:] or ["c3"]:
    dev = torch.device("cuda", 0)
    P, W, H, deg, skew, opacity = bench.WORKLOADS[wl]
    model = GaussianParams(make_cloud(P, sh_degree=deg, seed=0, skew_fraction=skew, opacity_const=opacity),
                           max_sh_degree=3, device=dev)
    cam = orbit_camera(W, H, angle_deg=0.0, device=dev)
    bg = torch.zeros(3, device=dev)
    gt = torch.rand(3, H, W, generator=torch.Generator().manual_seed(1)).to(dev)
    out = (C.c_ulonglong * 8)()
    dgr._lib.gsr_blend_stats(out, 1)
    o = render(cam, model, PipelineParams(fused_activations=True), bg)
    photometric_loss(o["render"], gt).backward()
    torch.cuda.synchronize()
    dgr._lib.gsr_blend_stats(out, 0)
    v = list(out)
    tiles = ((W + 15) // 16) * ((H + 15) // 16)
    print(f"{wl}: tiles {tiles}")
    print(f"  fwd staged {v[3]} ({v[3] / tiles:.0f}/tile)  wave iterations {v[0]} ({v[0] / tiles / 4:.0f}/wave)  "
          f"voted in {v[1]} ({100 * v[1] / max(v[0], 1):.1f}%)  lanes per voted-in {v[2] / max(v[1], 1):.1f} of 64")
    print(f"  bwd staged {v[7]} ({v[7] / tiles:.0f}/tile)  wave iterations {v[4]} ({v[4] / tiles / 4:.0f}/wave)  "
          f"voted in {v[5]} ({100 * v[5] / max(v[4], 1):.1f}%)  lanes per voted-in {v[6] / max(v[5], 1):.1f} of 64", flush=True)
