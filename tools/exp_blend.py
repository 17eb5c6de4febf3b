#!/usr/bin/env python3
"""Kernel-exact timing of the two blend kernels alone on the benchmark's scenes (the launch's own events):
    [GSR_HIP_LIB=/path/to/variant.so] python3 tools/exp_blend.py [c3 c3w c5] [--reps 30]
Prints one line per workload; used to compare variants of the kernels on the same forward state."""
import argparse
import math
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "gaussian-splatting-own_amd"))
sys.path.insert(0, ROOT)

import torch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("workloads", nargs="*", default=["c3"])
    ap.add_argument("--reps", type=int, default=30)
    args = ap.parse_args()
    import diff_gaussian_rasterization as dgr
    from bench import WORKLOADS
    from gsplat_host.cameras import orbit_camera
    from gsplat_host.synthetic import make_cloud
    dev = torch.device("cuda", 0)
    for wl in args.workloads:
        P, W, H, deg, skew, opacity = WORKLOADS[wl]
        c = make_cloud(P, sh_degree=deg, seed=0, skew_fraction=skew, opacity_const=opacity).to(dev)
        cam = orbit_camera(W, H, device=dev)
        s = dgr.GaussianRasterizationSettings(
            image_height=H, image_width=W, tanfovx=math.tan(cam.FoVx * 0.5), tanfovy=math.tan(cam.FoVy * 0.5),
            bg=torch.zeros(3, device=dev), scale_modifier=1.0, viewmatrix=cam.world_view_transform,
            projmatrix=cam.full_proj_transform, sh_degree=deg, campos=cam.camera_center, prefiltered=False, debug=False)
        kw = dict(shs=c.features, scales=c.scaling, rotations=c.rotation)
        d = torch.randn((3, H, W), device=dev, generator=torch.Generator(device=dev).manual_seed(1)) / (3 * H * W)
        for _ in range(3):
            st = dgr.rasterize_forward(s, c.xyz, c.opacity, **kw)
            dgr.blend_backward(st, d)
        timer = dgr.StageTimer(only=(), exact=dgr.StageTimer.EXACT_STAGES)
        dgr.set_stage_timer(timer)
        for _ in range(args.reps):
            st = dgr.rasterize_forward(s, c.xyz, c.opacity, **kw)
            ds = dgr.blend_backward(st, d)
        dgr.set_stage_timer(None)
        ex = timer.exact_ms()
        timer.close()
        chk = float(ds.double().abs().sum())
        print(f"{wl}: instances {st.num_rendered}  " + "  ".join(
            f"{k} mean {v[0] * 1e3:.1f} min {v[1] * 1e3:.1f} max {v[2] * 1e3:.1f} us" for k, v in sorted(ex.items())) +
            f"  |dsplat| {chk:.6e}", flush=True)
        del st, ds, c
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
