"""The two formulations of the blend backward (two pixels per lane: the default; one pixel per lane:
GSR_BLEND_BWD=lanes) are independent implementations of the same sums.  The variable is read once per process, so
the other formulation runs in a child process on the same seeded scene; the per-Gaussian sums and every gradient
must agree up to the order of the float additions."""
import os
import subprocess
import sys
import tempfile

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r"""
import sys, numpy as np, torch
sys.path[:0] = [{root!r}, {root!r} + "/gaussian-splatting-own_amd", {root!r} + "/tests"]
from test_gpu_blend_formulations import gradients
np.savez(sys.argv[1], **gradients())
"""


def gradients():
    import diff_gaussian_rasterization as dgr
    from scenes import make_scene
    from test_gpu_parity import _gpu_settings
    out = {}
    for name, kw in (("plain", dict(P=20_000, W=640, H=360)), ("deep", dict(P=8_000, W=333, H=200, opacity_const=0.1)),
                     ("skewed", dict(P=30_000, W=320, H=192, skew_fraction=0.5))):
        kw = dict(kw)
        P, W, H = kw.pop("P"), kw.pop("W"), kw.pop("H")
        cloud, cam, bg, _ = make_scene(P, W, H, sh_degree=3, seed=21, bg=(0.2, 0.1, 0.4), **kw)
        s = _gpu_settings(cam, bg, 3, 1.0)
        c = cloud.to("cuda")
        st = dgr.rasterize_forward(s, c.xyz, c.opacity, shs=c.features, scales=c.scaling, rotations=c.rotation)
        d = torch.randn((3, H, W), device="cuda", generator=torch.Generator(device="cuda").manual_seed(4))
        dsplat = dgr.blend_backward(st, d)
        g = dgr.rasterize_backward(s, st, d, c.xyz, shs=c.features, scales=c.scaling, rotations=c.rotation, dsplat=dsplat)
        out[name + "/sums"] = dsplat[:, :9].cpu().numpy()
        for k in ("means3D", "means2D", "shs", "opacities", "scales", "rotations"):
            out[name + "/" + k] = g[k].cpu().numpy()
    return out


def test_two_pixels_per_lane_equals_one_pixel_per_lane():
    assert os.environ.get("GSR_BLEND_BWD", "") != "lanes", "this test compares the default against GSR_BLEND_BWD=lanes"
    mine = gradients()
    with tempfile.TemporaryDirectory() as tmp:
        path = os.path.join(tmp, "lanes.npz")
        env = dict(os.environ, GSR_BLEND_BWD="lanes")
        subprocess.run([sys.executable, "-c", CHILD.format(root=ROOT), path], check=True, env=env, timeout=300)
        other = dict(np.load(path))
    assert set(other) == set(mine)
    for k in sorted(mine):
        scale = np.abs(other[k]).max() + 1e-20
        err = np.abs(mine[k] - other[k]).max() / scale
        print(k, "%.2e" % err, flush=True)
        assert err < 2e-5, (k, err)
        assert np.abs(mine[k]).max() > 0
