"""The HIP preprocess kernel fed directly with the committed golden inputs of the reference's ``eval_sh``
(tests/golden/reference_fragments.npz, made by running utils/sh_utils.py:57-112 and the +0.5 / clamp step of
gaussian_renderer/__init__.py:74-78 here): the colours it leaves in the splat records, and its clamp flags."""
import math
import os

import numpy as np
import pytest
import torch

from gsplat_host.cameras import orbit_camera

pytestmark = pytest.mark.gpu

G = np.load(os.path.join(os.path.dirname(__file__), "golden", "reference_fragments.npz"))


@pytest.mark.parametrize("deg", [0, 1, 2, 3])
def test_kernel_colours_match_the_reference_eval_sh(deg):
    import diff_gaussian_rasterization as dgr
    sh, dirs = G["sh_coeffs"], G["sh_dirs"]
    n = sh.shape[0]
    # The kernel's view direction is (position - campos) / |.|, and campos is a setting of its own: the Gaussians sit
    # at campos + 0.7 * dir, a unit-sized ball that a camera ten units away sees whole, so none is culled.
    campos = np.array([0.25, -0.5, 0.125], np.float32)
    xyz = torch.from_numpy(campos[None] + np.float32(0.7) * dirs).cuda()
    got_dir = (xyz.cpu().numpy() - campos[None])
    got_dir /= np.linalg.norm(got_dir, axis=1, keepdims=True)
    assert np.abs(got_dir - dirs).max() < 2e-7            # what the kernel will normalise is the golden direction
    cam = orbit_camera(256, 256, distance=10.0)
    s = dgr.GaussianRasterizationSettings(
        image_height=256, image_width=256, tanfovx=math.tan(cam.FoVx * 0.5), tanfovy=math.tan(cam.FoVy * 0.5),
        bg=torch.zeros(3, device="cuda"), scale_modifier=1.0, viewmatrix=cam.world_view_transform.cuda(),
        projmatrix=cam.full_proj_transform.cuda(), sh_degree=deg, campos=torch.from_numpy(campos).cuda(),
        prefiltered=False, debug=False)
    rot = torch.zeros((n, 4), device="cuda")
    rot[:, 0] = 1.0
    st = dgr.rasterize_forward(s, xyz, torch.full((n, 1), 0.5, device="cuda"), shs=torch.from_numpy(sh).cuda(),
                               scales=torch.full((n, 3), 0.02, device="cuda"), rotations=rot)
    torch.cuda.synchronize()
    assert int((st.radii > 0).sum()) == n
    want = G[f"sh_rgb_deg{deg}"]
    rgb = st.rgb.cpu().numpy()
    print(f"degree {deg}: max |kernel colour - reference eval_sh colour| = {np.abs(rgb - want).max():.2e}", flush=True)
    np.testing.assert_allclose(rgb, want, rtol=0, atol=2e-6)
    # clamp flags: set exactly where the reference's clamp_min(. + 0.5, 0) was active (away from the edge itself)
    raw = G[f"sh_raw_deg{deg}"] + np.float32(0.5)
    clear = np.abs(raw) > 1e-5
    flags = st.clamped.cpu().numpy().astype(bool)
    assert (want == 0).sum() > 10 and np.array_equal(flags[clear], (raw < 0)[clear])
    assert (rgb[flags] == 0).all()
