"""The C oracle (fp64 build) against the independent dense autograd
formulation: forward image, radii, final_T, n_contrib, and every gradient."""
import math

import numpy as np
import pytest
import torch

from oracle import torch_ref
from scenes import cloud_numpy, make_scene, oracle_settings


def _dense(cloud, cam, bg, mod, dL, use_shs=True, use_scales=True):
    t = lambda x: x.detach().to(torch.float64).clone().requires_grad_(True)
    m3, op, sc, ro, sh = t(cloud.xyz), t(cloud.opacity), t(cloud.scaling), t(cloud.rotation), t(cloud.features)
    kw = {}
    d = cam.camera_center.to(torch.float64)
    if use_shs:
        kw["shs"] = sh
    else:
        dirs = cloud.xyz.to(torch.float64) - d[None]
        dirs = dirs / dirs.norm(dim=1, keepdim=True)
        col = torch.clamp_min(torch_ref.sh_to_rgb(cloud.sh_degree, cloud.features.to(torch.float64), dirs) + 0.5, 0)
        col = col.detach().clone().requires_grad_(True)
        kw["colors_precomp"] = col
    if use_scales:
        kw["scales"], kw["rotations"] = sc, ro
    else:
        c6 = torch_ref.cov3d(cloud.scaling.to(torch.float64), mod, cloud.rotation.to(torch.float64))
        c6 = c6.detach().clone().requires_grad_(True)
        kw["cov3D_precomp"] = c6
    W, H = cam.image_width, cam.image_height
    out, radii, aux = torch_ref.render_dense(
        W, H, math.tan(cam.FoVx / 2), math.tan(cam.FoVy / 2), bg.to(torch.float64), mod,
        cam.world_view_transform.to(torch.float64), cam.full_proj_transform.to(torch.float64),
        cloud.sh_degree, d, m3, op, **kw)
    (out * torch.from_numpy(dL)).sum().backward()
    g = dict(means3D=m3.grad, opacities=op.grad)
    g["means2D"] = aux["pix"].grad * torch.tensor([0.5 * W, 0.5 * H], dtype=torch.float64)
    if use_shs:
        g["shs"] = sh.grad
    else:
        g["colors_precomp"] = kw["colors_precomp"].grad
    if use_scales:
        g["scales"], g["rotations"] = sc.grad, ro.grad
    else:
        g["cov3D_precomp"] = kw["cov3D_precomp"].grad
    return out.detach().numpy(), radii.numpy(), aux, {k: v.numpy() for k, v in g.items()}


@pytest.mark.parametrize("use_shs,use_scales,deg,bgc", [
    (True, True, 3, (0.3, 0.5, 0.7)),
    (False, False, 2, (0.0, 0.0, 0.0)),
    (True, False, 1, (1.0, 1.0, 1.0)),
    (False, True, 0, (0.2, 0.1, 0.0)),
])
def test_fp64_oracle_matches_dense_autograd(oracle64, use_shs, use_scales, deg, bgc):
    cloud, cam, bg, mod = make_scene(220, 44, 37, sh_degree=deg, seed=3 + deg, bg=bgc, scale_modifier=0.9,
                                     scale_mult=0.8)
    # keep opacity*G under the 0.99 cap away from equality, and z spread wide
    rng = np.random.default_rng(5)
    dL = rng.standard_normal((3, 37, 44))
    ref_img, ref_radii, aux, ref_g = _dense(cloud, cam, bg, mod, dL, use_shs, use_scales)

    s = oracle_settings(cam, bg, deg, mod)
    cn = cloud_numpy(cloud)
    kw = dict(means3D=cn["means3D"], opacities=cn["opacities"])
    if use_shs:
        kw["shs"] = cn["shs"]
    else:
        kw["colors_precomp"] = aux["rgb"].detach().numpy()
    if use_scales:
        kw["scales"], kw["rotations"] = cn["scales"], cn["rotations"]
    else:
        kw["cov3D_precomp"] = aux["cov3D"].detach().numpy()
    st = oracle64.forward(s, **kw)
    assert np.array_equal(st["radii"], ref_radii)
    assert (st["radii"] > 0).sum() > 50
    np.testing.assert_allclose(st["color"], ref_img, rtol=0, atol=1e-10)
    np.testing.assert_allclose(st["fwd"]["final_T"], aux["final_T"].detach().numpy(), atol=1e-12)
    assert np.array_equal(st["fwd"]["n_contrib"].astype(np.int64), aux["n_contrib"].numpy())

    g = oracle64.backward(s, st, dL, cn["means3D"], scales=kw.get("scales"), rotations=kw.get("rotations"),
                          shs=kw.get("shs"))
    for name, ref in ref_g.items():
        got = g[name]
        if name == "means2D":
            got = got[:, :2]
        scale = np.abs(ref).max() + 1e-30
        err = np.abs(got.reshape(ref.shape) - ref).max() / scale
        assert err < 2e-6, (name, err)
    assert np.all(g["means2D"][:, 2] == 0)
