"""debug=True (gaussian_renderer/__init__.py:48): same results, and when a stage reports an error the
inputs are saved before the exception goes on.  The failure is forced on the host (a bad argument
that the library rejects before launching anything); no kernel misbehaves."""
import math
import os

import pytest
import torch

from scenes import make_scene

pytestmark = pytest.mark.gpu


def _settings(dgr, cam, bg, debug):
    return dgr.GaussianRasterizationSettings(
        image_height=cam.image_height, image_width=cam.image_width, tanfovx=math.tan(cam.FoVx * 0.5),
        tanfovy=math.tan(cam.FoVy * 0.5), bg=bg.cuda(), scale_modifier=1.0, viewmatrix=cam.world_view_transform,
        projmatrix=cam.full_proj_transform, sh_degree=1, campos=cam.camera_center, prefiltered=False, debug=debug)


def test_debug_mode_gives_the_same_results_and_dumps_on_failure(tmp_path, monkeypatch):
    import diff_gaussian_rasterization as dgr
    cloud, cam, bg, _ = make_scene(2_000, 128, 96, sh_degree=1, seed=2, coeffs=4)
    cam.to("cuda")
    c = cloud.to("cuda")
    out = {}
    for debug in (False, True):
        xyz = c.xyz.clone().requires_grad_(True)
        sink = torch.zeros_like(xyz, requires_grad=True)
        rast = dgr.GaussianRasterizer(raster_settings=_settings(dgr, cam, bg, debug))
        color, radii = rast(means3D=xyz, means2D=sink, shs=c.features, colors_precomp=None, opacities=c.opacity,
                            scales=c.scaling, rotations=c.rotation, cov3D_precomp=None)
        color.square().sum().backward()
        out[debug] = (color.detach().clone(), radii.clone(), xyz.grad.clone())
    assert torch.equal(out[False][0], out[True][0]) and torch.equal(out[False][1], out[True][1])
    # float atomics: the summation order differs from run to run
    assert float((out[False][2] - out[True][2]).abs().max()) <= 2e-5 * float(out[False][2].abs().max())

    # forward failure: SH degree 7 is rejected by the library's argument check
    monkeypatch.chdir(tmp_path)
    bad = _settings(dgr, cam, bg, True)._replace(sh_degree=7)
    rast = dgr.GaussianRasterizer(raster_settings=bad)
    with pytest.raises(RuntimeError, match="SH degree"):
        rast(means3D=c.xyz, means2D=torch.zeros_like(c.xyz), shs=c.features, colors_precomp=None, opacities=c.opacity,
             scales=c.scaling, rotations=c.rotation, cov3D_precomp=None)
    dump = torch.load(os.path.join(tmp_path, "snapshot_fw.dump"), weights_only=True)
    assert torch.equal(dump["means3D"].cpu(), c.xyz.cpu()) and dump["settings"][8] == 7
    # without debug the same call fails the same way and writes nothing
    os.remove(os.path.join(tmp_path, "snapshot_fw.dump"))
    rast = dgr.GaussianRasterizer(raster_settings=bad._replace(debug=False))
    with pytest.raises(RuntimeError, match="SH degree"):
        rast(means3D=c.xyz, means2D=torch.zeros_like(c.xyz), shs=c.features, colors_precomp=None, opacities=c.opacity,
             scales=c.scaling, rotations=c.rotation, cov3D_precomp=None)
    assert not os.path.exists(os.path.join(tmp_path, "snapshot_fw.dump"))
