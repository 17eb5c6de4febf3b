"""``render()`` with the reference's signature and return contract
(``gaussian_renderer/__init__.py:18-100``), on top of the MI355X rasterizer.

A maintainer of the reference can keep their own ``gaussian_renderer`` and only
put ``gaussian-splatting-own_amd`` on ``sys.path`` (the drop-in module is named
``diff_gaussian_rasterization``); this function exists so that the tests and
the benchmark drive the rasterizer exactly the way the reference does.
"""
from __future__ import annotations

import math
from dataclasses import dataclass

import torch

from diff_gaussian_rasterization import (GaussianRasterizationSettings, GaussianRasterizer,
                                          rasterize_gaussians_raw)

from .sh import eval_sh


@dataclass
class PipelineParams:
    """arguments/__init__.py:64-69."""
    convert_SHs_python: bool = False
    compute_cov3D_python: bool = False
    debug: bool = False
    # extension (not in the reference): hand the model's raw parameters to the rasterizer and let
    # it apply exp / sigmoid / normalize / dc|rest concatenation, and their backward, in-kernel
    fused_activations: bool = False


_sink_zeros: dict = {}


def _gradient_sink(xyz: torch.Tensor) -> torch.Tensor:
    """A (P,3) leaf of zeros that requires grad: the sink of gaussian_renderer/__init__.py:26-30.  The zeros are
    allocated and cleared once per (shape, device) and every call gets a new leaf on the same storage: nothing
    ever writes the sink's values (the rasterizer only deposits its .grad), so a fill per render would clear
    12 MB at a million Gaussians for nothing."""
    key = (tuple(xyz.shape), xyz.device, xyz.dtype)
    zeros = _sink_zeros.get(key)
    if zeros is None:
        _sink_zeros.clear()                       # one size at a time: densification changes P
        zeros = _sink_zeros[key] = torch.zeros_like(xyz, requires_grad=False)
    return zeros.detach().requires_grad_(True)


def render(viewpoint_camera, pc, pipe: PipelineParams, bg_color: torch.Tensor, scaling_modifier: float = 1.0,
           override_color=None):
    """Returns {"render", "viewspace_points", "visibility_filter", "radii"}.
    ``bg_color`` must be on the GPU, as in the reference (:22)."""
    xyz = pc.get_xyz
    # gradient sink for the 2D means (:26-30); the rasterizer deposits (d/dx, d/dy, 0) in its .grad.
    # The reference writes `zeros_like(...) + 0` and retain_grad(); a leaf that requires grad is the
    # same sink without the extra add kernel and without the copy retain_grad makes in backward.
    screenspace_points = _gradient_sink(xyz)

    settings = GaussianRasterizationSettings(
        image_height=int(viewpoint_camera.image_height), image_width=int(viewpoint_camera.image_width),
        tanfovx=math.tan(viewpoint_camera.FoVx * 0.5), tanfovy=math.tan(viewpoint_camera.FoVy * 0.5),
        bg=bg_color, scale_modifier=scaling_modifier, viewmatrix=viewpoint_camera.world_view_transform,
        projmatrix=viewpoint_camera.full_proj_transform, sh_degree=pc.active_sh_degree,
        campos=viewpoint_camera.camera_center, prefiltered=False, debug=pipe.debug)
    if (pipe.fused_activations and override_color is None and not pipe.convert_SHs_python
            and not pipe.compute_cov3D_python and hasattr(pc, "_features_rest")):
        image, radii, visible = rasterize_gaussians_raw(xyz, screenspace_points, pc._features_dc, pc._features_rest,
                                                        pc._opacity, pc._scaling, pc._rotation, settings)
        return {"render": image, "viewspace_points": screenspace_points, "visibility_filter": visible,
                "radii": radii}
    rasterizer = GaussianRasterizer(raster_settings=settings)

    scales = rotations = cov3D_precomp = None
    if pipe.compute_cov3D_python:
        cov3D_precomp = pc.get_covariance(scaling_modifier)
    else:
        scales, rotations = pc.get_scaling, pc.get_rotation

    shs = colors_precomp = None
    if override_color is not None:
        colors_precomp = override_color
    elif pipe.convert_SHs_python:
        feats = pc.get_features
        shs_view = feats.transpose(1, 2).reshape(-1, 3, (pc.max_sh_degree + 1) ** 2)
        dirs = xyz - viewpoint_camera.camera_center[None, :]
        dirs = dirs / dirs.norm(dim=1, keepdim=True)
        colors_precomp = torch.clamp_min(eval_sh(pc.active_sh_degree, shs_view, dirs) + 0.5, 0.0)
    else:
        shs = pc.get_features

    image, radii = rasterizer(means3D=xyz, means2D=screenspace_points, shs=shs, colors_precomp=colors_precomp,
                              opacities=pc.get_opacity, scales=scales, rotations=rotations,
                              cov3D_precomp=cov3D_precomp)
    return {"render": image, "viewspace_points": screenspace_points, "visibility_filter": radii > 0, "radii": radii}
