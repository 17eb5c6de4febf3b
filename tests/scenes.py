"""Small seeded scenes shared by the tests (numpy views for the oracle, torch
tensors for the HIP path)."""
from __future__ import annotations

import math

import numpy as np
import torch

import oracle
from gsplat_host.cameras import orbit_camera
from gsplat_host.synthetic import make_cloud


def make_scene(P, W, H, sh_degree=3, seed=0, angle=0.0, scale_mult=1.0, bg=(0.0, 0.0, 0.0),
               skew_fraction=0.0, opacity_const=None, coeffs=16, scale_modifier=1.0, distance=4.0, anisotropy=1.0):
    cloud = make_cloud(P, sh_degree=sh_degree, seed=seed, skew_fraction=skew_fraction,
                       opacity_const=opacity_const, scale_mult=scale_mult, coeffs=coeffs, anisotropy=anisotropy)
    cam = orbit_camera(W, H, angle_deg=angle, distance=distance)
    return cloud, cam, torch.tensor(bg, dtype=torch.float32), scale_modifier


def oracle_settings(cam, bg, sh_degree, scale_modifier=1.0):
    return oracle.Settings(
        image_height=cam.image_height, image_width=cam.image_width,
        tanfovx=math.tan(cam.FoVx * 0.5), tanfovy=math.tan(cam.FoVy * 0.5),
        bg=bg.numpy(), scale_modifier=scale_modifier,
        viewmatrix=cam.world_view_transform.numpy(), projmatrix=cam.full_proj_transform.numpy(),
        sh_degree=sh_degree, campos=cam.camera_center.numpy())


def cloud_numpy(cloud):
    return dict(means3D=cloud.xyz.numpy(), opacities=cloud.opacity.numpy(), scales=cloud.scaling.numpy(),
                rotations=cloud.rotation.numpy(), shs=cloud.features.numpy())
