"""Camera matrices in the reference's conventions.

The rasterizer consumes three camera tensors (reference
``gaussian_renderer/__init__.py:43-46``):

* ``world_view_transform``  (4,4): world-to-camera matrix, STORED TRANSPOSED,
  so that a row vector ``[x y z 1] @ world_view_transform`` is the view-space
  point (``scene/cameras.py:54``, ``utils/graphics_utils.py:22-29``);
* ``full_proj_transform``   (4,4): ``world_view_transform @ projection^T``
  (``scene/cameras.py:55-56``);
* ``camera_center``         (3,): last row of the inverse of
  ``world_view_transform`` (``scene/cameras.py:57``).

``R`` is the camera-to-world rotation and ``T`` the world-to-camera
translation, as in the reference's ``Camera(R=..., T=...)``
(``scene/cameras.py:18``, ``utils/graphics_utils.py:38-49``).
"""
from __future__ import annotations

import math

import numpy as np
import torch

ZNEAR = 0.01   # scene/cameras.py:49
ZFAR = 100.0   # scene/cameras.py:48


def fov2focal(fov: float, pixels: float) -> float:
    """utils/graphics_utils.py:73-74."""
    return pixels / (2.0 * math.tan(fov / 2.0))


def focal2fov(focal: float, pixels: float) -> float:
    """utils/graphics_utils.py:76-77."""
    return 2.0 * math.atan(pixels / (2.0 * focal))


def world_to_view(R, T, translate=(0.0, 0.0, 0.0), scale: float = 1.0) -> np.ndarray:
    """World-to-camera 4x4 (NOT transposed), float32; same result as
    ``getWorld2View2`` (utils/graphics_utils.py:38-49), written in closed form:
    the camera centre c = -R t is moved to (c + translate) * scale and the
    translation rebuilt as -R^T c'."""
    R = np.asarray(R, dtype=np.float64)
    t = np.asarray(T, dtype=np.float64)
    centre = (-(R @ t) + np.asarray(translate, dtype=np.float64)) * float(scale)
    m = np.eye(4, dtype=np.float64)
    m[:3, :3] = R.T
    m[:3, 3] = -(R.T @ centre)
    return m.astype(np.float32)


def projection_matrix(znear: float, zfar: float, fovx: float, fovy: float) -> torch.Tensor:
    """Perspective matrix (NOT transposed) with the reference's depth mapping
    (utils/graphics_utils.py:51-71): symmetric frustum, w' = z,
    z' = zfar/(zfar-znear) * z - zfar*znear/(zfar-znear)."""
    thx, thy = math.tan(fovx / 2.0), math.tan(fovy / 2.0)
    right, top = thx * znear, thy * znear
    p = torch.zeros(4, 4)
    p[0, 0] = 2.0 * znear / (2.0 * right)
    p[1, 1] = 2.0 * znear / (2.0 * top)
    p[3, 2] = 1.0
    p[2, 2] = zfar / (zfar - znear)
    p[2, 3] = -(zfar * znear) / (zfar - znear)
    return p


class ViewCamera:
    """The fields of the reference's ``Camera`` / ``MiniCam`` that ``render()``
    reads (scene/cameras.py:17-70), without the image payload."""

    def __init__(self, R, T, fovx: float, fovy: float, width: int, height: int,
                 trans=(0.0, 0.0, 0.0), scale: float = 1.0, device="cpu",
                 znear: float = ZNEAR, zfar: float = ZFAR):
        self.R, self.T = np.asarray(R), np.asarray(T)
        self.FoVx, self.FoVy = float(fovx), float(fovy)
        self.image_width, self.image_height = int(width), int(height)
        self.znear, self.zfar = znear, zfar
        wv = torch.from_numpy(world_to_view(R, T, trans, scale)).transpose(0, 1).contiguous()
        pr = projection_matrix(znear, zfar, self.FoVx, self.FoVy).transpose(0, 1).contiguous()
        self.world_view_transform = wv.to(device)
        self.projection_matrix = pr.to(device)
        self.full_proj_transform = (wv @ pr).to(device)
        self.camera_center = torch.linalg.inv(wv)[3, :3].contiguous().to(device)

    def to(self, device):
        for k in ("world_view_transform", "projection_matrix", "full_proj_transform", "camera_center"):
            setattr(self, k, getattr(self, k).to(device))
        return self


def orbit_camera(width: int, height: int, fovx: float = 0.6911, distance: float = 4.0,
                 angle_deg: float = 0.0, device="cpu") -> ViewCamera:
    """Pinhole camera on a circle of the given radius around the y axis,
    looking at the origin; angle 0 puts it on +z.  FoVy follows from FoVx and
    the aspect ratio through the focal length, as the reference does for
    NeRF-synthetic cameras (scene/dataset_readers.py:212)."""
    th = math.radians(angle_deg)
    centre = np.array([distance * math.sin(th), 0.0, distance * math.cos(th)])
    zc = -centre / np.linalg.norm(centre)              # forward
    yc = np.array([0.0, 1.0, 0.0])
    xc = np.cross(yc, zc)
    xc /= np.linalg.norm(xc)
    R = np.stack([xc, yc, zc], axis=1)                 # camera-to-world, columns = camera axes
    T = -(R.T @ centre)
    fovy = focal2fov(fov2focal(fovx, width), height)
    return ViewCamera(R, T, fovx, fovy, width, height, device=device)
