"""Parameter store with the reference model's getters.

What the render path reads from ``GaussianModel``
(``scene/gaussian_model.py:95-118``): raw parameters and their activations
(exp for scales, sigmoid for opacity, L2-normalise for rotations, dc|rest
concatenation for SH features, ``:33-41``), plus ``get_covariance``
(``:27-31,117-118``); and the pieces either side of it: initialisation from a
point cloud with the 3-nearest-neighbour scales (``:124-147``), the
densification statistics fed by the rasterizer's screen-space gradient
(``:405-407``, ``train.py:113-116``) and the PLY model file (``:191-256``).
Densification itself (clone / split / prune, optimiser surgery) is host
bookkeeping outside the rasterizer path and is not reproduced.
"""
from __future__ import annotations

import torch
from torch import nn

from .synthetic import Cloud


def inverse_sigmoid(x: torch.Tensor) -> torch.Tensor:
    """utils/general_utils.py:18-19."""
    return torch.log(x / (1 - x))


def quaternion_to_matrix(q: torch.Tensor) -> torch.Tensor:
    """Rotation matrices of normalised (w,x,y,z) quaternions (utils/general_utils.py:78-99)."""
    q = q / q.norm(dim=1, keepdim=True)
    r, x, y, z = q.unbind(dim=1)
    rows = [1 - 2 * (y * y + z * z), 2 * (x * y - r * z), 2 * (x * z + r * y),
            2 * (x * y + r * z), 1 - 2 * (x * x + z * z), 2 * (y * z - r * x),
            2 * (x * z - r * y), 2 * (y * z + r * x), 1 - 2 * (x * x + y * y)]
    return torch.stack(rows, dim=1).reshape(-1, 3, 3)


def covariance_from_scaling_rotation(scaling, scaling_modifier, rotation) -> torch.Tensor:
    """(P,6) packed symmetric covariance (scene/gaussian_model.py:27-31, utils/general_utils.py:64-73)."""
    L = quaternion_to_matrix(rotation) * (scaling_modifier * scaling)[:, None, :]
    cov = L @ L.transpose(1, 2)
    return torch.stack([cov[:, 0, 0], cov[:, 0, 1], cov[:, 0, 2], cov[:, 1, 1], cov[:, 1, 2], cov[:, 2, 2]], dim=1)


class GaussianParams(nn.Module):
    def __init__(self, cloud: Cloud = None, max_sh_degree: int = 3, device="cuda"):
        super().__init__()
        self.max_sh_degree = max_sh_degree
        self.active_sh_degree = 0
        if cloud is None:
            return
        self.active_sh_degree = cloud.sh_degree
        M = (max_sh_degree + 1) ** 2
        feats = torch.zeros(cloud.count, M, 3)
        feats[:, :cloud.features.shape[1]] = cloud.features
        self._set_raw(cloud.xyz, feats[:, 0:1, :], feats[:, 1:, :], inverse_sigmoid(cloud.opacity.clamp(1e-6, 1 - 1e-6)),
                      torch.log(cloud.scaling), cloud.rotation, device)

    def _set_raw(self, xyz, features_dc, features_rest, opacity, scaling, rotation, device):
        p = lambda t: nn.Parameter(t.detach().contiguous().to(device).requires_grad_(True))
        self._xyz = p(xyz)
        self._features_dc = p(features_dc)
        self._features_rest = p(features_rest)
        self._scaling = p(scaling)
        self._rotation = p(rotation)
        self._opacity = p(opacity)
        n = self._xyz.shape[0]
        # training_setup (scene/gaussian_model.py:147,151-152)
        self.max_radii2D = torch.zeros((n,), device=device)
        self.xyz_gradient_accum = torch.zeros((n, 1), device=device)
        self.denom = torch.zeros((n, 1), device=device)

    @classmethod
    def from_point_cloud(cls, points, colors, max_sh_degree: int = 3, device="cuda"):
        """create_from_pcd (scene/gaussian_model.py:124-147): DC colour from RGB, isotropic scales from
        the mean squared distance to the three nearest neighbours (distCUDA2, on the GPU), identity
        rotations, opacity 0.1."""
        from simple_knn._C import distCUDA2
        from .sh import RGB2SH
        m = cls(None, max_sh_degree, device)
        pts = torch.as_tensor(points, dtype=torch.float32).to(device)
        col = RGB2SH(torch.as_tensor(colors, dtype=torch.float32).to(device))
        n = pts.shape[0]
        feats = torch.zeros((n, (max_sh_degree + 1) ** 2, 3), device=device)
        feats[:, 0, :] = col
        dist2 = torch.clamp_min(distCUDA2(pts), 0.0000001)
        scales = torch.log(torch.sqrt(dist2))[..., None].repeat(1, 3)
        rots = torch.zeros((n, 4), device=device)
        rots[:, 0] = 1
        opac = inverse_sigmoid(0.1 * torch.ones((n, 1), dtype=torch.float, device=device))
        m._set_raw(pts, feats[:, 0:1, :], feats[:, 1:, :], opac, scales, rots, device)
        return m

    def add_densification_stats(self, viewspace_point_tensor, update_filter, radii=None):
        """scene/gaussian_model.py:405-407 (and train.py:115 when ``radii`` is given): the norm of the
        screen-space position gradient the rasterizer left in ``viewspace_point_tensor.grad``."""
        if radii is not None:
            self.max_radii2D[update_filter] = torch.max(self.max_radii2D[update_filter], radii[update_filter])
        self.xyz_gradient_accum[update_filter] += torch.norm(viewspace_point_tensor.grad[update_filter, :2], dim=-1,
                                                             keepdim=True)
        self.denom[update_filter] += 1

    def save_ply(self, path: str) -> None:
        from .ply import save_ply
        save_ply(path, self._xyz, self._features_dc, self._features_rest, self._opacity, self._scaling, self._rotation)

    @classmethod
    def load_ply(cls, path: str, max_sh_degree: int = 3, device="cuda"):
        from .ply import load_ply
        t = load_ply(path, max_sh_degree, device)
        m = cls(None, max_sh_degree, device)
        m._set_raw(t["xyz"], t["features_dc"], t["features_rest"], t["opacity"], t["scaling"], t["rotation"], device)
        m.active_sh_degree = max_sh_degree                 # scene/gaussian_model.py:256
        return m

    @property
    def get_xyz(self):
        return self._xyz

    @property
    def get_scaling(self):
        return torch.exp(self._scaling)

    @property
    def get_rotation(self):
        return torch.nn.functional.normalize(self._rotation)

    @property
    def get_opacity(self):
        return torch.sigmoid(self._opacity)

    @property
    def get_features(self):
        return torch.cat((self._features_dc, self._features_rest), dim=1)

    def get_covariance(self, scaling_modifier: float = 1.0):
        return covariance_from_scaling_rotation(self.get_scaling, scaling_modifier, self._rotation)
