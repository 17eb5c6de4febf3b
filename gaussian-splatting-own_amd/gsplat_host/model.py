"""Parameter store with the reference model's getters.

Only what the render path reads from ``GaussianModel``
(``scene/gaussian_model.py:95-118``): raw parameters and their activations
(exp for scales, sigmoid for opacity, L2-normalise for rotations, dc|rest
concatenation for SH features, ``:33-41``), plus ``get_covariance``
(``:27-31,117-118``).  Densification, optimiser surgery and PLY I/O of the
reference's class are outside the rasterizer path and are not reproduced.
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
    def __init__(self, cloud: Cloud, max_sh_degree: int = 3, device="cuda"):
        super().__init__()
        self.max_sh_degree = max_sh_degree
        self.active_sh_degree = cloud.sh_degree
        M = (max_sh_degree + 1) ** 2
        feats = torch.zeros(cloud.count, M, 3)
        feats[:, :cloud.features.shape[1]] = cloud.features
        p = lambda t: nn.Parameter(t.contiguous().to(device).requires_grad_(True))
        self._xyz = p(cloud.xyz)
        self._features_dc = p(feats[:, 0:1, :])
        self._features_rest = p(feats[:, 1:, :])
        self._scaling = p(torch.log(cloud.scaling))
        self._rotation = p(cloud.rotation)
        self._opacity = p(inverse_sigmoid(cloud.opacity.clamp(1e-6, 1 - 1e-6)))

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
