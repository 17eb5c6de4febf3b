"""Seeded synthetic Gaussian clouds for tests and benchmarks.

Tensors have the layouts of the reference's model getters
(``scene/gaussian_model.py:95-118``): xyz (P,3); opacity (P,1) in (0,1);
scaling (P,3) > 0; rotation (P,4) unit quaternions (w,x,y,z); features
(P,16,3) coefficient-major, channel-minor.

Positions follow the reference's own random-cloud recipe, uniform in
[-1.3, 1.3]^3 (``scene/dataset_readers.py:240``); scales are log-normal around
half the mean point spacing (standing in for the 3-nearest-neighbour
initialisation of ``scene/gaussian_model.py:134-135``).
"""
from __future__ import annotations

import math
from dataclasses import dataclass

import torch

SH_C0 = 0.28209479177387814  # utils/sh_utils.py:26


@dataclass
class Cloud:
    xyz: torch.Tensor
    opacity: torch.Tensor
    scaling: torch.Tensor
    rotation: torch.Tensor
    features: torch.Tensor
    sh_degree: int

    def to(self, device):
        return Cloud(self.xyz.to(device), self.opacity.to(device), self.scaling.to(device),
                     self.rotation.to(device), self.features.to(device), self.sh_degree)

    @property
    def count(self) -> int:
        return self.xyz.shape[0]


def median_scale(P: int, extent: float = 2.6) -> float:
    return 0.5 * (extent ** 3 / P) ** (1.0 / 3.0)


def make_cloud(P: int, sh_degree: int = 3, seed: int = 0, skew_fraction: float = 0.0,
               opacity_const: float | None = None, scale_mult: float = 1.0,
               coeffs: int | None = None, anisotropy: float = 1.0) -> Cloud:
    """P random Gaussians.  ``skew_fraction`` of them are drawn N(0, 0.1^2)
    around the origin instead (hot tiles).  ``opacity_const`` replaces the
    sigmoid(N(0,2^2)) opacities by one value (0.1 = the reference's
    initialisation, scene/gaussian_model.py:139: no early termination).
    ``anisotropy`` a stretches the first axis by a and shrinks the other two by a (needles at
    random orientations for a >> 1); applied after all random draws."""
    g = torch.Generator().manual_seed(seed)
    xyz = (torch.rand(P, 3, generator=g) * 2.0 - 1.0) * 1.3
    if skew_fraction > 0.0:
        n = int(P * skew_fraction)
        xyz[:n] = torch.randn(n, 3, generator=g) * 0.1
    s0 = median_scale(P) * scale_mult
    scaling = torch.exp(math.log(s0) + 0.5 * torch.randn(P, 3, generator=g))
    if anisotropy != 1.0:
        scaling = scaling * torch.tensor([anisotropy, 1.0 / anisotropy, 1.0 / anisotropy])
    q = torch.randn(P, 4, generator=g)
    rotation = q / q.norm(dim=1, keepdim=True)
    if opacity_const is None:
        opacity = torch.sigmoid(2.0 * torch.randn(P, 1, generator=g))
    else:
        opacity = torch.full((P, 1), float(opacity_const))
    M = coeffs if coeffs is not None else (sh_degree + 1) ** 2
    features = torch.zeros(P, M, 3)
    features[:, 0, :] = (torch.rand(P, 3, generator=g) - 0.5) / SH_C0   # RGB2SH, utils/sh_utils.py:114
    if M > 1:
        features[:, 1:, :] = 0.05 * torch.randn(P, M - 1, 3, generator=g)
    return Cloud(xyz.contiguous(), opacity.contiguous(), scaling.contiguous(), rotation.contiguous(),
                 features.contiguous(), sh_degree)
