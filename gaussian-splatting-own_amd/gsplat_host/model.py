"""Parameter store with the reference model's getters.

What the render path reads from ``GaussianModel``
(``scene/gaussian_model.py:95-118``): raw parameters and their activations
(exp for scales, sigmoid for opacity, L2-normalise for rotations, dc|rest
concatenation for SH features, ``:33-41``), plus ``get_covariance``
(``:27-31,117-118``); and the pieces either side of it: initialisation from a
point cloud with the 3-nearest-neighbour scales (``:124-147``), the
densification statistics fed by the rasterizer's screen-space gradient
(``:405-407``, ``train.py:113-116``), the PLY model file (``:191-256``), and the
host bookkeeping that changes the number of Gaussians between steps
(``:120-122, 179-189, 291-403``: SH degree ramp, opacity reset, and the clone / split /
prune refinement as one pass) with the checkpoint tuple (``:61-93``), all on top of ``GaussianAdam``'s
state surgery.  These are plain PyTorch tensor operations, as in the reference.
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

    # ---- bookkeeping between steps ---------------------------------------------------------------------------
    GROUPS = (("xyz", "_xyz"), ("f_dc", "_features_dc"), ("f_rest", "_features_rest"), ("opacity", "_opacity"),
              ("scaling", "_scaling"), ("rotation", "_rotation"))

    def training_setup(self, optimizer, percent_dense: float = 0.01) -> None:
        """The part of training_setup (scene/gaussian_model.py:149-152) that belongs to the model: the
        optimiser whose state follows the model through densification, ``percent_dense``, fresh statistics."""
        self.optimizer, self.percent_dense = optimizer, percent_dense
        n, dev = self._xyz.shape[0], self._xyz.device
        self.xyz_gradient_accum = torch.zeros((n, 1), device=dev)
        self.denom = torch.zeros((n, 1), device=dev)

    def _adopt(self, tensors) -> None:
        for group, attr in self.GROUPS:
            setattr(self, attr, tensors[group])

    def oneupSHdegree(self) -> None:
        """:120-122."""
        if self.active_sh_degree < self.max_sh_degree:
            self.active_sh_degree += 1

    def reset_opacity(self) -> None:
        """:179-182: no opacity above 0.01; the opacity group's moments start again."""
        new = inverse_sigmoid(torch.min(self.get_opacity.detach(), torch.ones_like(self._opacity) * 0.01))
        self._opacity = self.optimizer.replace_tensor("opacity", new)

    def prune_points(self, mask: torch.Tensor) -> None:
        """:291-305: removes the Gaussians selected by ``mask``."""
        keep = ~mask
        self._adopt(self.optimizer.prune(keep))
        self.xyz_gradient_accum = self.xyz_gradient_accum[keep]
        self.denom = self.denom[keep]
        self.max_radii2D = self.max_radii2D[keep]

    def densification_postfix(self, new_xyz, new_features_dc, new_features_rest, new_opacities, new_scaling,
                              new_rotation) -> None:
        """:329-347: appends Gaussians; the densification statistics of ALL Gaussians start again."""
        self._adopt(self.optimizer.cat_tensors({"xyz": new_xyz, "f_dc": new_features_dc, "f_rest": new_features_rest,
                                                "opacity": new_opacities, "scaling": new_scaling,
                                                "rotation": new_rotation}))
        n, dev = self._xyz.shape[0], self._xyz.device
        self.xyz_gradient_accum = torch.zeros((n, 1), device=dev)
        self.denom = torch.zeros((n, 1), device=dev)
        self.max_radii2D = torch.zeros((n,), device=dev)

    @torch.no_grad()
    def densify_and_prune(self, max_grad, min_opacity, extent, max_screen_size, generator=None, children: int = 2) -> None:
        """The refinement step train.py:118-120 runs every 100 iterations, with the rules of
        scene/gaussian_model.py:349-403 applied in one pass over the current Gaussians:

        * a Gaussian is *restless* when its mean screen-space gradient norm (accumulated norm / times seen, 0 for
          one never seen) reaches ``max_grad``;
        * restless and small (largest scale <= percent_dense * extent): a copy of it is appended;
        * restless and large: ``children`` Gaussians are appended, centred on points drawn from the parent's
          own distribution, with the parent's scales divided by 0.8 * children, and the parent is removed;
        * finally everything with opacity below ``min_opacity`` goes (copies and children included), and, when
          ``max_screen_size`` is given, everything larger than a tenth of the scene or whose recorded screen
          radius exceeds it.  Appending resets the statistics of every Gaussian, so after an append the recorded
          radii are zero.

        Resulting order: the kept originals, the copies, the children (all first children, then all second ones);
        new Gaussians start with zero optimiser moments."""
        dev, count = self._xyz.device, self._xyz.shape[0]
        ratio = (self.xyz_gradient_accum / self.denom).reshape(-1)
        restless = torch.where(ratio.isnan(), torch.zeros_like(ratio), ratio) >= max_grad     # 0 / 0 (never seen) counts as 0
        scale = self.get_scaling
        large = scale.amax(dim=1) > self.percent_dense * extent
        copy_idx = torch.nonzero(restless & ~large).reshape(-1)
        parent_idx = torch.nonzero(restless & large).reshape(-1)
        family = parent_idx.repeat(children)                       # p0 p1 ... p0 p1 ...

        # children: offsets drawn in the parent's own frame, rotated into the world
        spread = scale[family]
        offsets = torch.randn(spread.shape, generator=generator, device=generator.device if generator else dev).to(dev) * spread
        world = torch.einsum("nij,nj->ni", quaternion_to_matrix(self._rotation[family]), offsets)
        born = {"xyz": torch.cat((self._xyz[copy_idx], self._xyz[family] + world)),
                "scaling": torch.cat((self._scaling[copy_idx], torch.log(spread / (0.8 * children))))}
        for group, attr in self.GROUPS:
            if group not in born:
                t = getattr(self, attr)
                born[group] = torch.cat((t[copy_idx], t[family]))
        appended = copy_idx.numel() + family.numel()
        if appended:
            self.densification_postfix(born["xyz"], born["f_dc"], born["f_rest"], born["opacity"], born["scaling"],
                                       born["rotation"])

        total = count + appended
        gone = torch.zeros((total,), dtype=torch.bool, device=dev)
        gone[parent_idx] = True
        gone |= self.get_opacity.reshape(-1) < min_opacity
        if max_screen_size:
            gone |= self.get_scaling.amax(dim=1) > 0.1 * extent
            gone |= self.max_radii2D > max_screen_size          # zeros after an append, as in the reference
        self.prune_points(gone)

    def capture(self):
        """The checkpoint tuple of :61-75, the optimiser's state in torch.optim.Adam's layout."""
        return (self.active_sh_degree, self._xyz, self._features_dc, self._features_rest, self._scaling, self._rotation,
                self._opacity, self.max_radii2D, self.xyz_gradient_accum, self.denom, self.optimizer.state_dict(),
                getattr(self, "spatial_lr_scale", 1.0))

    def restore(self, model_args, optimizer_factory, percent_dense: float = 0.01) -> None:
        """:77-93.  ``optimizer_factory(model)`` builds the optimiser for the restored tensors (the reference
        calls training_setup with its training arguments), then the saved state goes in."""
        (self.active_sh_degree, xyz, f_dc, f_rest, scaling, rotation, opacity, max_radii2D, grad_accum, denom, opt_dict,
         self.spatial_lr_scale) = model_args
        dev = xyz.device
        self._set_raw(xyz, f_dc, f_rest, opacity, scaling, rotation, dev)
        self.max_radii2D = max_radii2D
        self.training_setup(optimizer_factory(self), percent_dense)
        self.xyz_gradient_accum, self.denom = grad_accum, denom
        self.optimizer.load_state_dict(opt_dict)

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
