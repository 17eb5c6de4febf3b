"""Generates tests/golden/*.npz by RUNNING the reference's own Python on CPU.

Run once, in the build container, where /root/reference exists:

    python tests/golden/make_golden.py

Only inputs and expected outputs are stored; nothing of the reference's source
travels.  The reference functions used, all importable on CPU:

    utils.sh_utils.eval_sh, RGB2SH, SH2RGB              (utils/sh_utils.py:57-118)
    utils.graphics_utils.getWorld2View2,
        getProjectionMatrix, fov2focal, focal2fov       (utils/graphics_utils.py:38-77)
    utils.loss_utils.l1_loss, ssim                      (utils/loss_utils.py:17-63)
    utils.image_utils.psnr                              (utils/image_utils.py:17-19)
    utils.general_utils.get_expon_lr_func               (utils/general_utils.py:29-62)
    utils.general_utils.build_rotation,
        build_scaling_rotation, strip_symmetric         (utils/general_utils.py:64-110)

The last three allocate with a hard-coded device="cuda"; for the duration of
those calls torch.zeros is wrapped to allocate on the CPU instead (the
arithmetic executed is the reference's, unchanged).  The camera fields
world_view_transform / full_proj_transform / camera_center are derived with the
expressions of scene/cameras.py:54-57 minus the .cuda() calls (scene.cameras
itself needs plyfile, which is not installed).
"""
import math
import os
import sys

import numpy as np
import torch

REF = os.environ.get("GSR_REFERENCE", "/root/reference")
sys.path.insert(0, REF)
HERE = os.path.dirname(os.path.abspath(__file__))

from utils import general_utils, graphics_utils, image_utils, loss_utils, sh_utils  # noqa: E402


def random_rotation(g):
    q = torch.randn(4, generator=g).double()
    q = q / q.norm()
    w, x, y, z = q.tolist()
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y)],
                     [2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x)],
                     [2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)]])


def main():
    torch.manual_seed(0)   # cf. utils/general_utils.py:130-132
    g = torch.Generator().manual_seed(0)
    out = {}

    # 1. SH -> RGB, degrees 0..3, incl. the +0.5 / clamp_min(0) post-step of the renderer
    N = 64
    sh = torch.randn(N, 16, 3, generator=g) * 0.6          # (P,16,3) as get_features
    sh[:, 0] -= 0.9                                         # make a good share clamp at 0
    dirs = torch.randn(N, 3, generator=g)
    dirs = dirs / dirs.norm(dim=1, keepdim=True)
    out["sh_coeffs"] = sh.numpy()
    out["sh_dirs"] = dirs.numpy()
    for deg in range(4):
        shs_view = sh.transpose(1, 2).view(-1, 3, 16)      # gaussian_renderer/__init__.py:74
        raw = sh_utils.eval_sh(deg, shs_view, dirs)
        out[f"sh_raw_deg{deg}"] = raw.numpy()
        out[f"sh_rgb_deg{deg}"] = torch.clamp_min(raw + 0.5, 0.0).numpy()   # :78
    rgb = torch.rand(8, 3, generator=g)
    out["rgb2sh_in"] = rgb.numpy()
    out["rgb2sh_out"] = sh_utils.RGB2SH(rgb).numpy()
    out["sh2rgb_out"] = sh_utils.SH2RGB(sh_utils.RGB2SH(rgb)).numpy()

    # 2. cameras
    cams = []
    for k in range(4):
        R = random_rotation(g)
        T = (torch.randn(3, generator=g) * 2.0).double().numpy()
        trans = np.array([0.0, 0.0, 0.0]) if k < 2 else (torch.randn(3, generator=g)).double().numpy()
        scale = 1.0 if k < 2 else 1.7
        fovx = 0.5 + 0.3 * k
        W, H = (1920, 1080) if k % 2 == 0 else (800, 800)
        fovy = graphics_utils.focal2fov(graphics_utils.fov2focal(fovx, W), H)
        w2v = graphics_utils.getWorld2View2(R, T, trans, scale)
        proj = graphics_utils.getProjectionMatrix(znear=0.01, zfar=100.0, fovX=fovx, fovY=fovy)
        wvt = torch.tensor(w2v).transpose(0, 1)
        pm = proj.transpose(0, 1)
        full = (wvt.unsqueeze(0).bmm(pm.unsqueeze(0))).squeeze(0)
        center = wvt.inverse()[3, :3]
        pts = torch.randn(16, 3, generator=g)
        ndc = graphics_utils.geom_transform_points(pts, full)
        cams.append(dict(R=R, T=T, trans=trans, scale=scale, fovx=fovx, fovy=fovy, W=W, H=H, w2v=w2v,
                         proj=proj.numpy(), wvt=wvt.numpy(), full=full.numpy(), center=center.numpy(),
                         pts=pts.numpy(), ndc=ndc.numpy()))
    for k, c in enumerate(cams):
        for name, v in c.items():
            out[f"cam{k}_{name}"] = np.asarray(v)

    # 3. losses (train.py:91-92: 0.8*L1 + 0.2*(1-ssim)) with autograd gradients
    for tag, (h, w) in {"a": (64, 64), "b": (45, 80)}.items():
        img = torch.rand(3, h, w, generator=g).requires_grad_(True)
        gt = (img.detach() + 0.2 * torch.randn(3, h, w, generator=g)).clamp(0, 1)
        l1 = loss_utils.l1_loss(img, gt)
        ss = loss_utils.ssim(img, gt)
        loss = (1.0 - 0.2) * l1 + 0.2 * (1.0 - ss)
        loss.backward()
        out[f"loss_{tag}_img"] = img.detach().numpy()
        out[f"loss_{tag}_gt"] = gt.numpy()
        out[f"loss_{tag}_l1"] = np.float64(l1.item())
        out[f"loss_{tag}_ssim"] = np.float64(ss.item())
        out[f"loss_{tag}_total"] = np.float64(loss.item())
        out[f"loss_{tag}_grad"] = img.grad.numpy()
        out[f"loss_{tag}_psnr"] = image_utils.psnr(img.detach()[None], gt[None]).numpy()

    # 4. learning-rate schedule (scene/gaussian_model.py:161-164 arguments)
    f = general_utils.get_expon_lr_func(lr_init=1.6e-4, lr_final=1.6e-6, lr_delay_mult=0.01, max_steps=30000)
    steps = np.array([0, 1, 100, 15000, 30000, 40000])
    out["lr_steps"] = steps
    out["lr_values"] = np.array([f(int(s)) for s in steps], dtype=np.float64)

    # 5. 3D covariance from scale + quaternion (scene/gaussian_model.py:27-31)
    scal = torch.exp(torch.randn(32, 3, generator=g) - 3.0)
    rot = torch.randn(32, 4, generator=g)
    rot_n = torch.nn.functional.normalize(rot)            # get_rotation, gaussian_model.py:100-101
    real_zeros = torch.zeros

    def cpu_zeros(*a, **k):
        k.pop("device", None)
        return real_zeros(*a, **k)

    torch.zeros = cpu_zeros
    try:
        for mod in (1.0, 0.6):
            L = general_utils.build_scaling_rotation(mod * scal, rot_n)
            cov = general_utils.strip_symmetric(L @ L.transpose(1, 2))
            out[f"cov_mod{mod}"] = cov.numpy()
        out["cov_rotmat"] = general_utils.build_rotation(rot).numpy()     # normalises internally
    finally:
        torch.zeros = real_zeros
    out["cov_scales"] = scal.numpy()
    out["cov_rot_normalized"] = rot_n.numpy()
    out["cov_rot_raw"] = rot.numpy()

    np.savez_compressed(os.path.join(HERE, "reference_fragments.npz"), **out)
    print("wrote", os.path.join(HERE, "reference_fragments.npz"), len(out), "arrays")


if __name__ == "__main__":
    main()
