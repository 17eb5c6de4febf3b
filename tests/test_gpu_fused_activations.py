"""The raw-parameter entry points (activations folded into the per-Gaussian
kernels) against the reference's way of doing it: PyTorch activations
(scene/gaussian_model.py:95-118) feeding the standard operator, gradients by
autograd through both."""
import numpy as np
import pytest
import torch

from scenes import make_scene

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("P,W,H,deg,max_deg", [(20_000, 320, 200, 3, 3), (5_000, 128, 128, 1, 3), (3_000, 96, 64, 0, 0)])
def test_fused_activations_match_pytorch_activations(P, W, H, deg, max_deg):
    from gsplat_host.model import GaussianParams
    from gsplat_host.render import PipelineParams, render
    cloud, cam, bg, _ = make_scene(P, W, H, sh_degree=deg, seed=21, bg=(0.1, 0.3, 0.2), coeffs=(max_deg + 1) ** 2)
    cam.to("cuda")
    bg = bg.cuda()
    w = torch.randn(3, H, W, generator=torch.Generator().manual_seed(2)).cuda()
    results = []
    for fused in (False, True):
        model = GaussianParams(cloud, max_sh_degree=max_deg, device="cuda")
        with torch.no_grad():       # un-normalised quaternions, so that normalize's backward matters
            model._rotation.mul_(torch.linspace(0.5, 2.0, P, device="cuda")[:, None])
        out = render(cam, model, PipelineParams(fused_activations=fused), bg)
        (out["render"] * w).sum().backward()
        grads = {n: p.grad.clone() for n, p in model.named_parameters() if p.numel() and p.grad is not None}
        assert set(grads) >= {"_xyz", "_features_dc", "_scaling", "_rotation", "_opacity"}
        results.append((out["render"].detach(), out["radii"], out["viewspace_points"].grad.clone(), grads))
    (img_a, radii_a, vs_a, g_a), (img_b, radii_b, vs_b, g_b) = results
    assert float((radii_a != radii_b).float().mean()) < 1e-3          # exp/sigmoid differ by ulps
    same = (radii_a == radii_b)
    diff = (img_a - img_b).abs()
    print("fused vs unfused: image max diff %.2e mean %.2e" % (float(diff.max()), float(diff.mean())), flush=True)
    assert float(diff.mean()) < 1e-5 and float((diff > 1e-3).float().mean()) < 1e-3
    for n in g_a:
        a, b = g_a[n], g_b[n]
        scale = float(a.abs().max()) + 1e-20
        err = float((a - b).abs().max()) / scale
        print("fused vs unfused grad", n, "%.2e" % err, flush=True)
        assert err < 2e-3, (n, err)
        assert float((a - b).abs().mean()) / (float(a.abs().mean()) + 1e-20) < 1e-4, n
    err = float((vs_a - vs_b).abs().max()) / (float(vs_a.abs().max()) + 1e-20)
    assert err < 2e-3
    assert float(vs_b[:, 2].abs().max()) == 0.0


def test_single_stream_entry_points_match_the_split_ones():
    """gsr_preprocess_fwd / gsr_preprocess_fwd_raw (geometry + colour on one stream, for hosts that do
    not overlap them) write the same records, keys and list sizes as the module's two-stream path."""
    import ctypes as C
    import math
    import diff_gaussian_rasterization as dgr
    from gsplat_host.model import GaussianParams
    cloud, cam, bg, _ = make_scene(7_000, 200, 136, sh_degree=3, seed=8, bg=(0.0, 0.0, 0.0))
    cam.to("cuda")
    model = GaussianParams(cloud, device="cuda")
    s = dgr.GaussianRasterizationSettings(
        image_height=136, image_width=200, tanfovx=math.tan(cam.FoVx * 0.5), tanfovy=math.tan(cam.FoVy * 0.5),
        bg=bg.cuda(), scale_modifier=1.0, viewmatrix=cam.world_view_transform, projmatrix=cam.full_proj_transform,
        sh_degree=3, campos=cam.camera_center, prefiltered=False, debug=False)
    lib, ptr = dgr._lib, dgr._ptr
    vp, ci, cf = C.c_void_p, C.c_int, C.c_float
    lib.gsr_preprocess_fwd.argtypes = [ci, ci, ci, vp, vp, cf, vp, vp, vp, vp, vp, vp, vp, vp, ci, ci, cf, cf, ci,
                                       vp, vp, vp, vp, vp, vp, vp, vp]
    P = 7000
    stream = dgr._stream()

    def buffers():
        i32 = dict(dtype=torch.int32, device="cuda")
        return dict(records=torch.zeros((P, 16), **i32), radii=torch.zeros((P,), **i32), tiles=torch.zeros((P,), **i32),
                    keys=torch.zeros((P,), **i32), rects=torch.zeros((P, 2), **i32),
                    vals=torch.zeros((P,), dtype=torch.int64, device="cuda"),
                    totals=torch.zeros((lib.gsr_totals_words(P),), **i32))

    with torch.no_grad():
        xyz, op, sc, ro, sh = model.get_xyz, model.get_opacity, model.get_scaling, model.get_rotation, model.get_features
        st = dgr.rasterize_forward(s, xyz, op, shs=sh, scales=sc, rotations=ro)
        b = buffers()
        rc = lib.gsr_preprocess_fwd(P, 3, 16, ptr(xyz), ptr(sc), 1.0, ptr(ro), ptr(op), ptr(sh.contiguous()), None, None,
                                    ptr(st.view), ptr(st.proj), ptr(st.campos), 200, 136, float(s.tanfovx),
                                    float(s.tanfovy), 0, ptr(b["records"]), ptr(b["radii"]), ptr(b["tiles"]),
                                    ptr(b["keys"]), ptr(b["rects"]), ptr(b["vals"]), ptr(b["totals"]), stream)
        assert rc == 0, lib.gsr_last_error()
        torch.cuda.synchronize()
        vis = st.radii > 0
        assert torch.equal(b["radii"], st.radii) and torch.equal(b["keys"], st.depth_keys)
        assert torch.equal(b["records"][vis], st.records[vis]) and torch.equal(b["rects"], st.rect_words)
        assert b["totals"][:4].tolist() == [st.num_rendered, b["totals"][1].item(), b["totals"][2].item(), st.num_visible]

        raw = dict(features_dc=model._features_dc, features_rest=model._features_rest, scaling=model._scaling,
                   rotation=model._rotation, opacity=model._opacity)
        st2 = dgr.rasterize_forward(s, xyz, None, raw=raw)
        b = buffers()
        rc = lib.gsr_preprocess_fwd_raw(P, 3, 15, ptr(xyz), ptr(model._scaling), 1.0, ptr(model._rotation),
                                        ptr(model._opacity), ptr(model._features_dc), ptr(model._features_rest),
                                        ptr(st.view), ptr(st.proj), ptr(st.campos), 200, 136, float(s.tanfovx),
                                        float(s.tanfovy), ptr(b["records"]), ptr(b["radii"]), ptr(b["tiles"]),
                                        ptr(b["keys"]), ptr(b["rects"]), ptr(b["vals"]), ptr(b["totals"]), stream)
        assert rc == 0, lib.gsr_last_error()
        torch.cuda.synchronize()
        vis = st2.radii > 0
        assert torch.equal(b["radii"], st2.radii) and torch.equal(b["records"][vis], st2.records[vis])
        assert int(b["totals"][0]) == st2.num_rendered and int(b["totals"][3]) == st2.num_visible
