"""HIP kernels (through the C ABI, via the drop-in module) against the fp32 CPU
oracle on the same seeded inputs.

Bars: per-Gaussian preprocess outputs, sort keys / values and tile ranges are
BIT-EXACT; images within 1e-5 absolute per pixel, except pixels the oracle
flags as sitting on a discontinuous threshold (alpha = 1/255, T = 1e-4,
power = 0) within fp32 rounding, whose share is bounded; gradients within a
relative tolerance (their sums run in a different order on the GPU).
"""
import math

import numpy as np
import pytest
import torch

from scenes import cloud_numpy, make_scene, oracle_settings

pytestmark = pytest.mark.gpu

PIX_TOL = 1e-5          # north-star tolerance: 1e-5 fp32 per pixel
MARGIN_SAFE = 4.0       # oracle margin below which a threshold decision may flip in fp32


def _gpu_settings(cam, bg, deg, mod, debug=False):
    import diff_gaussian_rasterization as dgr
    dev = "cuda"
    return dgr.GaussianRasterizationSettings(
        image_height=cam.image_height, image_width=cam.image_width,
        tanfovx=math.tan(cam.FoVx * 0.5), tanfovy=math.tan(cam.FoVy * 0.5), bg=bg.to(dev),
        scale_modifier=mod, viewmatrix=cam.world_view_transform.to(dev), projmatrix=cam.full_proj_transform.to(dev),
        sh_degree=deg, campos=cam.camera_center.to(dev), prefiltered=False, debug=debug)


def _bits(x):
    return np.ascontiguousarray(x, dtype=np.float32).view(np.uint32)


CASES = {
    # name: (P, W, H, degree, coeffs, bg, scale_modifier, kwargs)
    "c1_10k_256_deg0": (10_000, 256, 256, 0, 1, (0.0, 0.0, 0.0), 1.0, {}),
    "ragged_5k_333x200_deg3": (5_000, 333, 200, 3, 16, (0.2, 0.4, 0.9), 0.8, {}),
    "deg2_of_16": (8_000, 400, 240, 2, 16, (1.0, 1.0, 1.0), 1.0, {}),
    "deg1_close_camera": (6_000, 320, 320, 1, 16, (0.0, 0.0, 0.0), 1.0, dict(distance=1.6, scale_mult=2.0)),
    "skewed_hot_tiles": (30_000, 320, 192, 3, 16, (0.1, 0.1, 0.1), 1.0, dict(skew_fraction=0.5)),
    "low_opacity_no_termination": (8_000, 256, 256, 3, 16, (0.0, 0.0, 0.0), 1.0, dict(opacity_const=0.1)),
    # thin, long splats at random angles: bounding boxes far larger than the ellipses the blend culls with
    "needles": (1_500, 320, 240, 1, 4, (0.0, 0.0, 0.0), 1.0, dict(anisotropy=6.0, scale_mult=0.25)),
    # binning extremes: rectangles that span every supercell, lists far longer than a partition piece,
    # a grid of 135 supercells with partial ones at both edges, a one-tile image
    "screen_filling_splats_1080p": (400, 1920, 1080, 0, 1, (0.0, 0.0, 0.0), 1.0, dict(scale_mult=6.0)),
    "uhd_3840x2160": (20_000, 3840, 2160, 1, 4, (0.0, 0.0, 0.0), 1.0, {}),
    "one_hot_tile": (12_000, 96, 64, 0, 1, (0.0, 0.0, 0.0), 1.0, dict(skew_fraction=1.0, distance=30.0)),
    "single_tile_image": (500, 16, 16, 0, 1, (0.3, 0.3, 0.3), 1.0, {}),
    # grids far from 1080p's: 510 supercells (30 x 17, partial at both edges), one row of 64 supercells, one column of 36
    "uhd8k_7680x4320": (8_000, 7680, 4320, 0, 1, (0.0, 0.0, 0.0), 1.0, {}),
    "panorama_16384x64": (20_000, 16384, 64, 0, 1, (0.0, 0.0, 0.0), 1.0, dict(scale_mult=2.0)),
    "tower_48x9000": (20_000, 48, 9000, 1, 4, (0.5, 0.5, 0.5), 1.0, {}),
}


# Faint, overlapping needles leave more pixels within rounding distance of a 1/255 decision, and
# their conics are ill-conditioned: the exponent is a small difference of large terms, so two
# roundings of it (the published expression's order in the default oracle, fused multiply-adds in the
# log2 domain in the kernels) differ by up to ~1e-4 relative in alpha.  No tolerance is widened for that:
# the scene is compared at the ordinary bars with the oracle build that evaluates the exponent in the
# kernels' order (oracle32k), and against the default oracle the allowance is the measured distance
# between the two oracle builds themselves (triangle inequality), printed beside the observed error.
MAX_BORDERLINE = {"needles": 1e-2}
ILL_CONDITIONED = ("needles",)
GRADS = ("means3D", "means2D", "shs", "opacities", "scales", "rotations")


def _run_case(oracle32, name, reference_lists=True):
    import diff_gaussian_rasterization as dgr
    P, W, H, deg, coeffs, bgc, mod, kw = CASES[name]
    cloud, cam, bg, _ = make_scene(P, W, H, sh_degree=deg, seed=11, bg=bgc, coeffs=coeffs, **kw)
    s_cpu = oracle_settings(cam, bg, deg, mod)
    cn = cloud_numpy(cloud)
    ref = oracle32.forward(s_cpu, **cn)
    s_gpu = _gpu_settings(cam, bg, deg, mod)
    c = cloud.to("cuda")
    st = dgr.rasterize_forward(s_gpu, c.xyz, c.opacity, shs=c.features, scales=c.scaling, rotations=c.rotation,
                               keep_tile_ids=True, reference_lists=reference_lists)
    torch.cuda.synchronize()
    return cloud, cam, bg, mod, s_cpu, s_gpu, cn, ref, st


def _check_ill_conditioned_forward(oracle32k, name, s_cpu, cn, ref, st):
    """Image and final_T at the ordinary bars against the kernel-order oracle; against the default oracle
    within those bars plus the distance between the two oracle builds."""
    refk = oracle32k.forward(s_cpu, **cn)
    fw, fk = ref["fwd"], refk["fwd"]
    color, final_T = st.color.cpu().numpy(), st.final_T.cpu().numpy()
    safe = (fw["margin"] >= MARGIN_SAFE) & (fk["margin"] >= MARGIN_SAFE)
    assert np.array_equal(st.n_contrib.cpu().numpy().view(np.uint32)[safe], fk["n_contrib"][safe])
    err_k = np.abs(color - fk["color"])[:, safe].max()
    between = np.abs(fk["color"] - fw["color"])[:, safe].max()
    err_c = np.abs(color - fw["color"])[:, safe].max()
    print(name, "image: vs kernel-order oracle %.2e, vs default oracle %.2e, the two oracles apart %.2e"
          % (err_k, err_c, between), flush=True)
    assert err_k <= PIX_TOL, err_k
    assert err_c <= PIX_TOL + between, (err_c, between)
    np.testing.assert_allclose(final_T[safe], fk["final_T"][safe], rtol=2e-5, atol=1e-7)


@pytest.mark.parametrize("name", list(CASES))
def test_forward_stages(oracle32, oracle32k, name):
    cloud, cam, bg, mod, s_cpu, s_gpu, cn, ref, st = _run_case(oracle32, name)
    pre, bn, fw = ref["pre"], ref["bin"], ref["fwd"]
    print(name, "P", st.P, "visible", int((pre["radii"] > 0).sum()), "instances", bn["n_dup"], flush=True)
    assert bn["n_dup"] > 0

    # --- preprocess: bit-exact ---------------------------------------------------
    radii = st.radii.cpu().numpy()
    assert np.array_equal(radii, pre["radii"])
    vis = radii > 0
    assert np.array_equal(st.tiles_touched.cpu().numpy().view(np.uint32), pre["tiles_touched"])
    assert np.array_equal(st.rects.cpu().numpy()[vis], pre["rects"][vis])
    assert np.array_equal(_bits(st.means2D.cpu().numpy())[vis], _bits(pre["means2D"])[vis])
    assert np.array_equal(_bits(st.depths.cpu().numpy())[vis], _bits(pre["depths"])[vis])
    assert np.array_equal(_bits(st.conic_opacity.cpu().numpy())[vis], _bits(pre["conic_opacity"])[vis])
    assert np.array_equal(_bits(st.rgb.cpu().numpy())[vis], _bits(pre["rgb"])[vis])
    assert np.array_equal(st.clamped.cpu().numpy()[vis], pre["clamped"][vis])

    # --- binning: bit-exact keys, values, ranges ------------------------------------
    assert st.num_rendered == bn["n_dup"]
    assert np.array_equal(st.point_list.cpu().numpy().view(np.uint32), bn["point_list"])
    assert np.array_equal(st.sort_keys().cpu().numpy().view(np.uint64), bn["keys"])
    assert np.array_equal(st.ranges.cpu().numpy().view(np.uint32), bn["ranges"])

    # --- blend: 1e-5 per pixel away from threshold coin-flips --------------------------
    color = st.color.cpu().numpy()
    safe = fw["margin"] >= MARGIN_SAFE
    frac_unsafe = 1.0 - safe.mean()
    print(name, "threshold-borderline pixels: %.4f%%" % (100 * frac_unsafe), flush=True)
    assert frac_unsafe < MAX_BORDERLINE.get(name, 5e-3)
    diff = np.abs(color - fw["color"])
    if name in ILL_CONDITIONED:
        _check_ill_conditioned_forward(oracle32k, name, s_cpu, cn, ref, st)
    else:
        print(name, "max |image - oracle| on safe pixels: %.2e" % diff[:, safe].max(), flush=True)
        assert diff[:, safe].max() <= PIX_TOL, diff[:, safe].max()
        np.testing.assert_allclose(st.final_T.cpu().numpy()[safe], fw["final_T"][safe], rtol=2e-5, atol=1e-7)
    assert diff.max() < 2e-2                       # a flipped decision moves a pixel by one contribution at most
    assert np.array_equal(st.n_contrib.cpu().numpy().view(np.uint32)[safe], fw["n_contrib"][safe])


@pytest.mark.parametrize("name", list(CASES))
def test_forward_with_short_lists(oracle32, oracle32k, name):
    """The optional short lists leave out the tiles that the bounding box of a Gaussian's
    alpha >= 1/255 ellipse cannot reach: every list is the module's list with entries removed (same
    order), and the image is the one the module's lists give, bit for bit."""
    import diff_gaussian_rasterization as dgr
    cloud, cam, bg, mod, s_cpu, s_gpu, cn, ref, st = _run_case(oracle32, name, reference_lists=False)
    c = cloud.to("cuda")
    full = dgr.rasterize_forward(s_gpu, c.xyz, c.opacity, shs=c.features, scales=c.scaling, rotations=c.rotation,
                                 keep_tile_ids=True, reference_lists=True)
    torch.cuda.synchronize()
    print(name, "instances", full.num_rendered, "->", st.num_rendered, flush=True)
    assert 0 < st.num_rendered <= full.num_rendered
    assert torch.equal(st.radii, full.radii) and torch.equal(st.visible, full.visible)
    assert torch.equal(st.records[st.radii > 0], full.records[full.radii > 0])
    # same pixels, bit for bit: what was left out contributed to no pixel
    assert torch.equal(st.color, full.color)
    assert torch.equal(st.final_T, full.final_T)
    # subsequence of the module's lists: (tile, Gaussian) pairs all present there, still in (tile, depth, index) order
    P = st.P
    got = st.tile_ids_sorted.to(torch.int64) * P + st.point_list.to(torch.int64)
    want = full.tile_ids_sorted.to(torch.int64) * P + full.point_list.to(torch.int64)
    assert bool(torch.isin(got, want).all())
    keys = st.sort_keys()
    assert bool((keys[1:] >= keys[:-1]).all())
    tie = keys[1:] == keys[:-1]
    assert bool((st.point_list[1:][tie] > st.point_list[:-1][tie]).all())
    r = st.ranges.to(torch.int64)
    assert int((r[:, 1] - r[:, 0]).sum()) == st.num_rendered
    # and against the oracle, as in the test above
    fw = ref["fwd"]
    safe = fw["margin"] >= MARGIN_SAFE
    diff = np.abs(st.color.cpu().numpy() - fw["color"])
    if name in ILL_CONDITIONED:
        fk = oracle32k.forward(s_cpu, **cn)["fwd"]
        safe = fk["margin"] >= MARGIN_SAFE
        diff = np.abs(st.color.cpu().numpy() - fk["color"])
    assert diff[:, safe].max() <= PIX_TOL


def _rel_err(got, ref):
    scale = np.abs(ref).max() + 1e-20
    return np.abs(got - ref).max() / scale


BACKWARD_CASES = [c for c in CASES if c != "uhd8k_7680x4320"]    # (it passes; the oracle's backward over 33M pixels takes 45 s)


def _backward_errors(orc, s_cpu, cn, ref, dL, g, ds):
    """Relative errors (of the largest entry) of the GPU's gradients against one oracle build's."""
    g_ref = orc.backward(s_cpu, ref, dL, cn["means3D"], scales=cn["scales"], rotations=cn["rotations"], shs=cn["shs"])
    bb = g_ref["blend"]
    blend = {nm: _rel_err(ds[nm], bb[key]) for nm, key in (("mean2D", "dL_dmean2D"), ("conic", "dL_dconic"),
                                                           ("opacity", "dL_dopacity"), ("color", "dL_dcolor"))}
    full = {nm: _rel_err(g[nm], g_ref[nm].reshape(g[nm].shape)) for nm in GRADS}
    return blend, full, g_ref


@pytest.mark.parametrize("name", BACKWARD_CASES)
def test_backward(oracle32, oracle32k, name):
    """Gradients at 2e-5 (blend sums) / 3e-5 (op inputs) of the largest entry.  Every scene against the default
    oracle; the ill-conditioned one against the kernel-order oracle at the same bars, and against the default
    oracle within those bars plus the measured distance between the two oracle builds."""
    import diff_gaussian_rasterization as dgr
    cloud, cam, bg, mod, s_cpu, s_gpu, cn, ref, st = _run_case(oracle32, name)
    rng = np.random.default_rng(3)
    dL = rng.standard_normal((3, cam.image_height, cam.image_width)).astype(np.float32)
    # Each side runs its backward from its OWN forward state.  Pixels whose forward decisions may
    # legitimately differ in fp32 (oracle margin below MARGIN_SAFE, ~0.1-0.2 % of pixels) get a zero
    # loss gradient, so that what is compared is the backward arithmetic on identical decisions.
    safe = ref["fwd"]["margin"] >= MARGIN_SAFE
    refk = None
    if name in ILL_CONDITIONED:
        refk = oracle32k.forward(s_cpu, **cn)
        safe &= refk["fwd"]["margin"] >= MARGIN_SAFE
    dL *= safe[None].astype(np.float32)
    c = cloud.to("cuda")
    dLg = torch.from_numpy(dL).cuda()
    dsplat = dgr.blend_backward(st, dLg)
    g = dgr.rasterize_backward(s_gpu, st, dLg, c.xyz, shs=c.features, scales=c.scaling, rotations=c.rotation,
                               dsplat=dsplat)
    torch.cuda.synchronize()
    g = {k: v.cpu().numpy() for k, v in g.items() if v is not None}
    assert all(np.isfinite(v).all() for v in g.values())
    ds = {k: v.cpu().numpy() for k, v in dgr.splat_sums_to_grads(st, dsplat).items()}
    blend, full, g_ref = _backward_errors(oracle32, s_cpu, cn, ref, dL, g, ds)
    if refk is None:
        for nm, e in blend.items():
            print(name, "blend_bwd", nm, "rel err %.2e" % e, flush=True)
            assert e < 2e-5, (nm, e)
        for nm, e in full.items():
            print(name, "grad", nm, "rel err %.2e" % e, flush=True)
            assert e < 3e-5, (nm, e)
        return
    blend_k, full_k, g_refk = _backward_errors(oracle32k, s_cpu, cn, refk, dL, g, ds)
    # (1) what the blend backward leaves per Gaussian, at the ordinary bar against the kernel-order oracle
    for nm in blend:
        print(name, "blend_bwd", nm, "rel err vs kernel-order oracle %.2e, vs default oracle %.2e" % (blend_k[nm], blend[nm]),
              flush=True)
        assert blend_k[nm] < 2e-5, (nm, blend_k[nm])
    # (2) the per-Gaussian backward is the oracle's function: fed the GPU's own per-Gaussian sums, the oracle's
    # preprocess backward returns the GPU's gradients.  (The map from the conic's gradient to the covariance's goes
    # through the inverse of a near-singular 2x2 here, so it amplifies the 1e-6 summation-order differences of (1)
    # several hundred times: end to end, means3D / scales / rotations are NOT within 3e-5 on this scene against
    # any oracle that sums in another order, and that number is printed, not asserted.)
    bb = dict(dL_dmean2D=ds["mean2D"], dL_dconic=ds["conic"], dL_dopacity=ds["opacity"], dL_dcolor=ds["color"])
    pb = oracle32k.preprocess_bwd(s_cpu, refk["pre"], bb, cn["means3D"], cn["scales"], cn["rotations"], cn["shs"])
    same_inputs = dict(means3D=pb["dL_dmeans3D"], shs=pb["dL_dshs"], scales=pb["dL_dscales"], rotations=pb["dL_drotations"])
    for nm, want in same_inputs.items():
        e = _rel_err(g[nm], want.reshape(g[nm].shape))
        print(name, "grad", nm, "rel err vs the oracle's per-Gaussian backward on the same sums %.2e" % e, flush=True)
        assert e < 3e-5, (nm, e)
    for nm in GRADS:
        between = _rel_err(g_refk[nm], g_ref[nm])
        print(name, "grad", nm, "end to end: rel err vs kernel-order oracle %.2e, vs default oracle %.2e, the two oracles apart %.2e"
              % (full_k[nm], full[nm], between), flush=True)
        if nm in ("means2D", "opacities", "shs"):             # linear in the sums of (1) with bounded coefficients
            assert full_k[nm] < 3e-5, (nm, full_k[nm])
        # a loose ceiling on the amplified figure (it moves with the order in which the float atomics land: 1.8e-3
        # and 2.9e-3 for the scales in two runs of the same code); steps (1) and (2) are the parity statement here
        assert full_k[nm] < 1e-2 and full[nm] < 1e-2 + between, (nm, full_k[nm], full[nm])


@pytest.mark.parametrize("name", ["ragged_5k_333x200_deg3", "skewed_hot_tiles", "low_opacity_no_termination"])
def test_backward_against_the_kernel_order_oracle(oracle32k, name):
    """The well-conditioned scenes pass against either oracle build (the builds differ by rounding only)."""
    import diff_gaussian_rasterization as dgr
    cloud, cam, bg, mod, s_cpu, s_gpu, cn, ref, st = _run_case(oracle32k, name)
    safe = ref["fwd"]["margin"] >= MARGIN_SAFE
    diff = np.abs(st.color.cpu().numpy() - ref["color"])
    assert diff[:, safe].max() <= PIX_TOL
    assert np.array_equal(st.n_contrib.cpu().numpy().view(np.uint32)[safe], ref["fwd"]["n_contrib"][safe])
    dL = np.random.default_rng(4).standard_normal((3, cam.image_height, cam.image_width)).astype(np.float32) * safe[None]
    c = cloud.to("cuda")
    dLg = torch.from_numpy(dL).cuda()
    dsplat = dgr.blend_backward(st, dLg)
    g = dgr.rasterize_backward(s_gpu, st, dLg, c.xyz, shs=c.features, scales=c.scaling, rotations=c.rotation, dsplat=dsplat)
    g = {k: v.cpu().numpy() for k, v in g.items() if v is not None}
    ds = {k: v.cpu().numpy() for k, v in dgr.splat_sums_to_grads(st, dsplat).items()}
    blend, full, _ = _backward_errors(oracle32k, s_cpu, cn, ref, dL, g, ds)
    assert all(e < 2e-5 for e in blend.values()), blend
    assert all(e < 3e-5 for e in full.values()), full


def compare_input_paths(rast, s, c, w, img, radii, grads, degree=3, tol=1e-4):
    """The reference's two interchangeable input paths (gaussian_renderer/__init__.py:62-66 and :72-80):
    ``colors_precomp = clamp_min(eval_sh(shs) + 0.5, 0)`` and ``cov3D_precomp = covariance(scales, rotations)``
    evaluated in PyTorch WITH autograd and handed to the operator, against the operator fed ``shs``, ``scales``
    and ``rotations`` directly (``img``, ``radii``, ``grads`` of loss = sum(img * w)).  Same image, and the same
    gradient of every parameter (the second path's reach SH coefficients, scales and rotations through PyTorch's
    backward of eval_sh and of the covariance): within ``tol`` of the largest entry, over the Gaussians whose
    radius is the same on both paths (a covariance rounded differently flips a ceil() for a few in ten thousand).
    A flipped radius changes which tiles that Gaussian is listed in, i.e. it adds or removes a faint contribution
    (alpha <= 0.011) on a few pixels, and with it the transmittance of everything behind it there; the handful of
    Gaussians that sit behind one may therefore differ by more: at most one in ten thousand may exceed ``tol``, none
    ten times ``tol``.  Returns {name: (largest relative error, share of Gaussians beyond tol)}."""
    from gsplat_host.model import covariance_from_scaling_rotation
    from gsplat_host.sh import eval_sh
    leaf = lambda t: t.detach().clone().requires_grad_(True)
    m3, op, sc, ro, sh = leaf(c.xyz), leaf(c.opacity), leaf(c.scaling), leaf(c.rotation), leaf(c.features)
    d = m3 - s.campos[None]
    d = d / d.norm(dim=1, keepdim=True)
    col = torch.clamp_min(eval_sh(degree, sh.transpose(1, 2), d) + 0.5, 0.0)
    cov = covariance_from_scaling_rotation(sc, 1.0, ro)
    col.retain_grad()
    cov.retain_grad()
    sink = torch.zeros_like(m3, requires_grad=True)
    img2, radii2 = rast(means3D=m3, means2D=sink, shs=None, colors_precomp=col, opacities=op, scales=None,
                        rotations=None, cov3D_precomp=cov)
    (img2 * w).sum().backward()
    assert col.grad.shape == (c.xyz.shape[0], 3) and cov.grad.shape == (c.xyz.shape[0], 6)
    same = radii2 == radii
    assert float((~same).float().mean()) < 1e-3
    diff = (img2 - img).abs()
    assert float(diff.max()) < 2e-2 and float(diff.mean()) < 1e-5 and float((diff > 1e-4).float().mean()) < 1e-3
    got = dict(means3D=m3.grad, opacities=op.grad, shs=sh.grad, scales=sc.grad, rotations=ro.grad, means2D=sink.grad)
    # The covariance helper normalises the quaternion (utils/general_utils.py:79-81), the operator takes it as
    # given: the first path's rotation gradient has a component along q that the second projects out, exactly
    # as normalize() does for the model's raw rotations in the reference (scene/gaussian_model.py:100-101).
    q = c.rotation / c.rotation.norm(dim=1, keepdim=True)
    tangent = lambda g: g - q * (q * g).sum(dim=1, keepdim=True)
    errs = {}
    for k, want in grads.items():
        have = got[k]
        if k == "rotations":
            have, want = tangent(have), tangent(want)
        scale = float(want.abs().max()) + 1e-20
        per = (have[same] - want[same]).abs().reshape(int(same.sum()), -1).amax(dim=1) / scale
        errs[k] = (float(per.max()), float((per > tol).float().mean()))
    assert all(worst <= 10 * tol and beyond <= 1e-4 for worst, beyond in errs.values()), errs
    return errs


def test_module_api_and_alternative_inputs(oracle32):
    """The nn.Module surface, autograd wiring, and the two interchangeable input
    paths of the reference (gaussian_renderer/__init__.py:62-66 and :72-80)."""
    import diff_gaussian_rasterization as dgr
    cloud, cam, bg, _ = make_scene(4_000, 200, 120, sh_degree=3, seed=5, bg=(0.3, 0.3, 0.3))
    s = _gpu_settings(cam, bg, 3, 1.0)
    c = cloud.to("cuda")
    leaf = lambda t: t.clone().requires_grad_(True)
    m3, op, sc, ro, sh = leaf(c.xyz), leaf(c.opacity), leaf(c.scaling), leaf(c.rotation), leaf(c.features)
    sink = torch.zeros_like(m3, requires_grad=True)
    rast = dgr.GaussianRasterizer(raster_settings=s)
    img, radii = rast(means3D=m3, means2D=sink, shs=sh, colors_precomp=None, opacities=op, scales=sc, rotations=ro,
                      cov3D_precomp=None)
    assert img.shape == (3, 120, 200) and img.dtype == torch.float32
    assert radii.shape == (4000,) and radii.dtype == torch.int32
    w = torch.randn_like(img)
    (img * w).sum().backward()
    for t in (m3, op, sc, ro, sh, sink):
        assert t.grad is not None and torch.isfinite(t.grad).all()
    assert sink.grad.shape == (4000, 3) and float(sink.grad[:, 2].abs().max()) == 0.0
    vis = radii > 0
    assert float(sink.grad[~vis].abs().max()) == 0.0 and float(sink.grad[vis].abs().max()) > 0.0
    assert torch.equal(rast.markVisible(c.xyz).cpu(), torch.from_numpy(
        oracle32.mark_visible(oracle_settings(cam, bg, 3, 1.0), cloud.xyz.numpy())))

    # SH evaluated in Python and handed over as colors_precomp; covariance precomputed in Python
    errs = compare_input_paths(rast, s, c, w, img, radii, dict(means3D=m3.grad, opacities=op.grad, shs=sh.grad,
                                                                scales=sc.grad, rotations=ro.grad, means2D=sink.grad))
    print("two input paths, gradients (largest error, share beyond 1e-4):", {k: "%.1e %.1e" % v for k, v in errs.items()}, flush=True)
    col = torch.rand((4000, 3), device="cuda")
    cov = c.scaling.new_zeros((4000, 6))

    with pytest.raises(Exception):
        rast(means3D=m3, means2D=sink, shs=sh, colors_precomp=col, opacities=op, scales=sc, rotations=ro)
    with pytest.raises(Exception):
        rast(means3D=m3, means2D=sink, shs=sh, opacities=op, scales=sc, rotations=ro, cov3D_precomp=cov)


def test_empty_and_fully_culled():
    import diff_gaussian_rasterization as dgr
    cloud, cam, bg, _ = make_scene(64, 64, 48, sh_degree=0, seed=1, bg=(0.5, 0.25, 0.125), coeffs=1)
    s = _gpu_settings(cam, bg, 0, 1.0)
    c = cloud.to("cuda")
    behind = c.xyz.clone()
    behind[:, 2] += 100.0          # all behind the camera
    st = dgr.rasterize_forward(s, behind, c.opacity, shs=c.features, scales=c.scaling, rotations=c.rotation)
    torch.cuda.synchronize()
    assert st.num_rendered == 0 and int(st.radii.abs().sum()) == 0
    want = bg.cuda()[:, None, None].expand(3, 48, 64)
    assert torch.equal(st.color, want)
    g = dgr.rasterize_backward(s, st, torch.ones_like(st.color), behind, shs=c.features, scales=c.scaling,
                               rotations=c.rotation)
    assert all(float(v.abs().max()) == 0.0 for v in g.values() if v is not None)
