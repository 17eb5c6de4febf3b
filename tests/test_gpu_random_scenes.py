"""Seeded random scene shapes against the oracle: image sizes that are not multiples of the tile,
supercell or loss-tile sizes, every SH degree and coefficient count, scale modifiers, dense and
sparse clouds.  Lists bit-exact, image within the blend tolerance.  GSR_RANDOM_SCENES=<n> widens the sweep
beyond the sixteen seeds the suite runs; GSR_RANDOM_SEGMENTS=<entries> runs it with depth segments of that length."""
import os

import numpy as np
import pytest
import torch

from scenes import cloud_numpy, make_scene, oracle_settings
from test_gpu_parity import MARGIN_SAFE, PIX_TOL, _gpu_settings, _rel_err

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True, scope="module")
def _segments():
    entries = int(os.environ.get("GSR_RANDOM_SEGMENTS", "0"))
    if entries:
        import diff_gaussian_rasterization as dgr
        dgr.set_segment_length(entries)
        yield
        dgr.set_segment_length(2048)
    else:
        yield


def _draw(seed):
    rng = np.random.default_rng(1000 + seed)
    deg = int(rng.integers(0, 4))
    coeffs = int(rng.choice([(deg + 1) ** 2, 16]))
    return dict(P=int(rng.integers(1, 9000)), W=int(rng.integers(1, 700)), H=int(rng.integers(1, 500)), deg=deg,
                coeffs=coeffs, mod=float(rng.choice([1.0, 0.5, 1.7])), scale_mult=float(rng.choice([0.3, 1.0, 3.0])),
                distance=float(rng.choice([2.5, 4.0, 9.0])), skew=float(rng.choice([0.0, 0.0, 0.7])),
                bg=tuple(float(v) for v in rng.random(3)), angle=float(rng.uniform(0, 360)))


@pytest.mark.parametrize("seed", range(int(os.environ.get("GSR_RANDOM_SCENES", "16"))))
def test_random_scene(oracle32, seed):
    import diff_gaussian_rasterization as dgr
    k = _draw(seed)
    print(seed, k, flush=True)
    cloud, cam, bg, _ = make_scene(k["P"], k["W"], k["H"], sh_degree=k["deg"], seed=seed, angle=k["angle"],
                                   scale_mult=k["scale_mult"], bg=k["bg"], skew_fraction=k["skew"], coeffs=k["coeffs"],
                                   distance=k["distance"])
    s_cpu = oracle_settings(cam, bg, k["deg"], k["mod"])
    ref = oracle32.forward(s_cpu, **cloud_numpy(cloud))
    c = cloud.to("cuda")
    st = dgr.rasterize_forward(_gpu_settings(cam, bg, k["deg"], k["mod"]), c.xyz, c.opacity, shs=c.features,
                               scales=c.scaling, rotations=c.rotation, keep_tile_ids=True, reference_lists=True)
    torch.cuda.synchronize()
    pre, bn, fw = ref["pre"], ref["bin"], ref["fwd"]
    assert np.array_equal(st.radii.cpu().numpy(), pre["radii"])
    assert st.num_rendered == bn["n_dup"]
    assert np.array_equal(st.point_list.cpu().numpy().view(np.uint32), bn["point_list"])
    assert np.array_equal(st.ranges.cpu().numpy().view(np.uint32), bn["ranges"])
    if bn["n_dup"]:
        assert np.array_equal(st.sort_keys().cpu().numpy().view(np.uint64), bn["keys"])
    safe = fw["margin"] >= MARGIN_SAFE
    diff = np.abs(st.color.cpu().numpy() - fw["color"])
    if safe.any():
        assert diff[:, safe].max() <= PIX_TOL, (diff[:, safe].max(), 1.0 - safe.mean())
    assert diff.max() < 2e-2
    # the short lists render the same pixels
    short = dgr.rasterize_forward(_gpu_settings(cam, bg, k["deg"], k["mod"]), c.xyz, c.opacity, shs=c.features,
                                  scales=c.scaling, rotations=c.rotation, reference_lists=False)
    torch.cuda.synchronize()
    assert torch.equal(short.color, st.color)

    # gradients, from the short lists, with the loss gradient zeroed on borderline pixels
    if seed % 2 == 0 and bn["n_dup"] and safe.any():
        cn = cloud_numpy(cloud)
        dL = np.random.default_rng(seed).standard_normal((3, cam.image_height, cam.image_width)).astype(np.float32)
        dL *= safe[None].astype(np.float32)
        g_ref = oracle32.backward(s_cpu, ref, dL, cn["means3D"], scales=cn["scales"], rotations=cn["rotations"],
                                  shs=cn["shs"])
        dLg = torch.from_numpy(dL).cuda()
        dsplat = dgr.blend_backward(short, dLg)
        g = dgr.rasterize_backward(_gpu_settings(cam, bg, k["deg"], k["mod"]), short, dLg, c.xyz, shs=c.features,
                                   scales=c.scaling, rotations=c.rotation, dsplat=dsplat)
        torch.cuda.synchronize()
        amplified = {}
        for nm in ("means3D", "means2D", "shs", "opacities", "scales", "rotations"):
            got = g[nm].cpu().numpy()
            e = _rel_err(got, g_ref[nm].reshape(got.shape))
            print(seed, "grad", nm, "rel err %.2e" % e, flush=True)
            if e >= 3e-5 and nm in ("scales", "rotations"):
                amplified[nm] = e
                continue
            assert e < 3e-5, (nm, e)
        if amplified:
            # Flat splats multiply the 1e-6 by which two summation orders differ in the per-Gaussian sums through the
            # inverse of the 2x2 screen covariance (one seed in some hundreds lands a hair above the bar, and not on
            # every run: the sums' order is the atomics').  As in the full-size test no bar is moved: the sums are
            # held to their own bar, the per-Gaussian backward is shown to be the oracle's function on those same
            # sums, and the end-to-end figure gets a ceiling.
            ds = {k2: v.cpu().numpy() for k2, v in dgr.splat_sums_to_grads(short, dsplat).items()}
            for nm, key in (("mean2D", "dL_dmean2D"), ("conic", "dL_dconic"), ("opacity", "dL_dopacity"), ("color", "dL_dcolor")):
                e = _rel_err(ds[nm], g_ref["blend"][key])
                assert e < 2e-5, (nm, e)
            bb = dict(dL_dmean2D=ds["mean2D"], dL_dconic=ds["conic"], dL_dopacity=ds["opacity"], dL_dcolor=ds["color"])
            pb = oracle32.preprocess_bwd(s_cpu, ref["pre"], bb, cn["means3D"], cn["scales"], cn["rotations"], cn["shs"])
            for nm, e in amplified.items():
                got = g[nm].cpu().numpy()
                same = _rel_err(got, pb["dL_d" + nm].reshape(got.shape))
                print(seed, "grad", nm, "rel err vs the oracle's per-Gaussian backward on the same sums %.2e" % same, flush=True)
                assert same < 3e-5 and e < 3e-4, (nm, same, e)
