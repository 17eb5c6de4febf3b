"""Depth segments of the blend backward (gsr_blend_bwd_segments): tiles whose walk is long are walked by several
workgroups, each from the state the forward wrote down at a segment boundary.  The library's own segment
length (2048 entries) is beyond what small test scenes reach, so the tests lower it; the results must be the
oracle's at the ordinary bars, and those of whole walks up to rounding."""
import os

import numpy as np
import pytest
import torch

from scenes import cloud_numpy, make_scene, oracle_settings
from test_gpu_parity import GRADS, MARGIN_SAFE, PIX_TOL, _backward_errors, _gpu_settings

pytestmark = [pytest.mark.gpu,
              pytest.mark.skipif(os.environ.get("GSR_NO_SEGMENTS") is not None, reason="GSR_NO_SEGMENTS switches the segments off")]

# name: (P, W, H, degree, coeffs, bg, kwargs): faint splats, so that the walks go deep; one tile, hot tiles, all tiles
CASES = {
    "one_hot_tile_faint": (12_000, 96, 64, 0, 1, (0.0, 0.0, 0.0), dict(skew_fraction=1.0, distance=30.0, opacity_const=0.01)),
    "skewed_hot_tiles_faint": (30_000, 320, 192, 3, 16, (0.1, 0.1, 0.1), dict(skew_fraction=0.5, opacity_const=0.02)),
    "low_opacity_no_termination": (8_000, 256, 256, 3, 16, (0.3, 0.0, 0.6), dict(opacity_const=0.1)),
    "ragged_image_faint": (9_000, 333, 200, 1, 4, (0.0, 0.0, 0.0), dict(opacity_const=0.05, scale_mult=1.5)),
}


@pytest.fixture
def segment_length():
    import diff_gaussian_rasterization as dgr
    yield dgr.set_segment_length
    dgr.set_segment_length(2048, 0)


def _forward(dgr, s_gpu, c):
    # (the module's own lists, whatever GSR_SHORT_LISTS says: the scenes' walk depths are chosen for them)
    st = dgr.rasterize_forward(s_gpu, c.xyz, c.opacity, shs=c.features, scales=c.scaling, rotations=c.rotation,
                               prepare_backward=True, single_call=True, reference_lists=True)
    torch.cuda.synchronize()
    return st


def _backward(dgr, s_gpu, st, dLg, c):
    dsplat = dgr.blend_backward(st, dLg)
    g = dgr.rasterize_backward(s_gpu, st, dLg, c.xyz, shs=c.features, scales=c.scaling, rotations=c.rotation, dsplat=dsplat)
    torch.cuda.synchronize()
    ds = {k: v.cpu().numpy() for k, v in dgr.splat_sums_to_grads(st, dsplat).items()}
    return {k: v.cpu().numpy() for k, v in g.items() if v is not None}, ds, dsplat


@pytest.mark.parametrize("length", [256, 512])
@pytest.mark.parametrize("name", list(CASES))
def test_segmented_backward_against_the_oracle(oracle32, segment_length, name, length):
    import diff_gaussian_rasterization as dgr
    P, W, H, deg, coeffs, bgc, kw = CASES[name]
    cloud, cam, bg, mod = make_scene(P, W, H, sh_degree=deg, seed=11, bg=bgc, coeffs=coeffs, **kw)
    s_cpu, cn = oracle_settings(cam, bg, deg, mod), cloud_numpy(cloud)
    ref = oracle32.forward(s_cpu, **cn)
    s_gpu, c = _gpu_settings(cam, bg, deg, mod), cloud.to("cuda")
    safe = ref["fwd"]["margin"] >= MARGIN_SAFE
    dL = np.random.default_rng(5).standard_normal((3, H, W)).astype(np.float32) * safe[None]
    dLg = torch.from_numpy(dL).cuda()

    segment_length(0)
    whole = _forward(dgr, s_gpu, c)
    assert whole.segments is None
    g_whole, ds_whole, sums_whole = _backward(dgr, s_gpu, whole, dLg, c)

    segment_length(length)
    st = _forward(dgr, s_gpu, c)
    asked, used_length, per_tile = dgr.segment_counts(st)
    deepest = int(st.n_contrib.max())
    print(name, "segment length", length, "deepest walk", deepest, "dumps", asked, "most in one tile", int(per_tile.max()), flush=True)
    assert used_length == length          # lists this small fit the pool at any length
    assert 0 < asked <= dgr.SEGMENT_POOL and int(per_tile.sum()) == asked          # the case does cut walks
    assert int(per_tile.max()) >= min(128, (deepest - 1) // length - 1)
    # the forward's own outputs do not depend on the dumps
    assert torch.equal(st.color, whole.color) and torch.equal(st.n_contrib, whole.n_contrib) and torch.equal(st.final_T, whole.final_T)
    diff = np.abs(st.color.cpu().numpy() - ref["color"])
    assert diff[:, safe].max() <= PIX_TOL

    g, ds, sums = _backward(dgr, s_gpu, st, dLg, c)
    blend, full, _ = _backward_errors(oracle32, s_cpu, cn, ref, dL, g, ds)
    for nm, e in blend.items():
        print(name, "blend_bwd", nm, "rel err vs oracle %.2e" % e, flush=True)
        assert e < 2e-5, (nm, e)
    for nm, e in full.items():
        print(name, "grad", nm, "rel err vs oracle %.2e" % e, flush=True)
        assert e < 3e-5, (nm, e)
    # against whole walks on the GPU: the same sums up to rounding
    scale = float(sums_whole.abs().max())
    apart = float((sums - sums_whole).abs().max()) / scale
    print(name, "per-Gaussian sums, segmented vs whole walks: %.2e of the largest" % apart, flush=True)
    assert apart < 2e-5
    # a second backward on the same forward state finds the segments intact
    again = dgr.blend_backward(st, dLg)
    assert float((again - sums).abs().max()) / scale < 1e-5


def test_stage_calls_keep_segments_too(segment_length):
    """gsr_blend_fwd_segments behind the sequence of stage calls: the same boundaries and sums as the single call."""
    import diff_gaussian_rasterization as dgr
    P, W, H, deg, coeffs, bgc, kw = CASES["skewed_hot_tiles_faint"]
    cloud, cam, bg, mod = make_scene(P, W, H, sh_degree=deg, seed=11, bg=bgc, coeffs=coeffs, **kw)
    s_gpu, c = _gpu_settings(cam, bg, deg, mod), cloud.to("cuda")
    dLg = torch.randn(3, H, W, device="cuda", generator=torch.Generator("cuda").manual_seed(4))
    segment_length(256)
    one = _forward(dgr, s_gpu, c)
    staged = dgr.rasterize_forward(s_gpu, c.xyz, c.opacity, shs=c.features, scales=c.scaling, rotations=c.rotation,
                                   prepare_backward=True, single_call=False, reference_lists=True)
    a, la, ta = dgr.segment_counts(one)
    b, lb, tb = dgr.segment_counts(staged)
    assert staged.segments_block is not None and (a, la) == (b, lb) and a > 0 and torch.equal(ta, tb)
    assert torch.equal(one.color, staged.color)
    sums_one, sums_staged = dgr.blend_backward(one, dLg), dgr.blend_backward(staged, dLg)
    assert float((sums_one - sums_staged).abs().max()) / float(sums_one.abs().max()) < 1e-5
    # without a backward to prepare, the stage calls keep none
    plain = dgr.rasterize_forward(s_gpu, c.xyz, c.opacity, shs=c.features, scales=c.scaling, rotations=c.rotation, single_call=False)
    assert plain.segments is None


def test_the_pool_running_dry_only_ends_the_cutting(segment_length):
    """More boundaries than the pool has dumps (the library sizes its segments so that this cannot happen; the test
    shrinks the pool): the tiles that come too late are walked whole, the sums are the same."""
    import diff_gaussian_rasterization as dgr
    cloud, cam, bg, mod = make_scene(60_000, 640, 384, sh_degree=0, seed=5, coeffs=1, opacity_const=0.004, scale_mult=2.0)
    s_gpu, c = _gpu_settings(cam, bg, 0, mod), cloud.to("cuda")
    dLg = torch.randn(3, 384, 640, device="cuda", generator=torch.Generator("cuda").manual_seed(2))
    segment_length(0)
    whole = _forward(dgr, s_gpu, c)
    sums_whole = dgr.blend_backward(whole, dLg)
    segment_length(256, pool=1000)
    st = _forward(dgr, s_gpu, c)
    asked, used_length, per_tile = dgr.segment_counts(st)
    print("segments of", used_length, "dumps asked for", asked, "pool 1000, taken", int(per_tile.sum()), "deepest walk",
          int(st.n_contrib.max()), "entries", st.num_rendered, flush=True)
    assert asked > 1000 and int(per_tile.sum()) == 1000
    sums = dgr.blend_backward(st, dLg)
    torch.cuda.synchronize()
    scale = float(sums_whole.abs().max())
    assert float((sums - sums_whole).abs().max()) / scale < 2e-5
    # and a tile may keep kMaxTileDumps = 128 dumps at most: a walk deeper than 128 segments ends in one long stretch
    cloud, cam, bg, mod = make_scene(48_000, 96, 64, sh_degree=0, seed=6, coeffs=1, skew_fraction=1.0, distance=30.0,
                                     opacity_const=0.004)
    s_gpu, c = _gpu_settings(cam, bg, 0, mod), cloud.to("cuda")
    dLg = torch.randn(3, 64, 96, device="cuda", generator=torch.Generator("cuda").manual_seed(3))
    segment_length(0)
    sums_whole = dgr.blend_backward(_forward(dgr, s_gpu, c), dLg)
    segment_length(256, pool=0)
    st = _forward(dgr, s_gpu, c)
    asked, _, per_tile = dgr.segment_counts(st)
    print("one hot tile: deepest walk", int(st.n_contrib.max()), "dumps per tile", per_tile.tolist(), flush=True)
    assert int(per_tile.max()) == 128 and int(st.n_contrib.max()) > 129 * 256
    sums = dgr.blend_backward(st, dLg)
    assert float((sums - sums_whole).abs().max()) / float(sums_whole.abs().max()) < 2e-5


def test_segments_grow_with_the_lists_so_that_the_pool_holds_every_boundary(segment_length):
    """Lists of N entries are cut every max(configured, N / pool rounded up to a multiple of 256) entries."""
    import diff_gaussian_rasterization as dgr
    cloud, cam, bg, mod = make_scene(60_000, 640, 384, sh_degree=0, seed=5, coeffs=1, opacity_const=0.004, scale_mult=2.0)
    s_gpu, c = _gpu_settings(cam, bg, 0, mod), cloud.to("cuda")
    segment_length(256)
    st = _forward(dgr, s_gpu, c)
    asked, used_length, per_tile = dgr.segment_counts(st)
    N = st.num_rendered
    print("entries", N, "segments of", used_length, "dumps", asked, flush=True)
    assert used_length == max(256, (N // dgr.SEGMENT_POOL + 256) // 256 * 256)
    assert asked == int(per_tile.sum()) <= dgr.SEGMENT_POOL
    lengths = (st.ranges[:, 1] - st.ranges[:, 0]).cpu()
    assert bool((per_tile <= lengths // used_length).all())


def test_the_library_s_own_segment_length_against_the_oracle(oracle32):
    """48 000 faint Gaussians over four tiles: walks 48 000 entries deep, cut every 2048 entries as shipped."""
    import diff_gaussian_rasterization as dgr
    P, W, H = 48_000, 96, 64
    cloud, cam, bg, mod = make_scene(P, W, H, sh_degree=0, seed=6, coeffs=1, skew_fraction=1.0, distance=30.0, opacity_const=0.004)
    s_cpu, cn = oracle_settings(cam, bg, 0, mod), cloud_numpy(cloud)
    ref = oracle32.forward(s_cpu, **cn)
    s_gpu, c = _gpu_settings(cam, bg, 0, mod), cloud.to("cuda")
    safe = ref["fwd"]["margin"] >= MARGIN_SAFE
    dL = np.random.default_rng(8).standard_normal((3, H, W)).astype(np.float32) * safe[None]
    dLg = torch.from_numpy(dL).cuda()
    st = _forward(dgr, s_gpu, c)
    asked, length, per_tile = dgr.segment_counts(st)
    print("deepest walk", int(st.n_contrib.max()), "segments of", length, "boundaries", asked, flush=True)
    assert length == 2048 and int(per_tile.max()) >= 20
    assert np.abs(st.color.cpu().numpy() - ref["color"])[:, safe].max() <= PIX_TOL
    g, ds, _ = _backward(dgr, s_gpu, st, dLg, c)
    blend, full, _ = _backward_errors(oracle32, s_cpu, cn, ref, dL, g, ds)
    print({k: "%.1e" % v for k, v in {**blend, **full}.items()}, flush=True)
    assert all(e < 2e-5 for e in blend.values()), blend
    assert all(e < 3e-5 for e in full.values()), full


def test_full_size_after_opacity_reset_segments_match_whole_walks(segment_length):
    """The benchmark's c5r workload (6M Gaussians, half of them in a hot spot, every opacity 0.01 as after the
    reference's reset_opacity, scene/gaussian_model.py:210-213; 1920x1080): as shipped the backward cuts its deepest
    walks (45,000 entries) into segments of N / 8192 entries; the sums are those of whole walks."""
    import diff_gaussian_rasterization as dgr
    W, H, deg = 1920, 1080, 3
    cloud, cam, bg, mod = make_scene(6_000_000, W, H, sh_degree=deg, seed=0, skew_fraction=0.5, opacity_const=0.01)
    s_gpu, c = _gpu_settings(cam, bg, deg, mod), cloud.to("cuda")
    del cloud
    dLg = torch.randn(3, H, W, device="cuda", generator=torch.Generator("cuda").manual_seed(6)) / (3 * H * W)
    st = _forward(dgr, s_gpu, c)
    asked, length, per_tile = dgr.segment_counts(st)
    deepest = int(st.n_contrib.max())
    print("entries", st.num_rendered, "deepest walk", deepest, "segments of", length, "boundaries", asked, "most in one tile",
          int(per_tile.max()), flush=True)
    assert length == (st.num_rendered // dgr.SEGMENT_POOL + 256) // 256 * 256 > 2048
    assert asked == int(per_tile.sum()) and int(per_tile.max()) == (deepest - 1) // length >= 4
    sums = dgr.blend_backward(st, dLg).clone()
    color = st.color.clone()
    del st
    segment_length(0)
    whole = _forward(dgr, s_gpu, c)
    assert whole.segments is None and torch.equal(whole.color, color)
    sums_whole = dgr.blend_backward(whole, dLg)
    scale = float(sums_whole.abs().max())
    apart = float((sums - sums_whole).abs().max()) / scale
    print("per-Gaussian sums, segmented vs whole walks: %.2e of the largest" % apart, flush=True)
    assert apart < 2e-5
    del whole, sums, sums_whole, c
    torch.cuda.empty_cache()


def test_default_length_leaves_ordinary_scenes_alone():
    """At the library's own segment length an ordinary scene takes no dump: the backward is the whole-walk one."""
    import diff_gaussian_rasterization as dgr
    cloud, cam, bg, mod = make_scene(20_000, 320, 192, sh_degree=1, seed=3, coeffs=4)
    s_gpu, c = _gpu_settings(cam, bg, 1, mod), cloud.to("cuda")
    st = _forward(dgr, s_gpu, c)
    asked, length, per_tile = dgr.segment_counts(st)
    assert st.segments is not None and length == 2048 and asked == 0 and int(per_tile.sum()) == 0
