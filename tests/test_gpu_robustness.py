"""Edges of the operator that the parity scenes do not reach: opacity exactly zero, the reference's own
gradient-sink idiom, and what happens when the tile lists cannot be built (a depth sort that gave up, list
sizes that are not the call's own).  Failures are forced from the host: a patience of one look-back step,
a flag word the test sets itself.  Every kernel still ends normally."""
import math

import numpy as np
import pytest
import torch

from scenes import make_scene
from test_gpu_parity import _gpu_settings

pytestmark = pytest.mark.gpu


def test_opacity_exactly_zero_gives_zero_gradients_not_nan():
    """A visible Gaussian with opacity 0 blends nowhere; its opacity gradient is 0 (sum of G dL/dalpha over
    no pixel), not 0/0.  Both entry points: activated opacities with exact zeros, raw opacities of -200
    (sigmoid underflows to 0)."""
    import diff_gaussian_rasterization as dgr
    from gsplat_host.model import GaussianParams
    from gsplat_host.render import PipelineParams, render
    P, W, H = 6_000, 256, 160
    cloud, cam, bg, _ = make_scene(P, W, H, sh_degree=2, seed=31, bg=(0.2, 0.1, 0.3), coeffs=9)
    s = _gpu_settings(cam, bg, 2, 1.0)
    c = cloud.to("cuda")
    dead = torch.arange(P, device="cuda") % 3 == 0
    op = c.opacity.clone()
    op[dead] = 0.0
    leaf = lambda t: t.clone().requires_grad_(True)
    m3, opl, sc, ro, sh = leaf(c.xyz), leaf(op), leaf(c.scaling), leaf(c.rotation), leaf(c.features)
    sink = torch.zeros_like(m3, requires_grad=True)
    img, radii = dgr.GaussianRasterizer(raster_settings=s)(means3D=m3, means2D=sink, shs=sh, colors_precomp=None,
                                                           opacities=opl, scales=sc, rotations=ro, cov3D_precomp=None)
    w = torch.randn(img.shape, device="cuda", generator=torch.Generator(device="cuda").manual_seed(1))
    (img * w).sum().backward()
    vis_dead = dead & (radii > 0)
    assert int(vis_dead.sum()) > 100
    for t in (m3, opl, sc, ro, sh, sink):
        assert bool(torch.isfinite(t.grad).all())
    assert float(opl.grad[vis_dead].abs().max()) == 0.0
    assert float(sink.grad[vis_dead].abs().max()) == 0.0 and float(m3.grad[vis_dead].abs().max()) == 0.0
    assert float(opl.grad[~dead & (radii > 0)].abs().max()) > 0.0

    # the image is the one rendered without the dead Gaussians at all
    keep = ~dead
    st = dgr.rasterize_forward(s, c.xyz[keep].contiguous(), op[keep].contiguous(), shs=c.features[keep].contiguous(),
                               scales=c.scaling[keep].contiguous(), rotations=c.rotation[keep].contiguous())
    assert torch.equal(st.color, img.detach())

    model = GaussianParams(cloud, max_sh_degree=2, device="cuda")
    with torch.no_grad():
        model._opacity[dead] = -200.0
    out = render(cam.to("cuda"), model, PipelineParams(fused_activations=True), bg.cuda())
    (out["render"] * w).sum().backward()
    for n, p in model.named_parameters():
        if p.grad is not None:
            assert bool(torch.isfinite(p.grad).all()), n
    assert float(model._opacity.grad[dead].abs().max()) == 0.0
    assert float((out["render"].detach() - img.detach()).abs().mean()) < 1e-5      # exp(log(s)) differs from s by ulps


def test_the_reference_gradient_sink_idiom():
    """gaussian_renderer/__init__.py:26-30: the sink is a NON-leaf (zeros_like(...) + 0) with retain_grad();
    add_densification_stats reads its .grad[:, :2] (scene/gaussian_model.py:405-407)."""
    import diff_gaussian_rasterization as dgr
    cloud, cam, bg, _ = make_scene(5_000, 240, 160, sh_degree=3, seed=8)
    s = _gpu_settings(cam, bg, 3, 1.0)
    c = cloud.to("cuda")
    xyz = c.xyz.clone().requires_grad_(True)
    grads = {}
    for idiom in ("reference", "leaf"):
        if idiom == "reference":
            screenspace_points = torch.zeros_like(xyz, dtype=xyz.dtype, requires_grad=True, device="cuda") + 0
            try:
                screenspace_points.retain_grad()
            except Exception:          # as the reference: :27-30
                pass
            assert not screenspace_points.is_leaf
        else:
            screenspace_points = torch.zeros_like(xyz, requires_grad=True)
        img, radii = dgr.GaussianRasterizer(raster_settings=s)(
            means3D=xyz, means2D=screenspace_points, shs=c.features, colors_precomp=None, opacities=c.opacity,
            scales=c.scaling, rotations=c.rotation, cov3D_precomp=None)
        xyz.grad = None
        img.square().sum().backward()
        g = screenspace_points.grad
        assert g is not None and g.shape == (5_000, 3)
        grads[idiom] = g.clone()
        vis = radii > 0
        norm = torch.norm(g[vis, :2], dim=-1, keepdim=True)          # what the densifier accumulates
        assert bool(torch.isfinite(norm).all()) and float(norm.max()) > 0.0
        assert float(g[~vis].abs().max()) == 0.0 and float(g[:, 2].abs().max()) == 0.0
    scale = float(grads["leaf"].abs().max())
    assert float((grads["reference"] - grads["leaf"]).abs().max()) <= 2e-5 * scale      # float atomics: order differs


def _forward(dgr, s, c, **kw):
    return dgr.rasterize_forward(s, c.xyz, c.opacity, shs=c.features, scales=c.scaling, rotations=c.rotation, **kw)


def test_list_builder_refuses_a_failed_sort_and_foreign_sizes():
    """gsr_bin_lists with the guards of gsr.h, driven directly: a non-zero sort-error word, or list sizes
    that differ from the device's own, make it write no list: all ranges (0,0), the status word says why,
    and nothing outside the output buffers is touched (canaries)."""
    import diff_gaussian_rasterization as dgr
    lib, ptr = dgr._lib, dgr._ptr
    cloud, cam, bg, _ = make_scene(20_000, 640, 360, sh_degree=0, seed=4, coeffs=1)
    s = _gpu_settings(cam, bg, 0, 1.0)
    c = cloud.to("cuda")
    st = _forward(dgr, s, c)
    torch.cuda.synchronize()
    N, V = st.num_rendered, st.num_visible
    # the device's totals of this scene, recomputed through the ABI
    P, W, H = st.P, st.W, st.H
    u32 = dict(dtype=torch.int32, device="cuda")
    totals = torch.empty((lib.gsr_totals_words(P),), **u32)
    records, radii, keys = torch.empty((P, 16), **u32), torch.empty((P,), **u32), torch.empty((P,), **u32)
    sort_vals = torch.empty((P,), dtype=torch.int64, device="cuda")
    dgr._run("preprocess_fwd", lib.gsr_preprocess_fwd, P, 0, 1, ptr(c.xyz), ptr(c.scaling), 1.0, ptr(c.rotation),
             ptr(c.opacity), ptr(c.features), None, None, ptr(st.view), ptr(st.proj), ptr(st.campos), W, H,
             float(s.tanfovx), float(s.tanfovy), 0, int(dgr.REFERENCE_LISTS), ptr(records), ptr(radii), None, ptr(keys), None, ptr(sort_vals),
             ptr(totals), None, None, dgr._stream())
    n, e1, e2, v = (int(x) for x in totals[:4].tolist())
    assert (n, v) == (N, V) and int(totals[4]) == 0
    tiles = ((W + 15) // 16) * ((H + 15) // 16)

    def build(V_, N_, E1_, E2_, error_value):
        nb = lib.gsr_bin_lists_bytes(V_, E1_, E2_, W, H)
        scratch = torch.empty((nb,), dtype=torch.uint8, device="cuda")
        pad = 1024
        plist = torch.full((N_ + 2 * pad,), 0x5A5A5A5A, **u32)
        ranges = torch.full((2 * tiles + 2 * pad,), 0x5A5A5A5A, **u32)
        err = torch.tensor([error_value], **u32)
        status = torch.zeros((1,), **u32)
        dgr._run("bin_lists", lib.gsr_bin_lists, P, V_, N_, E1_, E2_, W, H, ptr(st.sorted_vals), ptr(plist[pad:]),
                 ptr(ranges[pad:]), None, ptr(err), ptr(totals), ptr(status), ptr(scratch), nb, dgr._stream())
        torch.cuda.synchronize()
        for buf in (plist, ranges):
            assert bool((buf[:pad] == 0x5A5A5A5A).all()) and bool((buf[-pad:] == 0x5A5A5A5A).all())
        return plist[pad:pad + N_], ranges[pad:pad + 2 * tiles].reshape(tiles, 2), int(status[0])

    plist, ranges, status = build(v, n, e1, e2, 0)
    assert status == 0 and torch.equal(plist, st.point_list) and torch.equal(ranges, st.ranges)
    plist, ranges, status = build(v, n, e1, e2, 4)                       # "pass 2 gave up"
    assert status == dgr.STATUS_SORT and int(ranges.abs().sum()) == 0
    assert bool((plist == 0x5A5A5A5A).all())                             # no list written at all
    plist, ranges, status = build(v - 1, n - 1, e1 - 1, e2 - 1, 0)       # somebody else's sizes (smaller buffers)
    assert status == dgr.STATUS_TOTALS and int(ranges.abs().sum()) == 0
    assert bool((plist == 0x5A5A5A5A).all())
    plist, ranges, status = build(v, n + 5, e1, e2, 1)
    assert status == (dgr.STATUS_SORT | dgr.STATUS_TOTALS)


def test_a_sort_that_gives_up_in_release_mode_surfaces_at_the_next_forward():
    """debug=False: nobody waits for the sort's status.  With a patience of one look-back step the 1M-key
    sort of this scene (almost certainly) gives up; the forward then renders the background, and the
    NEXT forward on this thread raises RasterizerFault from the read-back it does anyway.  After that
    the status is clear and rendering works again."""
    import diff_gaussian_rasterization as dgr
    cloud, cam, bg, _ = make_scene(1_000_000, 640, 360, sh_degree=0, seed=6, bg=(0.25, 0.5, 0.75), coeffs=1)
    s = _gpu_settings(cam, bg, 0, 1.0)
    c = cloud.to("cuda")
    good = _forward(dgr, s, c)
    torch.cuda.synchronize()
    dgr._lib.gsr_debug_sort_spin_limit(1)
    try:
        shaky = _forward(dgr, s, c)
        torch.cuda.synchronize()
    finally:
        dgr._lib.gsr_debug_sort_spin_limit(0)
    status = int(dgr._status_word(c.xyz.device)[0])
    print("status after the impatient sort:", status, flush=True)
    if status == 0:
        assert torch.equal(shaky.color, good.color)                      # no workgroup happened to wait
        again = _forward(dgr, s, c)
    else:
        assert status == dgr.STATUS_SORT
        want = bg.cuda()[:, None, None].expand_as(shaky.color)
        assert torch.equal(shaky.color, want) and int(shaky.n_contrib.sum()) == 0
        g = dgr.rasterize_backward(s, shaky, torch.ones_like(shaky.color), c.xyz, shs=c.features, scales=c.scaling,
                                   rotations=c.rotation)
        assert all(float(v.abs().max()) == 0.0 for v in g.values() if v is not None)
        with pytest.raises(dgr.RasterizerFault, match="GSR_E_SORT"):
            _forward(dgr, s, c)
        assert int(dgr._status_word(c.xyz.device)[0]) == 0
        again = _forward(dgr, s, c)
    torch.cuda.synchronize()
    assert torch.equal(again.color, good.color) and torch.equal(again.point_list, good.point_list)


@pytest.mark.parametrize("raw", [False, True])
def test_single_call_forward_equals_the_stage_calls(raw):
    """gsr_forward (one call, one workspace, lists through the resize callback) against the sequence of
    stage entry points: every output and every intermediate buffer bit for bit, and the same gradients."""
    import diff_gaussian_rasterization as dgr
    from gsplat_host.model import GaussianParams
    cloud, cam, bg, _ = make_scene(40_000, 500, 300, sh_degree=3, seed=12, bg=(0.1, 0.2, 0.3))
    s = _gpu_settings(cam, bg, 3, 1.0)
    c = cloud.to("cuda")
    kw = dict(keep_tile_ids=True, prepare_backward=True)
    if raw:
        m = GaussianParams(cloud, device="cuda")
        rawd = dict(features_dc=m._features_dc.detach(), features_rest=m._features_rest.detach(),
                    scaling=m._scaling.detach(), rotation=m._rotation.detach(), opacity=m._opacity.detach())
        one = dgr.rasterize_forward(s, c.xyz, None, raw=rawd, single_call=True, **kw)
        many = dgr.rasterize_forward(s, c.xyz, None, raw=rawd, single_call=False, **kw)
    else:
        one = _forward(dgr, s, c, single_call=True, **kw)
        many = _forward(dgr, s, c, single_call=False, **kw)
    torch.cuda.synchronize()
    assert one.num_rendered == many.num_rendered > 0 and one.num_visible == many.num_visible
    vis = many.radii > 0
    for name in ("color", "radii", "visible", "final_T", "n_contrib", "ranges", "point_list", "tile_ids_sorted",
                 "depth_keys"):
        assert torch.equal(getattr(one, name), getattr(many, name)), name
    assert torch.equal(one.records[vis], many.records[vis])
    V = many.num_visible
    assert torch.equal(one.sorted_vals[:V], many.sorted_vals[:V])
    d = torch.randn(one.color.shape, device="cuda", generator=torch.Generator(device="cuda").manual_seed(4))
    if raw:
        g1 = dgr.rasterize_backward_raw(s, one, d, c.xyz, rawd)
        g2 = dgr.rasterize_backward_raw(s, many, d, c.xyz, rawd)
    else:
        g1 = dgr.rasterize_backward(s, one, d, c.xyz, shs=c.features, scales=c.scaling, rotations=c.rotation)
        g2 = dgr.rasterize_backward(s, many, d, c.xyz, shs=c.features, scales=c.scaling, rotations=c.rotation)
    for k in g1:
        if g1[k] is not None:
            assert float((g1[k] - g2[k]).abs().max()) <= 2e-5 * float(g2[k].abs().max() + 1e-20), k

    # empty input and a fully culled cloud through the single call
    behind = c.xyz.clone()
    behind[:, 2] += 100.0
    st = dgr.rasterize_forward(s, behind, c.opacity, shs=c.features, scales=c.scaling, rotations=c.rotation,
                               single_call=True)
    torch.cuda.synchronize()
    assert st.num_rendered == 0 and torch.equal(st.color, bg.cuda()[:, None, None].expand_as(st.color))


def test_the_blend_kernels_clear_what_the_backward_wants_zeroed():
    """prepare_backward through the single call: the (P,16) accumulator of the blend backward is gsr_forward's
    zero-fill, and the gradient tensors carved from the same allocation are the zero-fill of gsr_blend_bwd_fill.
    Whatever the block held before (the allocator's cache is poisoned first), the accumulator is all zeros when
    the forward's work is done and the gradient tensors are when the blend backward's is; also when no list is
    built (everything culled): then no blend backward runs and the per-Gaussian backward writes every row."""
    import diff_gaussian_rasterization as dgr
    if not dgr._CARRIED_FILLS or dgr._OVERLAP:
        pytest.skip("the fills are switched to kernels of their own (GSR_SEPARATE_FILLS / GSR_OVERLAP)")
    cloud, cam, bg, _ = make_scene(30_001, 500, 300, sh_degree=0, seed=21, coeffs=1)
    s = _gpu_settings(cam, bg, 0, 1.0)
    c = cloud.to("cuda")
    behind = c.xyz.clone()
    behind[:, 2] += 100.0
    for xyz, rendered in ((c.xyz, True), (behind, False)):
        poison = torch.full((c.xyz.shape[0], 48), float("nan"), device="cuda")
        del poison                                     # back to the caching allocator, still NaN
        st = dgr.rasterize_forward(s, xyz, c.opacity, shs=c.features, scales=c.scaling, rotations=c.rotation,
                                   single_call=True, prepare_backward=True)
        torch.cuda.synchronize()
        assert (st.num_rendered > 0) == rendered
        assert st.dsplat is not None and st.dsplat.shape == (c.xyz.shape[0], 16)
        assert int(st.dsplat.view(torch.int32).abs().max()) == 0
        held = st.zeroed_grads
        assert held is not None and not held.cleared
        assert held.touched.shape[0] >= c.xyz.shape[0] and int(held.touched.max()) == 0   # cleared with the accumulator
        dsplat = dgr.blend_backward(st, torch.zeros_like(st.color))
        torch.cuda.synchronize()
        assert held.cleared == rendered
        if rendered:
            for name, v in held.views.items():
                assert int(v.view(torch.int32).abs().max()) == 0, name
        g = dgr.rasterize_backward(s, st, torch.zeros_like(st.color), xyz, shs=c.features, scales=c.scaling,
                                   rotations=c.rotation, dsplat=dsplat)
        for name, v in g.items():
            if v is not None:
                assert int(v.view(torch.int32).abs().max()) == 0, name


def test_blend_backward_with_a_zero_fill_through_the_c_abi():
    """gsr_blend_bwd_fill against gsr_blend_bwd: the same sums (up to the order of the atomic additions), the
    fill is all zeros, for sizes that do not divide by the number of workgroups and for one smaller than it, and
    every Gaussian with a non-zero row of sums has its `touched` byte set (no byte beyond the P is written)."""
    import ctypes as C
    import diff_gaussian_rasterization as dgr
    cloud, cam, bg, _ = make_scene(20_000, 333, 222, sh_degree=0, seed=5, coeffs=1)
    s = _gpu_settings(cam, bg, 0, 1.0)
    c = cloud.to("cuda")
    st = dgr.rasterize_forward(s, c.xyz, c.opacity, shs=c.features, scales=c.scaling, rotations=c.rotation)
    d = torch.rand_like(st.color)
    lib, p = dgr._lib, lambda t: C.c_void_p(t.data_ptr())
    stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    common = (st.W, st.H, p(st.ranges), p(st.point_list), p(st.records), p(st.bg), p(st.final_T), p(st.n_contrib), p(d))
    plain = torch.zeros((st.P, 16), device="cuda")
    assert lib.gsr_blend_bwd(*common, p(plain), stream) == 0
    for nbytes in (16, 16 * 77, 16 * 1_000_003):
        guard = 64
        buf = torch.full((nbytes // 4 + 2 * guard,), float("nan"), device="cuda")
        got = torch.zeros((st.P, 16), device="cuda")
        touched = torch.zeros((st.P + guard,), dtype=torch.uint8, device="cuda")
        rc = lib.gsr_blend_bwd_fill(*common, p(got), p(touched), C.c_void_p(buf.data_ptr() + 4 * guard), nbytes, None, None,
                                    stream)
        assert rc == 0, lib.gsr_last_error().decode()
        torch.cuda.synchronize()
        has_sums = (got != 0).any(dim=1)
        assert int(has_sums.sum()) > 100 and bool((touched[:st.P][has_sums] == 1).all())
        assert int(touched.max()) == 1 and int(touched[st.P:].max()) == 0
        assert int(touched.sum()) < st.P                     # hidden and culled Gaussians stay unmarked
        assert int(buf[guard:-guard].view(torch.int32).abs().max()) == 0
        assert bool(torch.isnan(buf[:guard]).all()) and bool(torch.isnan(buf[-guard:]).all())   # nothing beyond it
        assert float((got - plain).abs().max()) <= 1e-5 * float(plain.abs().max())   # atomic additions in another order
    # misaligned or ragged fills are refused
    buf = torch.zeros((64,), device="cuda")
    assert lib.gsr_blend_bwd_fill(*common, p(plain), None, C.c_void_p(buf.data_ptr() + 4), 16, None, None, stream) != 0
    assert lib.gsr_blend_bwd_fill(*common, p(plain), None, p(buf), 24, None, None, stream) != 0


def test_two_host_threads_on_their_own_streams():
    """Two host threads render different scenes at the same time, each on its own stream (forward through the
    single call, backward after it).  The library keeps no state between calls and the binding's landing
    buffer, status word, event and callbacks are per thread: every result equals the one the scene gives
    alone."""
    import threading
    import diff_gaussian_rasterization as dgr
    scenes = []
    for seed, (W, H) in ((41, (400, 240)), (42, (320, 320))):
        cloud, cam, bg, _ = make_scene(30_000, W, H, sh_degree=1, seed=seed)
        s = _gpu_settings(cam, bg, 1, 1.0)
        scenes.append((s, cloud.to("cuda")))

    def run(s, c):
        st = dgr.rasterize_forward(s, c.xyz, c.opacity, shs=c.features, scales=c.scaling, rotations=c.rotation,
                                   prepare_backward=True)
        d = torch.ones_like(st.color)
        g = dgr.rasterize_backward(s, st, d, c.xyz, shs=c.features, scales=c.scaling, rotations=c.rotation)
        return st.color.clone(), st.point_list.clone(), g["means3D"].clone()

    alone = [run(s, c) for s, c in scenes]
    torch.cuda.synchronize()
    got = [[] for _ in scenes]
    errors = []

    def worker(k):
        try:
            stream = torch.cuda.Stream()
            with torch.cuda.stream(stream):
                for _ in range(20):
                    got[k].append(run(*scenes[k]))
            stream.synchronize()
        except Exception as e:      # noqa: BLE001
            errors.append(e)

    threads = [threading.Thread(target=worker, args=(k,)) for k in range(len(scenes))]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    torch.cuda.synchronize()
    assert not errors, errors
    for k, (color, plist, gmean) in enumerate(alone):
        for c2, p2, g2 in got[k]:
            assert torch.equal(c2, color) and torch.equal(p2, plist)
            assert float((g2 - gmean).abs().max()) <= 2e-5 * float(gmean.abs().max())


@pytest.mark.parametrize("raw", [False, True])
def test_backward_into_cleared_gradients_equals_the_dense_backward(raw):
    """prepare_backward through the single call also clears the gradient tensors of the per-Gaussian backward,
    which then writes only the rows of Gaussians that have gradients (outputs_zeroed).  Same numbers as the
    backward that writes every row, for the reference's inputs and for raw parameters; Gaussians that are
    culled or hidden have exact zeros either way (the allocator's cache is poisoned first)."""
    import diff_gaussian_rasterization as dgr
    from gsplat_host.model import GaussianParams
    if not dgr._CARRIED_FILLS or dgr._OVERLAP:
        pytest.skip("the fills are switched to kernels of their own (GSR_SEPARATE_FILLS / GSR_OVERLAP)")
    cloud, cam, bg, _ = make_scene(60_000, 480, 272, sh_degree=3, seed=17, bg=(0.2, 0.1, 0.0))
    s = _gpu_settings(cam, bg, 3, 1.0)
    c = cloud.to("cuda")
    m = GaussianParams(cloud, device="cuda")
    rawd = dict(features_dc=m._features_dc.detach(), features_rest=m._features_rest.detach(),
                scaling=m._scaling.detach(), rotation=m._rotation.detach(), opacity=m._opacity.detach())
    d = torch.randn((3, 272, 480), device="cuda", generator=torch.Generator(device="cuda").manual_seed(9))

    def forward(prepare):
        poison = torch.full((60_000 * 80,), float("nan"), device="cuda")
        del poison
        if raw:
            return dgr.rasterize_forward(s, c.xyz, None, raw=rawd, prepare_backward=prepare, single_call=True)
        return dgr.rasterize_forward(s, c.xyz, c.opacity, shs=c.features, scales=c.scaling, rotations=c.rotation,
                                     prepare_backward=prepare, single_call=True)

    def backward(st, sums):
        if raw:
            return dgr.rasterize_backward_raw(s, st, d, c.xyz, rawd, dsplat=sums)
        return dgr.rasterize_backward(s, st, d, c.xyz, shs=c.features, scales=c.scaling, rotations=c.rotation, dsplat=sums)

    prepared, plain = forward(True), forward(False)
    assert prepared.zeroed_grads is not None and plain.zeroed_grads is None
    sums = dgr.blend_backward(prepared, d)            # one set of sums for both: float atomics land in any order
    sparse = backward(prepared, sums)
    assert prepared.zeroed_grads is None              # handed out once
    dense = backward(plain, sums)
    torch.cuda.synchronize()
    some = 0
    for k, v in dense.items():
        if v is None:
            assert sparse[k] is None
            continue
        assert torch.equal(sparse[k], v), k           # the same arithmetic, only the untouched rows are not written
        some += int((v.reshape(v.shape[0], -1).abs().sum(1) != 0).sum())
    assert some > 0
    again = backward(prepared, sums)                  # a second backward of the same state: the dense way
    for k, v in dense.items():
        assert v is None or torch.equal(again[k], v), k


def test_the_single_call_with_a_pageable_landing_buffer():
    """gsr_forward's read-back of the list sizes has two ways: a landing buffer the GPU can address (pinned: the
    depth sort's first kernel sums and stores the sizes, the call polls for its ticket) and one it cannot
    (pageable: the preprocess sums them, a copy command and an event carry them).  Same lists, same image."""
    import diff_gaussian_rasterization as dgr
    cloud, cam, bg, _ = make_scene(25_000, 400, 300, sh_degree=2, seed=12, coeffs=9)
    s = _gpu_settings(cam, bg, 2, 1.0)
    c = cloud.to("cuda")
    pinned = _forward(dgr, s, c, single_call=True)
    slot = dgr._per_device(torch.device("cuda", torch.cuda.current_device()))
    kept = slot["pinned"]
    assert kept.is_pinned()
    slot["pinned"] = torch.full((dgr.TOTALS_HEAD,), -1, dtype=torch.int32)      # ordinary host memory
    try:
        pageable = _forward(dgr, s, c, single_call=True)
        landed = slot["pinned"].tolist()
    finally:
        slot["pinned"] = kept
    assert landed[0] == pageable.num_rendered and landed[3] == pageable.num_visible
    assert pageable.num_rendered == pinned.num_rendered > 0 and pageable.num_visible == pinned.num_visible
    assert torch.equal(pageable.point_list, pinned.point_list) and torch.equal(pageable.ranges, pinned.ranges)
    assert torch.equal(pageable.color, pinned.color)
    # the pinned way leaves its ticket behind the sizes
    again = _forward(dgr, s, c, single_call=True)
    assert int(kept[0]) == again.num_rendered and int(kept[7]) != 0


def test_two_prepared_forwards_alive_at_once_backward_in_reverse_order():
    """Two forwards with prepared backwards (each with its own block of accumulator, flags and cleared gradients)
    are alive at the same time and their backwards run in the opposite order, on a stream of their own: every
    gradient equals the one its scene gives alone, bit for bit where the sums are shared, and the second
    backward of a state (which finds nothing prepared) still gives the same numbers."""
    import diff_gaussian_rasterization as dgr
    scenes = []
    for seed, (W, H), P in ((51, (384, 216), 40_000), (52, (300, 300), 25_000)):
        cloud, cam, bg, _ = make_scene(P, W, H, sh_degree=2, seed=seed, coeffs=9)
        scenes.append((_gpu_settings(cam, bg, 2, 1.0), cloud.to("cuda")))

    def forward(s, c):
        return dgr.rasterize_forward(s, c.xyz, c.opacity, shs=c.features, scales=c.scaling, rotations=c.rotation,
                                     single_call=True, prepare_backward=True)

    def backward(s, c, st, d, sums=None):
        return dgr.rasterize_backward(s, st, d, c.xyz, shs=c.features, scales=c.scaling, rotations=c.rotation, dsplat=sums)

    stream = torch.cuda.Stream()
    with torch.cuda.stream(stream):
        states = [forward(s, c) for s, c in scenes]
        ds = [torch.randn_like(st.color) for st in states]
        sums = [None, None]
        grads = [None, None]
        for k in (1, 0):                                           # reverse order
            s, c = scenes[k]
            sums[k] = dgr.blend_backward(states[k], ds[k])
            grads[k] = backward(s, c, states[k], ds[k], sums[k])
        again = [backward(scenes[k][0], scenes[k][1], states[k], ds[k], sums[k]) for k in (0, 1)]
        alone = []
        for k in (0, 1):
            s, c = scenes[k]
            st = dgr.rasterize_forward(s, c.xyz, c.opacity, shs=c.features, scales=c.scaling, rotations=c.rotation,
                                       single_call=False)
            assert torch.equal(st.color, states[k].color) and torch.equal(st.point_list, states[k].point_list)
            alone.append(backward(s, c, st, ds[k], sums[k]))
    stream.synchronize()
    for k in (0, 1):
        for name, v in alone[k].items():
            if v is None:
                continue
            assert torch.equal(grads[k][name], v), (k, name)
            assert torch.equal(again[k][name], v), (k, name)
        assert float(alone[k]["means3D"].abs().max()) > 0.0


@pytest.mark.parametrize("P", [300_000, 600_000])
def test_a_frame_with_more_instances_than_32_bits_hold_is_refused(P):
    """The list sizes are 32-bit sums, as the replaced module's instance count is an int.  Gaussians that have grown
    until each covers the whole 1080p frame (8160 tiles) bring 2.4e9 instances at 300k, and 4.9e9 at 600k, where the
    sum has wrapped to a harmless-looking 0.6e9: sizing the lists by it would let the list builder write past them.
    The forward stops before any list is sized, on the single call (both read-backs) and on the stage calls, and
    the next frame is not affected."""
    import diff_gaussian_rasterization as dgr
    cloud, cam, bg, _ = make_scene(P, 1920, 1080, sh_degree=0, seed=3, coeffs=1)
    s = _gpu_settings(cam, bg, 0, 1.0)
    c = cloud.to("cuda")
    before = _forward(dgr, s, c, single_call=True)
    huge = torch.full_like(c.scaling, 60.0)
    opaque = torch.full_like(c.opacity, 0.9)

    def grown(**kw):
        return dgr.rasterize_forward(s, c.xyz, opaque, shs=c.features, scales=huge, rotations=c.rotation, **kw)

    with pytest.raises(RuntimeError, match="tile instances"):
        grown(single_call=True)
    slot = dgr._per_device(torch.device("cuda", torch.cuda.current_device()))
    kept = slot["pinned"]
    slot["pinned"] = torch.full((dgr.TOTALS_HEAD,), -1, dtype=torch.int32)      # ordinary host memory: the other read-back
    try:
        with pytest.raises(RuntimeError, match="tile instances"):
            grown(single_call=True)
    finally:
        slot["pinned"] = kept
    with pytest.raises(RuntimeError, match="tile instances"):
        grown(single_call=False)
    torch.cuda.synchronize()
    for single in (True, False):
        after = _forward(dgr, s, c, single_call=single)
        assert after.num_rendered == before.num_rendered > 0
        assert torch.equal(after.point_list, before.point_list) and torch.equal(after.color, before.color)
    # just how many there were: every Gaussian in front of the near plane, on every tile
    z = (torch.cat((cloud.xyz, torch.ones(P, 1)), dim=1) @ cam.world_view_transform)[:, 2]
    assert int((z > 0.2).sum()) * 8160 >= dgr.MAX_INSTANCES


def test_parameters_that_have_diverged_do_no_harm_beyond_their_own_pixels(oracle32):
    """A training run that diverges hands the rasterizer NaN, infinities and 1e38 in any parameter.  Such a Gaussian
    is culled or binned exactly as the oracle does it (a NaN covariance gives the largest radius there is: the
    whole frame), the lists stay bit-exact, every kernel ends, and the next clean frame is what it was."""
    import diff_gaussian_rasterization as dgr
    from scenes import cloud_numpy, oracle_settings
    P = 3000
    cloud, cam, bg, _ = make_scene(P, 200, 150, sh_degree=1, seed=5, coeffs=4)
    s = _gpu_settings(cam, bg, 1, 1.0)
    clean = _forward(dgr, s, cloud.to("cuda"), single_call=True)
    rng = np.random.default_rng(0)
    bad = [np.nan, np.inf, -np.inf, 1e38, -1e38, 1e-45]
    for t in (cloud.xyz, cloud.scaling, cloud.rotation, cloud.opacity, cloud.features):
        a = t.numpy()                                           # shares the tensor's memory
        for j, i in enumerate(rng.choice(P, 60, replace=False)):
            a[i].reshape(-1)[rng.integers(a[i].size)] = bad[j % len(bad)]
    ref = oracle32.forward(oracle_settings(cam, bg, 1, 1.0), **cloud_numpy(cloud))
    assert ref["pre"]["radii"].max() == 2147483647 and np.isnan(ref["fwd"]["color"]).any()
    c = cloud.to("cuda")
    for single in (True, False):
        st = _forward(dgr, s, c, single_call=single, reference_lists=True, keep_tile_ids=True)
        torch.cuda.synchronize()
        assert np.array_equal(st.radii.cpu().numpy(), ref["pre"]["radii"])
        assert st.num_rendered == ref["bin"]["n_dup"]
        assert np.array_equal(st.point_list.cpu().numpy().view(np.uint32), ref["bin"]["point_list"])
        assert np.array_equal(st.ranges.cpu().numpy().view(np.uint32), ref["bin"]["ranges"])
        # (the pixels themselves are not compared: the oracle, like the published kernel, blends a Gaussian whose
        # exponent is NaN, "power > 0" being false for it, and every pixel of this frame ends up NaN there; the
        # blend kernels ask "power <= 0" and leave such a Gaussian out)
    # the short lists, forward and backward: everything ends, the gradients of the untouched Gaussians are numbers
    dL = torch.randn((3, 150, 200), generator=torch.Generator().manual_seed(1)).cuda()
    for single in (True, False):
        st = _forward(dgr, s, c, single_call=single, prepare_backward=single)
        g = dgr.rasterize_backward(s, st, dL, c.xyz, shs=c.features, scales=c.scaling, rotations=c.rotation)
        torch.cuda.synchronize()
        assert all(v.shape[0] == P for v in g.values() if v is not None)
        # a Gaussian none of whose parameters was touched and that no poisoned one shares a pixel with cannot be told
        # apart here (the whole-frame ones share every pixel), so only this: the clean rows of the means' gradient
        # are not all NaN, i.e. the backward did its work around the poison
        assert bool(torch.isfinite(g["means3D"]).any())
    again = _forward(dgr, s, make_scene(P, 200, 150, sh_degree=1, seed=5, coeffs=4)[0].to("cuda"), single_call=True)
    assert torch.equal(again.color, clean.color) and torch.equal(again.point_list, clean.point_list)
