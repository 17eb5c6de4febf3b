"""The N>1 path on CPU: two processes over gloo shard four views, render their
share (with the CPU oracle standing in for the GPU kernels), all-reduce the
scalar loss, and must agree with a single process doing all four views."""
import math
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


def _setup_path():
    for p in (ROOT, os.path.join(ROOT, "gaussian-splatting-own_amd"), HERE):
        if p not in sys.path:
            sys.path.insert(0, p)


def _scene():
    _setup_path()
    import oracle
    from gsplat_host.cameras import orbit_camera
    from gsplat_host.synthetic import make_cloud
    cloud = make_cloud(600, sh_degree=1, seed=4, coeffs=4)
    cams = [orbit_camera(96, 64, angle_deg=45.0 * k) for k in range(4)]
    targets = [torch.rand(3, 64, 96, generator=torch.Generator().manual_seed(10 + k)) for k in range(4)]
    o = oracle.Oracle(np.float32, threads=1)

    def render_view(cam):
        s = oracle.Settings(64, 96, math.tan(cam.FoVx / 2), math.tan(cam.FoVy / 2), np.zeros(3, np.float32), 1.0,
                            cam.world_view_transform.numpy(), cam.full_proj_transform.numpy(), 1,
                            cam.camera_center.numpy())
        out = o.forward(s, means3D=cloud.xyz.numpy(), opacities=cloud.opacity.numpy(), scales=cloud.scaling.numpy(),
                        rotations=cloud.rotation.numpy(), shs=cloud.features.numpy(), want_margin=False)
        return torch.from_numpy(out["color"])
    return cams, targets, render_view


def _worker(rank, world, port, q):
    _setup_path()
    from oracle.loss_ref import photometric_loss
    from gsplat_host.multiview import multiview_step
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    cams, targets, render_view = _scene()
    loss, mine = multiview_step(cams, render_view, photometric_loss, targets, rank=rank, world=world, backward=False)
    q.put((rank, float(loss), mine))
    dist.barrier()
    dist.destroy_process_group()


def test_view_sharding_is_a_partition():
    _setup_path()
    from gsplat_host.multiview import views_for_rank
    for world in (1, 2, 3, 8):
        seen = sorted(v for r in range(world) for v in views_for_rank(8, r, world))
        assert seen == list(range(8))
    assert views_for_rank(8, 3, 8) == [3]
    assert views_for_rank(2, 5, 8) == []
    with pytest.raises(ValueError):
        views_for_rank(8, 8, 8)


def test_two_ranks_match_one(tmp_path):
    _setup_path()
    from oracle.loss_ref import photometric_loss
    from gsplat_host.multiview import multiview_step
    cams, targets, render_view = _scene()
    want, mine = multiview_step(cams, render_view, photometric_loss, targets, rank=0, world=1, backward=False)
    assert mine == [0, 1, 2, 3]

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    results = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert results[0][2] == [0, 2] and results[1][2] == [1, 3]
    assert abs(results[0][1] - results[1][1]) < 1e-7          # both ranks hold the same reduced loss
    assert abs(results[0][1] - float(want)) < 1e-6


def _arena_worker(rank, world, port, q):
    _setup_path()
    from gsplat_host.optim import GradArena
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    params = {"xyz": torch.zeros(7, 3), "opacity": torch.zeros(7, 1), "rotation": torch.zeros(7, 4)}
    arena = GradArena(params)
    arena.attach(params)
    g = torch.Generator().manual_seed(100 + rank)
    for k, p in params.items():
        p.grad.copy_(torch.randn(p.shape, generator=g))          # what this rank's backward left behind
    separate = {k: torch.nn.Parameter(torch.zeros_like(p)) for k, p in params.items()}
    for k, p in separate.items():
        p.grad = params[k].grad.clone()                          # the same gradients as free-standing tensors
    from gsplat_host.optim import EXCHANGES, all_reduce_gradients
    before = arena.flat.clone()
    assert arena.flat.numel() % 1024 == 0                        # splits evenly over 2, 4, 8 ... ranks
    sums = {}
    for method in EXCHANGES:                                     # reduce-scatter + all-gather, its point-to-point form, one all-reduce
        arena.flat.copy_(before)
        scale = arena.all_reduce(method=method)
        sums[method] = arena.flat.clone()
    for method in EXCHANGES:
        assert torch.equal(sums[method], sums["all_reduce"]), method   # two addends: the same bits whatever the route
    assert float(sums["all_reduce"].abs().sum()) > 0
    for method in EXCHANGES:                                     # and on free-standing gradient tensors
        for k, p in separate.items():
            p.grad = before[arena.offsets[k]:arena.offsets[k] + p.numel()].view_as(p).clone()
        assert all_reduce_gradients(separate, method=method) == scale
        for k in params:
            assert torch.equal(separate[k].grad, params[k].grad), (method, k)
    # as numpy arrays: a tensor in a queue travels as a shared-memory handle that the receiver must fetch from this
    # process while it is still alive, which a worker that finishes first does not wait for
    q.put((rank, scale, {k: (p.grad * scale).numpy().copy() for k, p in params.items()}))
    # losses already normalised over all ranks' views: the sum is the gradient, nothing left to scale
    assert all_reduce_gradients({}, average=False) == 1.0 and all_reduce_gradients({}) == 0.5
    dist.barrier()
    dist.destroy_process_group()


def test_gradient_arena_all_reduce_gives_the_mean_gradient():
    """Data-parallel training: every rank's gradients sit in one flat buffer, one all-reduce sums
    them, and the factor handed to the optimiser step turns the sum into the mean."""
    _setup_path()
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_arena_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    results = sorted((q.get(timeout=120) for _ in procs), key=lambda r: r[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    want = {}
    for k, shape in (("xyz", (7, 3)), ("opacity", (7, 1)), ("rotation", (7, 4))):
        want[k] = 0
    for rank in range(2):
        g = torch.Generator().manual_seed(100 + rank)
        for k, shape in (("xyz", (7, 3)), ("opacity", (7, 1)), ("rotation", (7, 4))):
            want[k] = want[k] + torch.randn(shape, generator=g) / 2
    for rank, scale, grads in results:
        assert scale == 0.5
        for k in want:
            assert torch.allclose(torch.from_numpy(grads[k]), want[k], atol=1e-6), (rank, k)
