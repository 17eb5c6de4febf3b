"""Two processes on the one GPU, gloo between them (host tensors only): each renders its share of
four views with the HIP path, back-propagates, the losses and gradients are summed over the
ranks, and must equal one process doing all four views."""
import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


def _setup_path():
    for p in (ROOT, os.path.join(ROOT, "gaussian-splatting-own_amd"), HERE):
        if p not in sys.path:
            sys.path.insert(0, p)


def _run_views(rank, world):
    """(mean loss, {name: summed gradient on the host}) of this rank's share, reduced over ranks."""
    _setup_path()
    from gsplat_host.cameras import orbit_camera
    from gsplat_host.loss import photometric_loss
    from gsplat_host.model import GaussianParams
    from gsplat_host.multiview import multiview_step
    from gsplat_host.optim import GaussianAdam, all_reduce_gradients
    from gsplat_host.render import PipelineParams, render
    from gsplat_host.synthetic import make_cloud
    dev = torch.device("cuda", 0)
    model = GaussianParams(make_cloud(3_000, sh_degree=2, seed=9), max_sh_degree=2, device=dev)
    cams = [orbit_camera(160, 96, angle_deg=40.0 * k, device=dev) for k in range(4)]
    targets = [torch.rand(3, 96, 160, generator=torch.Generator().manual_seed(20 + k)).to(dev) for k in range(4)]
    bg = torch.zeros(3, device=dev)
    pipe = PipelineParams(fused_activations=True)
    opt = GaussianAdam.for_model(model)
    loss, mine = multiview_step(cams, lambda cam: render(cam, model, pipe, bg)["render"], photometric_loss, targets,
                                rank=rank, world=world)
    scale = all_reduce_gradients(opt.params)
    torch.cuda.synchronize()
    return float(loss), mine, scale, {k: p.grad.detach().cpu() for k, p in opt.params.items()}


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        loss, mine, scale, grads = _run_views(rank, world)
        q.put((rank, loss, mine, scale, {k: v.numpy() for k, v in grads.items()}))
        dist.barrier()
    finally:
        dist.destroy_process_group()


def test_two_ranks_on_one_gpu_match_one_rank():
    want_loss, mine, scale, want = _run_views(0, 1)
    assert mine == [0, 1, 2, 3] and scale == 1.0
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    results = sorted((q.get(timeout=240) for _ in procs), key=lambda r: r[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert results[0][2] == [0, 2] and results[1][2] == [1, 3]
    for rank, loss, _, scale, grads in results:
        assert scale == 0.5
        assert abs(loss - want_loss) < 1e-6
        for k, g in grads.items():
            ref = want[k].numpy()
            tol = 2e-5 * max(float(abs(ref).max()), 1e-12)      # float atomics: the summation order differs per run
            assert float(abs(g - ref).max()) <= tol, (rank, k)
