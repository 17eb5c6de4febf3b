"""Two processes on the one GPU, gloo between them (host tensors only): each renders its share of the
views with the HIP path, back-propagates, the losses and gradients are summed over the ranks, one fused
Adam step follows, and losses, gradients and updated parameters must equal one process doing all the
views.  On a small scene with four views and with the benchmark's eight views 45 degrees apart (1 x 8 against
2 x 4), and on the benchmark's multi-view workload at full size (1M Gaussians, 1920x1080, SH degree 3, one
view per rank).  The gradient exchange runs in each of its three forms.

The RCCL process group itself (backend "nccl") is formed by bench.py when it runs under a launcher; the last
test starts it that way as a child process, with one rank."""
import json
import subprocess
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

SCENES = {
    # name: (Gaussians, width, height, SH degree, views, degrees between views, cloud seed)
    "small_4_views": (3_000, 160, 96, 2, 4, 40.0, 9),
    # the benchmark's multi-view shape: 8 views 45 degrees apart (one rank does all 8, two ranks do 4 each)
    "small_8_views": (3_000, 160, 96, 2, 8, 45.0, 9),
    "c4_1M_1080p": (1_000_000, 1920, 1080, 3, 2, 45.0, 0),
}


# which form of the gradient exchange a scene's ranks use (all give the same sums)
EXCHANGE_OF = {"small_4_views": "all_reduce", "small_8_views": "scatter_gather", "c4_1M_1080p": "direct"}


def _setup_path():
    for p in (ROOT, os.path.join(ROOT, "gaussian-splatting-own_amd"), HERE):
        if p not in sys.path:
            sys.path.insert(0, p)


def _run_views(scene, rank, world):
    """This rank's share of the views, gradients reduced over the ranks, one optimiser step:
    (mean loss, view indices, factor left for the step, gradients, updated parameters) on the host."""
    _setup_path()
    from gsplat_host.cameras import orbit_camera
    from gsplat_host.loss import photometric_loss
    from gsplat_host.model import GaussianParams
    from gsplat_host.multiview import multiview_step
    from gsplat_host.optim import GaussianAdam, all_reduce_gradients
    from gsplat_host.render import PipelineParams, render
    from gsplat_host.synthetic import make_cloud
    P, W, H, deg, n_views, step_deg, seed = SCENES[scene]
    dev = torch.device("cuda", 0)
    model = GaussianParams(make_cloud(P, sh_degree=deg, seed=seed), max_sh_degree=deg, device=dev)
    cams = [orbit_camera(W, H, angle_deg=step_deg * k, device=dev) for k in range(n_views)]
    targets = [torch.rand(3, H, W, generator=torch.Generator().manual_seed(20 + k)).to(dev) for k in range(n_views)]
    bg = torch.zeros(3, device=dev)
    pipe = PipelineParams(fused_activations=True)
    opt = GaussianAdam.for_model(model)
    loss, mine = multiview_step(cams, lambda cam: render(cam, model, pipe, bg)["render"], photometric_loss, targets,
                                rank=rank, world=world)
    # multiview_step normalised every view's loss by the GLOBAL view count: the sum over ranks is the gradient
    scale = all_reduce_gradients(opt.params, average=False, method=EXCHANGE_OF.get(scene, "all_reduce"))
    torch.cuda.synchronize()
    grads = {k: p.grad.detach().cpu() for k, p in opt.params.items()}
    opt.step(grad_scale=scale)
    torch.cuda.synchronize()
    return float(loss), mine, scale, grads, {k: p.detach().cpu() for k, p in opt.params.items()}


def _worker(scene, rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        loss, mine, scale, grads, params = _run_views(scene, rank, world)
        if rank == 0:
            q.put((rank, loss, mine, scale, {k: v.numpy() for k, v in grads.items()},
                   {k: v.numpy() for k, v in params.items()}))
        else:       # the reduced gradients are the same words on every rank: one checksum per tensor is enough
            q.put((rank, loss, mine, scale, {k: float(v.double().sum()) for k, v in grads.items()}, None))
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("scene", list(SCENES))
def test_two_ranks_on_one_gpu_match_one_rank(scene):
    n_views = SCENES[scene][4]
    want_loss, mine, scale, want, want_params = _run_views(scene, 0, 1)
    assert mine == list(range(n_views)) and scale == 1.0
    torch.cuda.empty_cache()
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(scene, r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    results = sorted((q.get(timeout=600) for _ in procs), key=lambda r: r[0])
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert results[0][2] == list(range(0, n_views, 2)) and results[1][2] == list(range(1, n_views, 2))
    for rank, loss, _, scale, grads, params in results:
        assert scale == 1.0                                     # the losses were normalised over all views already
        assert abs(loss - want_loss) < 1e-6
    _, _, _, _, grads, params = results[0]
    for k, g in grads.items():
        ref = want[k].numpy()
        # float atomics: the summation order differs per run (seen: up to 2.2e-5 of the largest element on the
        # rotation gradient of the 1M-Gaussian scene, whose terms cancel)
        tol = 5e-5 * max(float(abs(ref).max()), 1e-12)
        err = float(abs(g - ref).max())
        print(scene, "grad", k, "max err %.2e of %.2e" % (err, float(abs(ref).max())), flush=True)
        assert err <= tol, k
        assert abs(results[1][4][k] - float(g.astype("float64").sum())) <= 1e-6 * max(1.0, float(abs(g).astype("float64").sum()))
    # one Adam step from the same state: the first step moves every element by lr * g / (|g| + eps), so the
    # parameters agree wherever the gradients' signs do; elements whose gradient is summation noise may flip
    for k, p in params.items():
        ref = want_params[k].numpy()
        frac = float((abs(p - ref) > 1e-7 * max(1.0, float(abs(ref).max()))).mean())
        print(scene, "param", k, "share of elements that differ %.2e" % frac, flush=True)
        assert frac < 1e-3, k


def test_bench_under_a_launcher_forms_the_rccl_group():
    """bench.py as torch.distributed.run starts it (RANK / WORLD_SIZE / MASTER_* in the environment), one rank:
    the process group is RCCL ("nccl"), the warm-up collective and the reduction of the mean loss before the
    closing barrier run on it, and the line says how many ranks the communicator has."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "3", "--warmup", "1",
                          "--no-cpu-baseline", "--no-variants", "--workload", "c2"], env=env, capture_output=True,
                         text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    line = json.loads(out.stdout.strip().splitlines()[-1])
    assert line["n_gpus"] == 1 and line["communicator"] == {"backend": "nccl", "ranks": 1}
    assert line["value"] > 0 and line["ms_per_step"] > 0 and line["mean_loss_over_ranks"] > 0
