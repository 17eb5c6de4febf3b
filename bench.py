#!/usr/bin/env python3
"""Benchmark of the MI355X Gaussian rasterizer on the reference's training step.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload c3]

One "step" is the reference's own timed span of a training iteration
(train.py:67-95): model activations -> render() -> 0.8*L1 + 0.2*(1-SSIM) ->
backward.  The optimiser step is outside that span in the reference
(train.py:126-128) and outside it here.  The default workload is the headline
configuration: 1M synthetic Gaussians, 1920x1080, SH degree 3 (see
gsplat_host/synthetic.py for the seeded recipe).  With --gpus N (launched by
torch.distributed.run, one rank per GPU) every rank renders its own view of the
same replicated cloud (camera rotated by rank*45 degrees) and the scalar loss is
all-reduced over RCCL: per-GPU work is fixed, scaling is weak.

Rank 0 prints ONE JSON line.  `value` is render throughput, Msplats/s = Gaussians
submitted per second of forward render (preprocess + binning + blend), summed
over ranks, measured with device events inside the timed training steps;
`ms_per_step` is the whole train step.  `roofline` prices the dominant
hand-written kernel against HBM with the ALGORITHMIC bytes of one launch;
`cpu_baseline` times the C oracle's forward render of the same workload on the
host cores (rank 0, N=1 only).
"""
from __future__ import annotations

import argparse
import json
import math
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "gaussian-splatting-own_amd"))
sys.path.insert(0, ROOT)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

HBM_PEAK_GBS = 8000.0     # MI355X HBM3E spec peak

WORKLOADS = {
    # name: (P, W, H, sh_degree, skew_fraction)
    "c1": (10_000, 256, 256, 0, 0.0),
    "c2": (100_000, 800, 800, 3, 0.0),
    "c3": (1_000_000, 1920, 1080, 3, 0.0),
    "c5": (6_000_000, 1920, 1080, 3, 0.5),
}


DOMINANT_STAGE = "blend_bwd"      # the longest kernel of the step on every workload here


def algorithmic_bytes(P, N, W, H, M=16):
    """Per-launch algorithmic HBM bytes of each stage (minimum traffic with
    perfect on-chip reuse): B_in = 44 + 12*M bytes of attributes per Gaussian,
    ~80 B of per-Gaussian state out; one 40-byte splat (id 4, xy 8, conic+opacity
    16, rgb 12) per (tile, splat) instance; 20 B per pixel out of the forward
    blend, 32 B per pixel into the backward blend plus 36 B of gradient per
    Gaussian."""
    b_in = 44 + 12 * M
    return {
        "preprocess_fwd": P * (12 * M + 44 + 80),
        "blend_fwd": 40 * N + 20 * W * H,
        "blend_bwd": 40 * N + 32 * W * H + 36 * P,
        "preprocess_bwd": P * (b_in + 80 + 36 + b_in + 12),
    }


def published_binning_bytes(P, V, N, W, H):
    """Bytes the published binning chain moves for the same lists: prefix sum 8 P, key/value
    duplication 12 N + 8 V, 64-bit (tile | depth) radix sort 24 N per 8-bit pass over
    32 + ceil(log2 tiles) key bits, range identification 8 N + 8 tiles.  The binning here
    produces the identical lists without sorting instances, so it moves far fewer bytes;
    this figure only prices its time against that chain."""
    tiles = ((W + 15) // 16) * ((H + 15) // 16)
    key_bits = 32 + max(1, math.ceil(math.log2(tiles)))
    return 8 * P + 12 * N + 8 * V + 24 * N * math.ceil(key_bits / 8) + 8 * N + 8 * tiles


def measured_traffic(workload):
    """HBM bytes per launch from the committed rocprofv3 counter runs (FETCH_SIZE / WRITE_SIZE
    passes, corrected as profiles/r01_traffic.json says), or {} when none exists for this workload."""
    path = os.path.join(ROOT, "profiles", "r01_traffic.json")
    try:
        with open(path) as f:
            t = json.load(f)
        return t["bytes_per_launch"] if t.get("workload") == workload else {}
    except (OSError, ValueError, KeyError):
        return {}


def measured_valu(workload):
    """Vector-ALU counters per launch from the same committed counter runs, or {}."""
    path = os.path.join(ROOT, "profiles", "r01_traffic.json")
    try:
        with open(path) as f:
            t = json.load(f)
        return t.get("valu", {}) if t.get("workload") == workload else {}
    except (OSError, ValueError):
        return {}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", default="c3", choices=sorted(WORKLOADS))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--unfused", action="store_true",
                    help="apply the model's activations with PyTorch ops, as the reference does, instead of in-kernel")
    ap.add_argument("--short-lists", action="store_true",
                    help="bin each Gaussian only into the tiles its alpha >= 1/255 ellipse's bounding box reaches "
                         "(same images and gradients; not the reference's internal lists)")
    ap.add_argument("--dp-grads", action="store_true",
                    help="data-parallel training: all-reduce the gradients of all ranks (one flat buffer, RCCL) "
                         "inside the timed step, instead of the scalar loss only")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    # under torch.distributed.run (RANK set) the process group is always formed, also for one
    # rank, so that the RCCL path is the one exercised whatever N is
    use_dist = world > 1 or ("RANK" in os.environ and "MASTER_PORT" in os.environ)
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    if args.gpus != world and rank == 0:
        print(f"warning: --gpus {args.gpus} but WORLD_SIZE {world}", file=sys.stderr)

    import diff_gaussian_rasterization as dgr
    if args.short_lists:
        dgr.REFERENCE_LISTS = False
    from gsplat_host.cameras import orbit_camera
    from gsplat_host.loss import photometric_loss
    from gsplat_host.model import GaussianParams
    from gsplat_host.render import PipelineParams, render
    from gsplat_host.synthetic import make_cloud

    P, W, H, deg, skew = WORKLOADS[args.workload]
    cloud = make_cloud(P, sh_degree=deg, seed=0, skew_fraction=skew)
    model = GaussianParams(cloud, max_sh_degree=3 if deg == 3 else deg, device=dev)
    cam = orbit_camera(W, H, angle_deg=45.0 * rank, device=dev)
    bg = torch.zeros(3, device=dev)                                   # train.py:41
    gt = torch.rand(3, H, W, generator=torch.Generator().manual_seed(1)).to(dev)
    pipe = PipelineParams(fused_activations=not args.unfused)
    params = [p for p in model.parameters()]

    from gsplat_host.optim import GaussianAdam, all_reduce_gradients
    opt = GaussianAdam.for_model(model)

    # Inside the timed steps only the dominant kernel is bracketed by events (every bracket costs
    # the step a few microseconds of device idle time: all eight stages together ~0.09 ms of a
    # 1.06 ms step); the per-stage table comes from a second, untimed pass.
    timer = dgr.StageTimer(only=(DOMINANT_STAGE,))
    all_stages = dgr.StageTimer()
    fwd_events = []

    def step(timed: bool):
        for p in params:
            p.grad = None
        if timed:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
        out = render(cam, model, pipe, bg)
        if timed:
            e1.record()
            fwd_events.append((e0, e1))
        loss = photometric_loss(out["render"], gt)
        loss.backward()
        if use_dist:
            # RCCL over xGMI: one fp32 scalar.  Nothing in the next step needs the mean loss, so the
            # collective runs on RCCL's own stream beside it; the previous one is waited for first, so
            # at most one is outstanding, and the last one before the closing barrier.
            if step.pending is not None:
                step.pending[0].wait()
            lv = loss.detach().clone()
            step.pending = (dist.all_reduce(lv, async_op=True), lv)
        if args.dp_grads:
            step.grad_scale = all_reduce_gradients(opt.params)        # RCCL over xGMI: 59 floats per Gaussian
        return out
    step.grad_scale = 1.0
    step.pending = None

    def barrier():
        if step.pending is not None:
            step.pending[0].wait()
            step.pending = None
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step(False)
    barrier()
    dgr.set_stage_timer(timer)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = step(True)
    barrier()
    elapsed = time.perf_counter() - t0
    dgr.set_stage_timer(all_stages)
    for _ in range(args.steps):
        step(False)
    barrier()
    dgr.set_stage_timer(None)

    if use_dist:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    render_ms = sum(a.elapsed_time(b) for a, b in fwd_events) / len(fwd_events)
    if use_dist:
        t = torch.tensor([render_ms], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        render_ms = float(t.item())
    ms_per_step = 1e3 * elapsed / args.steps
    msplats = world * P / (render_ms * 1e-3) / 1e6

    # the optimiser step sits outside the reference's timed span (train.py:95 precedes :126-128):
    # timed on its own, after the step loop, on the last step's gradients
    adam_events = []
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        opt.step(grad_scale=step.grad_scale)
        e1.record()
        adam_events.append((e0, e1))
    torch.cuda.synchronize()
    adam_ms = sorted(a.elapsed_time(b) for a, b in adam_events)[len(adam_events) // 2]
    n_params = sum(p.numel() for p in params)

    if rank == 0:
        stage_ms = all_stages.mean_ms()
        stage_ms.update(timer.mean_ms())          # the dominant kernel: as measured inside the timed steps
        N = int((out["radii"] > 0).sum().item())
        st_N = _instances(dgr, cam, model, bg)
        alg = algorithmic_bytes(P, st_N, W, H)
        dominant = max(alg, key=lambda k: stage_ms.get(k, 0.0))
        if dominant != DOMINANT_STAGE:
            print(f"warning: {dominant} outlasts {DOMINANT_STAGE}; its time comes from the untimed pass", file=sys.stderr)
        achieved = alg[dominant] / (stage_ms[dominant] * 1e-3) / 1e9
        kernels = {k: {"ms": round(stage_ms[k], 4), "algorithmic_GB": round(alg[k] / 1e9, 4) if k in alg else None,
                       "GBps": round(alg[k] / (stage_ms[k] * 1e-3) / 1e9, 1) if k in alg else None}
                   for k in stage_ms}
        bin_ms = sum(stage_ms.get(k, 0.0) for k in ("bin_totals", "bin_sort", "bin_lists"))
        pub = published_binning_bytes(P, N, st_N, W, H)
        kernels["binning_total"] = {"ms": round(bin_ms, 4), "published_chain_GB": round(pub / 1e9, 4),
                                    "GBps_vs_published_chain": round(pub / (bin_ms * 1e-3) / 1e9, 1)}
        traffic = measured_traffic(args.workload).get(dominant)
        valu = measured_valu(args.workload).get(dominant)
        valu_note = None
        if valu:
            # SQ_ACTIVE_INST_VALU counts quad-cycles over all 1024 SIMDs (256 CUs x 4)
            busy = 4.0 * valu["active_quad_cycles"] / 1024.0 / (stage_ms[dominant] * 1e-3 * 2.4e9)
            valu_note = {"wave_instructions": valu["wave_instructions"], "busy_frac_at_2.4GHz": round(busy, 3),
                         "source": "profiles/r01_traffic.json (rocprofv3 SQ_INSTS_VALU, SQ_ACTIVE_INST_VALU)"}
        line = {
            "metric": "train-step ms + render Msplats/s @1080p, 1M Gaussians, 1/2/4/8 MI355X",
            "value": round(msplats, 2), "unit": "Msplats/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(ms_per_step, 3), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"{args.workload}: {P} Gaussians, {W}x{H}, SH degree {deg}, full train step "
                                   "(activations + render + 0.8*L1+0.2*(1-SSIM) + backward), one view per GPU",
                       "gaussians": P, "visible": N, "tile_instances": st_N, "views_per_step": world,
                       "activations": "torch ops (as the reference)" if args.unfused else "in the preprocess kernels",
                       "tile_lists": "short (ellipse bounding boxes)" if args.short_lists else "the reference's"},
            "train_step_ms": round(ms_per_step, 3), "render_ms": round(render_ms, 3),
            "render_Ginstances_per_s": round(world * st_N / (render_ms * 1e-3) / 1e9, 2),
            "roofline": {"bound": "hbm", "kernel": dominant, "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
                         "traffic_unit": "bytes/launch (rocprofv3 FETCH_SIZE x2 + WRITE_SIZE, profiles/r01_traffic.json)",
                         "algorithmic_bytes": alg[dominant], "valu": valu_note,
                         "limited_by": "vector-instruction issue (VALU), not HBM" if valu_note and
                                       valu_note["busy_frac_at_2.4GHz"] > 0.8 else None},
            "kernels": kernels,
            "optimizer": {"what": "fused Adam over the six parameter tensors, outside the timed step as in the reference",
                          "ms": round(adam_ms, 4), "parameters": n_params,
                          "GBps": round(28 * n_params / (adam_ms * 1e-3) / 1e9, 1)},
            "gradient_exchange": "all-reduce of all gradients inside the timed step" if args.dp_grads
                                 else "none (scalar loss only)",
        }
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(cloud, W, H, deg)
        print(json.dumps(line), flush=True)
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


def _instances(dgr, cam, model, bg):
    """Number of (tile, splat) instances of the benchmarked view (untimed)."""
    with torch.no_grad():
        s = dgr.GaussianRasterizationSettings(
            image_height=cam.image_height, image_width=cam.image_width, tanfovx=math.tan(cam.FoVx * 0.5),
            tanfovy=math.tan(cam.FoVy * 0.5), bg=bg, scale_modifier=1.0, viewmatrix=cam.world_view_transform,
            projmatrix=cam.full_proj_transform, sh_degree=model.active_sh_degree, campos=cam.camera_center,
            prefiltered=False, debug=False)
        st = dgr.rasterize_forward(s, model.get_xyz, model.get_opacity, shs=model.get_features,
                                   scales=model.get_scaling, rotations=model.get_rotation)
        n = st.num_rendered
        del st
        return n


def cpu_baseline(cloud, W, H, deg):
    """Forward render of the same workload by the C oracle (a port: the
    reference has no CPU rasterizer), on the host cores, timed once."""
    import numpy as np
    import oracle
    from gsplat_host.cameras import orbit_camera
    threads = int(os.environ.get("OMP_NUM_THREADS", "0")) or min(os.cpu_count() or 1, 16)
    os.environ["OMP_NUM_THREADS"] = str(threads)
    oracle.build()
    o = oracle.Oracle(np.float32)
    cam = orbit_camera(W, H)
    s = oracle.Settings(H, W, math.tan(cam.FoVx / 2), math.tan(cam.FoVy / 2), np.zeros(3, np.float32), 1.0,
                        cam.world_view_transform.numpy(), cam.full_proj_transform.numpy(), deg,
                        cam.camera_center.numpy())
    arrs = dict(means3D=cloud.xyz.numpy(), opacities=cloud.opacity.numpy(), scales=cloud.scaling.numpy(),
                rotations=cloud.rotation.numpy(), shs=cloud.features.numpy())
    o.forward(s, want_margin=False, **arrs)                 # warm-up: page faults, thread pool
    reps, t0 = 0, time.perf_counter()
    while reps < 3 or time.perf_counter() - t0 < 10.0:
        o.forward(s, want_margin=False, **arrs)
        reps += 1
    dt = (time.perf_counter() - t0) / reps
    return {"value": round(cloud.count / dt / 1e6, 4), "unit": "Msplats/s", "cores": threads, "kind": "port",
            "sample": f"{reps} forward renders (preprocess + binning + blend) of the full workload, {dt:.2f} s each; "
                      "blend and preprocess run on all cores (OpenMP), the sort on one"}


if __name__ == "__main__":
    main()
