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
same replicated cloud (camera rotated by rank*45 degrees); the path has no exchange
between the ranks, the mean loss of the timed steps is all-reduced over RCCL once,
before the closing barrier: per-GPU work is fixed, scaling is weak.

A measurement starts with 40 untimed steps (the W warm-up steps are the last of
them): a fresh process reaches its steady step time only after some thirty (GPU
clocks, allocator), and K timed steps right behind five would measure that ramp.

Rank 0 prints ONE JSON line.  `value` is render throughput, Msplats/s = Gaussians
submitted per second of forward render (preprocess + binning + blend), summed
over ranks, measured with device events inside the timed training steps;
`ms_per_step` is the whole train step.  `roofline` prices the dominant
hand-written kernel against HBM with the ALGORITHMIC bytes of one launch over
the kernel's own duration (events stamped by the launch itself, not brackets on
the stream): that is `achieved` / `frac`, an upper bound on what the kernel moves.
Beside it the line says, from committed counter runs of the same workload, what
the kernel measurably does: `frac_measured_bytes` (HBM bytes counted by the
hardware over the same time), the busiest unit and its fraction, and `bound` =
that unit when it is above 80 % of its peak, "latency" otherwise.  `lists`
gives the measured tile-list statistics (instances, longest and mean list, how
deep the lists are walked, pixel-splat pairs).  `cpu_baseline` times the C
oracle's forward render of the same workload on the host cores (rank 0, N=1
only).  On the default workload at one GPU the line also carries the same step
through the reference's unchanged operator API (`reference_operator_api`:
PyTorch activations feeding GaussianRasterizer, what a user of the drop-in gets
without changing a line), and as `variants` the no-early-termination scene
(every opacity 0.1, the reference's initial value, scene/gaussian_model.py:139)
and the skewed 6M cloud with the same opacities (c5w) and with every opacity
0.01 (c5r: the state after the reference's reset_opacity).
"""
from __future__ import annotations

import argparse
import glob
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
CLOCK_HZ = 2.4e9          # peak shader clock; the chip may hold less under load, so the busy fractions below are lower bounds
SIMDS, CUS = 1024, 256

WORKLOADS = {
    # name: (P, W, H, sh_degree, skew_fraction, opacity_const)
    "c1": (10_000, 256, 256, 0, 0.0, None),
    "c2": (100_000, 800, 800, 3, 0.0, None),
    "c3": (1_000_000, 1920, 1080, 3, 0.0, None),
    # the C3 cloud with every opacity 0.1: no pixel saturates early, every tile list is walked far deeper
    "c3w": (1_000_000, 1920, 1080, 3, 0.0, 0.1),
    "c5": (6_000_000, 1920, 1080, 3, 0.5, None),
    # the worst case for tile-load skew: the C5 cloud (half of the Gaussians in a hot spot) with every opacity 0.1
    "c5w": (6_000_000, 1920, 1080, 3, 0.5, 0.1),
    # the same cloud as the reference's reset_opacity leaves it (scene/gaussian_model.py:210-213: every opacity at most
    # 0.01): the hot spot's tiles are walked tens of thousands of entries deep, the case the depth segments are for
    "c5r": (6_000_000, 1920, 1080, 3, 0.5, 0.01),
}

SETTLE_STEPS = 40                 # untimed steps a measurement starts with, the W warm-up steps included
DOMINANT_STAGE = "blend_bwd"      # the longest kernel of the step on the headline workload: what the roofline object prices


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


def committed_counters(workload):
    """(file name, contents) of the newest committed counter summary for this workload
    (profiles/rNN_traffic*.json, written by tools/make_traffic_json.py from rocprofv3 --pmc runs), or (None, {})."""
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_traffic*.json")), reverse=True):
        try:
            with open(path) as f:
                t = json.load(f)
        except (OSError, ValueError):
            continue
        if t.get("workload") == workload:
            return os.path.relpath(path, ROOT), t
    return None, {}


def limits(kernel, ms, alg_bytes, counters, source):
    """How close one kernel runs to each of its three candidate limits, from its duration and the committed
    counters of the same workload.  hbm: measured HBM bytes over the 8 TB/s peak.  valu: vector
    wave-instructions x 2 issue cycles (a wave64 instruction occupies a SIMD-32 for two cycles) over the
    cycles of the 1024 SIMDs.  lds: cycles the LDS arrays were busy (SQ_LDS_IDX_ACTIVE, bank conflicts
    included) over the cycles of the 256 CUs' LDS units.  The last two assume the 2.4 GHz peak clock, so they
    are lower bounds.  A kernel none of whose fractions is near 1 is bound by latency and instruction
    dependencies at the occupancy it runs at, not by any one of these."""
    t = ms * 1e-3
    out = {"algorithmic_bytes": alg_bytes, "source": source}
    traffic = counters.get("bytes_per_launch", {}).get(kernel)
    fr = {"hbm_algorithmic": alg_bytes / t / (HBM_PEAK_GBS * 1e9)}
    if traffic:
        out["measured_hbm_bytes"] = traffic
        fr["hbm_measured"] = traffic / t / (HBM_PEAK_GBS * 1e9)
    valu = counters.get("valu", {}).get(kernel)
    if valu:
        out["valu_wave_instructions"] = valu["wave_instructions"]
        fr["valu_issue"] = 2.0 * valu["wave_instructions"] / (SIMDS * t * CLOCK_HZ)
    lds = counters.get("lds", {}).get(kernel)
    if lds:
        out["lds_wave_instructions"] = lds.get("wave_instructions")
        out["lds_bank_conflict_cycles"] = lds.get("bank_conflict_cycles")
        if lds.get("idx_active_cycles"):
            fr["lds_array"] = lds["idx_active_cycles"] / (CUS * t * CLOCK_HZ)
    out["fractions_of_peak"] = {k: round(v, 3) for k, v in fr.items()}
    real = {k: v for k, v in fr.items() if k != "hbm_algorithmic"}
    if real:
        top = max(real, key=real.get)
        out["busiest_unit"], out["busiest_unit_frac"] = top, round(real[top], 3)
        out["bound"] = top if real[top] > 0.8 else "latency"
        out["limited_by"] = {"hbm_measured": "HBM bandwidth", "valu_issue": "vector-instruction issue",
                             "lds_array": "the LDS"}[top] + f" is the busiest unit ({real[top]:.0%} of its peak; " + \
            ", ".join(f"{k} {v:.0%}" for k, v in real.items() if k != top) + ")" + \
            ("" if real[top] > 0.8 else "; none is saturated: latency and instruction dependencies at 8 waves per SIMD")
    return out


class Scene:
    def __init__(self, workload, dev, rank):
        from gsplat_host.cameras import orbit_camera
        from gsplat_host.model import GaussianParams
        from gsplat_host.synthetic import make_cloud
        self.name = workload
        self.P, self.W, self.H, self.deg, skew, opacity = WORKLOADS[workload]
        self.cloud = make_cloud(self.P, sh_degree=self.deg, seed=0, skew_fraction=skew, opacity_const=opacity)
        self.model = GaussianParams(self.cloud, max_sh_degree=3 if self.deg == 3 else self.deg, device=dev)
        self.cam = orbit_camera(self.W, self.H, angle_deg=45.0 * rank, device=dev)
        self.bg = torch.zeros(3, device=dev)                                   # train.py:41
        self.gt = torch.rand(3, self.H, self.W, generator=torch.Generator().manual_seed(1)).to(dev)
        self.params = [p for p in self.model.parameters()]


def measure(scene, dgr, fused, steps, warmup, use_dist=False, dp_grads=False, opt=None, stage_table=True,
            events_in_timed_steps=False):
    """`steps` timed train steps of one scene: wall time over all of them (between barriers) and the forward
    render by device events inside them; then, untimed, the same steps again with the two blend kernels
    stamped by their launches' own events (inside the timed steps instead with `events_in_timed_steps`:
    such launches cost the step ~0.04 ms each), and once more with every stage bracketed for the per-stage table."""
    from gsplat_host.loss import photometric_loss
    from gsplat_host.optim import all_reduce_gradients
    from gsplat_host.render import PipelineParams, render
    pipe = PipelineParams(fused_activations=fused)
    timer = dgr.StageTimer(only=(), exact=dgr.StageTimer.EXACT_STAGES)
    all_stages = dgr.StageTimer()
    fwd_events = []
    state = {"losses": [], "mean_loss": None, "grad_scale": 1.0, "out": None}

    def step(timed):
        for p in scene.params:
            p.grad = None
        if timed:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
        out = render(scene.cam, scene.model, pipe, scene.bg)
        if timed:
            e1.record()
            fwd_events.append((e0, e1))
        loss = photometric_loss(out["render"], scene.gt)
        loss.backward()
        if use_dist:
            # The render path has no exchange step between the ranks (one view each).  The mean loss over all views
            # is for the log only: the steps keep their loss tensors and ONE collective (RCCL over xGMI, one fp32
            # scalar) reduces their mean before the closing barrier.  A collective per step would cost every step
            # a copy kernel, two stream markers and a kernel that waits for the slowest rank.
            state["losses"].append(loss.detach())
        if dp_grads:
            state["grad_scale"] = all_reduce_gradients(opt.params)        # RCCL over xGMI: 59 floats per Gaussian
        state["out"] = out

    def barrier():
        # every rank waits for its own GPU, then for all others: the collective that reduces the mean loss of the steps
        # since the last barrier is the point every rank has to arrive at (a dist.barrier() behind it would be a second
        # collective of the same kind inside the timed region)
        torch.cuda.synchronize()
        if use_dist:
            mean_loss = torch.stack(state["losses"]).mean() if state["losses"] else torch.zeros((), device=scene.bg.device)
            state["losses"].clear()
            dist.all_reduce(mean_loss)
            state["mean_loss"] = float(mean_loss.item()) / dist.get_world_size()
        torch.cuda.synchronize()

    # A fresh process takes some forty steps to reach its steady step time (the GPU's clocks ramp up over tens of
    # milliseconds, the caching allocator settles): those come first, untimed like the W warm-up steps behind them
    for _ in range(max(0, SETTLE_STEPS - warmup)):
        step(False)
    for _ in range(warmup):
        step(False)
    barrier()
    if events_in_timed_steps:
        dgr.set_stage_timer(timer)
    t0 = time.perf_counter()
    for _ in range(steps):
        step(True)
    barrier()
    elapsed = time.perf_counter() - t0
    if not events_in_timed_steps:
        dgr.set_stage_timer(timer)
        for _ in range(steps):
            step(False)
        barrier()
    dgr.set_stage_timer(None)
    exact = timer.exact_ms()
    timer.close()
    stage_ms = {}
    if stage_table:
        dgr.set_stage_timer(all_stages)
        for _ in range(steps):
            step(False)
        barrier()
        dgr.set_stage_timer(None)
        stage_ms = all_stages.mean_ms()
    render_ms = sum(a.elapsed_time(b) for a, b in fwd_events) / len(fwd_events)
    radii = state["out"]["radii"]
    return {"elapsed": elapsed, "render_ms": render_ms, "exact": exact, "stage_ms": stage_ms,
            "visible": int((radii > 0).sum().item()), "grad_scale": state["grad_scale"], "mean_loss": state["mean_loss"]}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--workload", default="c3", choices=sorted(WORKLOADS))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--events-in-timed-steps", action="store_true",
                    help="stamp the blend kernels with their launches' own events inside the timed steps instead of in "
                         "a pass of the same steps right after them (such launches slow the step down a little)")
    ap.add_argument("--no-variants", action="store_true",
                    help="skip the extra measurements of the default line (reference operator API, opacity-0.1 scene)")
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
        # the communicator is set up by its first collective, which takes seconds, and PyTorch loads a kernel's code
        # at its first use (tens of milliseconds): here, not in the barrier between the warm-up and the timed steps,
        # where the GPU would sit idle for that long and start the timed steps at idle clocks
        z = torch.zeros((), device=dev)
        dist.all_reduce(torch.stack([z, z]).mean())     # also loads the kernels the barrier's reduction of the losses uses
        torch.cuda.synchronize()
    if args.gpus != world and rank == 0:
        print(f"warning: --gpus {args.gpus} but WORLD_SIZE {world}", file=sys.stderr)

    import diff_gaussian_rasterization as dgr
    # one process per GPU and nothing else on it: the depth sort may give workgroup b chunk b instead of a ticket
    dgr.assume_exclusive_device(True)
    if args.short_lists:
        dgr.REFERENCE_LISTS = False
    from gsplat_host.optim import GaussianAdam

    scene = Scene(args.workload, dev, rank)
    P, W, H, deg = scene.P, scene.W, scene.H, scene.deg
    opt = GaussianAdam.for_model(scene.model)
    m = measure(scene, dgr, not args.unfused, args.steps, args.warmup, use_dist, args.dp_grads, opt,
                events_in_timed_steps=args.events_in_timed_steps)

    elapsed, render_ms = m["elapsed"], m["render_ms"]
    peak_mb = torch.cuda.max_memory_allocated(dev) / 2 ** 20
    if use_dist:
        t = torch.tensor([elapsed, render_ms], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed, render_ms = float(t[0].item()), float(t[1].item())
    ms_per_step = 1e3 * elapsed / args.steps
    msplats = world * P / (render_ms * 1e-3) / 1e6

    # the tile lists of the scene the steps were timed on (the optimiser steps below move it)
    lists = _list_stats(dgr, scene) if rank == 0 else None

    # the optimiser step sits outside the reference's timed span (train.py:95 precedes :126-128):
    # timed on its own, after the step loop, on the last step's gradients
    adam_events = []
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        opt.step(grad_scale=m["grad_scale"])
        e1.record()
        adam_events.append((e0, e1))
    torch.cuda.synchronize()
    adam_ms = sorted(a.elapsed_time(b) for a, b in adam_events)[len(adam_events) // 2]
    n_params = sum(p.numel() for p in scene.params)

    if rank == 0:
        stage_ms = dict(m["stage_ms"])
        for k, (mean, lo, hi, n) in m["exact"].items():
            stage_ms[k] = mean                     # the blend kernels: the kernel's own duration inside the timed steps
        N = m["visible"]
        st_N = lists["instances"]
        alg = algorithmic_bytes(P, st_N, W, H)
        # The roofline object is about the blend backward, the longest kernel of the headline workload.  On a
        # workload where another stage outlasts it (6M Gaussians: the per-Gaussian backward) that is said on
        # stderr; that kernel skips the Gaussians no pixel blended, so its algorithmic bytes overstate its traffic.
        longest = max(alg, key=lambda k: stage_ms.get(k, 0.0))
        if longest != DOMINANT_STAGE:
            print(f"note: {longest} ({stage_ms[longest]:.3f} ms) outlasts {DOMINANT_STAGE} on this workload", file=sys.stderr)
        dominant = DOMINANT_STAGE
        dom_ms = stage_ms[dominant]
        achieved = alg[dominant] / (dom_ms * 1e-3) / 1e9
        kernels = {k: {"ms": round(stage_ms[k], 4), "algorithmic_GB": round(alg[k] / 1e9, 4) if k in alg else None,
                       "GBps": round(alg[k] / (stage_ms[k] * 1e-3) / 1e9, 1) if k in alg else None}
                   for k in stage_ms}
        for k, (mean, lo, hi, n) in m["exact"].items():
            kernels[k].update({"timing": "kernel start to end, stamped by the launch", "min_ms": round(lo, 4),
                               "max_ms": round(hi, 4), "launches": n})
        bin_ms = sum(stage_ms.get(k, 0.0) for k in ("bin_sort", "bin_lists"))
        pub = published_binning_bytes(P, N, st_N, W, H)
        if bin_ms > 0:
            kernels["binning_total"] = {"ms": round(bin_ms, 4), "published_chain_GB": round(pub / 1e9, 4),
                                        "GBps_vs_published_chain": round(pub / (bin_ms * 1e-3) / 1e9, 1)}
        source, counters = committed_counters(args.workload)
        lim = limits(dominant, dom_ms, alg[dominant], counters, source)
        line = {
            "metric": "train-step ms + render Msplats/s @1080p, 1M Gaussians, 1/2/4/8 MI355X",
            "value": round(msplats, 2), "unit": "Msplats/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "untimed_steps_before_the_timed_ones": max(SETTLE_STEPS, args.warmup),
            "ms_per_step": round(ms_per_step, 3), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"{args.workload}: {P} Gaussians, {W}x{H}, SH degree {deg}, full train step "
                                   "(activations + render + 0.8*L1+0.2*(1-SSIM) + backward), one view per GPU",
                       "gaussians": P, "visible": N, "tile_instances": st_N, "views_per_step": world,
                       "activations": "torch ops (as the reference)" if args.unfused else "in the preprocess kernels",
                       "tile_lists": "short (ellipse bounding boxes)" if args.short_lists else "the reference's",
                       "exclusive_device": True},
            "lists": lists,
            "train_step_ms": round(ms_per_step, 3), "render_ms": round(render_ms, 3),
            "render_Ginstances_per_s": round(world * st_N / (render_ms * 1e-3) / 1e9, 2),
            "roofline": {"bound": lim.pop("bound", "unmeasured (no committed counters for this workload)"),
                         "priced_against": "hbm: achieved = algorithmic bytes of one launch / kernel time; an upper bound on "
                                           "what the kernel moves, not a utilisation (see frac_measured_bytes)",
                         "kernel": dominant, "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4),
                         "frac_measured_bytes": lim.get("fractions_of_peak", {}).get("hbm_measured"),
                         "traffic": lim.get("measured_hbm_bytes"),
                         "traffic_unit": "bytes/launch (rocprofv3 FETCH_SIZE x2 + WRITE_SIZE)",
                         "kernel_ms": round(dom_ms, 4),
                         "kernel_also_clears_bytes": 248 * P,
                         "frac_with_the_cleared_bytes": round((alg[dominant] + 248 * P) / (dom_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                         "kernel_also_clears": "the gradient tensors of preprocess_bwd (62 floats per Gaussian), a share "
                                               "per workgroup after its own work (gsr_blend_bwd_fill); not counted in "
                                               "the algorithmic bytes, included in kernel_ms",
                         "kernel_ms_from": "HIP events stamped by the kernel's launch (its own start and end), mean over "
                                           + (f"the {args.steps} timed steps" if args.events_in_timed_steps else
                                              f"{args.steps} steps run right after the timed ones (inside them such "
                                              "launches would slow the step down)"),
                         **lim},
            "kernels": kernels,
            "optimizer": {"what": "fused Adam over the six parameter tensors, outside the timed step as in the reference "
                                  "(train.py:67-95 brackets render, loss and backward; optimizer.step() follows at :126-128)",
                          "ms": round(adam_ms, 4), "parameters": n_params,
                          "GBps": round(28 * n_params / (adam_ms * 1e-3) / 1e9, 1),
                          "train_step_plus_optimizer_ms": round(ms_per_step + adam_ms, 3)},
            "gradient_exchange": "all-reduce of all gradients inside the timed step" if args.dp_grads
                                 else "none (the mean loss, one scalar, once per timed region)",
            "communicator": {"backend": dist.get_backend(), "ranks": dist.get_world_size()} if use_dist else None,
            "peak_device_memory_MB": round(peak_mb, 1),
            "mean_loss_over_ranks": m["mean_loss"],
        }
        if world == 1 and args.workload == "c3" and not args.no_variants and not args.unfused:
            line["reference_operator_api"], line["variants"] = variants(scene, dgr, args, dev)
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(scene.cloud, W, H, deg)
        print(json.dumps(line), flush=True)
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


def variants(scene, dgr, args, dev):
    """The default workload's step measured three more ways, shorter runs of the same loop: through the
    reference's unchanged operator API (its model getters as PyTorch ops feeding GaussianRasterizer,
    gaussian_renderer/__init__.py:53-93), returned first; then, as a dict, on the same cloud with every opacity 0.1
    (the reference's initial opacity: nothing saturates early, the hard case for the blend kernels) and on the
    skewed 6M cloud with those opacities (the worst case for tile-load skew) and with every opacity 0.01 (the worst
    case for the depth of single tiles' walks)."""
    steps, warmup = max(5, args.steps // 2), 3
    m = measure(scene, dgr, False, steps, warmup, stage_table=False)
    api = {
        "what": "activations as PyTorch ops (exp, sigmoid, normalize, cat) feeding GaussianRasterizer, as the reference "
                "calls it (gaussian_renderer/__init__.py:53-93): the step a user of the drop-in gets without changing a line",
        "value": round(scene.P / (m["render_ms"] * 1e-3) / 1e6, 1), "unit": "Msplats/s",
        "render_ms": round(m["render_ms"], 3), "train_step_ms": round(1e3 * m["elapsed"] / steps, 3), "steps": steps}
    out = {}
    for name, key, what in (
            ("c3w", "c3w_opacity_0.1", "the same cloud with every opacity 0.1 (scene/gaussian_model.py:139): no early termination"),
            ("c5w", "c5w_skew_opacity_0.1", "6M Gaussians, half of them in a hot spot, every opacity 0.1: tile-load skew "
                                            "without early termination"),
            ("c5r", "c5r_skew_opacity_0.01", "the same 6M cloud with every opacity 0.01, as after the reference's reset_opacity "
                                             "(scene/gaussian_model.py:210-213): single tiles are walked many times deeper than "
                                             "the rest, the backward cuts those walks into depth segments")):
        hard = Scene(name, dev, 0)
        m = measure(hard, dgr, True, steps, warmup, stage_table=True)
        lists = _list_stats(dgr, hard)
        n_inst = lists["instances"]
        alg = algorithmic_bytes(hard.P, n_inst, hard.W, hard.H)
        bwd_ms, fwd_ms = m["exact"]["blend_bwd"][0], m["exact"]["blend_fwd"][0]
        source, counters = committed_counters(name)
        out[key] = {
            "what": what,
            "render_ms": round(m["render_ms"], 3), "train_step_ms": round(1e3 * m["elapsed"] / steps, 3),
            "render_Msplats_per_s": round(hard.P / (m["render_ms"] * 1e-3) / 1e6, 1), "steps": steps,
            "lists": lists, "stage_ms": {k: round(v, 4) for k, v in m["stage_ms"].items()},
            "blend_fwd": {"kernel_ms": round(fwd_ms, 4), "algorithmic_GBps": round(alg["blend_fwd"] / fwd_ms / 1e6, 1),
                          "frac_of_hbm_peak": round(alg["blend_fwd"] / fwd_ms / 1e6 / HBM_PEAK_GBS, 4)},
            "blend_bwd": {"kernel_ms": round(bwd_ms, 4), "algorithmic_GBps": round(alg["blend_bwd"] / bwd_ms / 1e6, 1),
                          "frac_of_hbm_peak": round(alg["blend_bwd"] / bwd_ms / 1e6 / HBM_PEAK_GBS, 4),
                          **limits("blend_bwd", bwd_ms, alg["blend_bwd"], counters, source)}}
        del hard
        torch.cuda.empty_cache()
    return api, out


def _list_stats(dgr, scene):
    """Measured statistics of the benchmarked view's tile lists, from one untimed forward: the number of
    (tile, splat) instances, the longest and the mean list, how deep the lists are walked (per tile, the position of
    the deepest splat any of its pixels blended: where its forward stops and its backward starts), and pixel-splat
    pairs: `listed` = 256 per instance (what a kernel without culling or early termination would evaluate), and
    `to_last_contributor` = the sum over pixels of the position of the pixel's last contributor (the entries a
    pixel's backward walks, and what its forward walks at least)."""
    cam, model = scene.cam, scene.model
    with torch.no_grad():
        s = dgr.GaussianRasterizationSettings(
            image_height=cam.image_height, image_width=cam.image_width, tanfovx=math.tan(cam.FoVx * 0.5),
            tanfovy=math.tan(cam.FoVy * 0.5), bg=scene.bg, scale_modifier=1.0, viewmatrix=cam.world_view_transform,
            projmatrix=cam.full_proj_transform, sh_degree=model.active_sh_degree, campos=cam.camera_center,
            prefiltered=False, debug=False)
        st = dgr.rasterize_forward(s, model.get_xyz, model.get_opacity, shs=model.get_features,
                                   scales=model.get_scaling, rotations=model.get_rotation)
        r = st.ranges.to(torch.int64)
        length = r[:, 1] - r[:, 0]
        H, W = st.H, st.W
        gy, gx = (H + 15) // 16, (W + 15) // 16
        nc = torch.zeros((gy * 16, gx * 16), dtype=torch.int64, device=length.device)
        nc[:H, :W] = st.n_contrib.to(torch.int64)
        depth = nc.reshape(gy, 16, gx, 16).amax(dim=(1, 3)).reshape(-1)
        out = {"instances": int(st.num_rendered), "tiles": int(length.numel()),
               "tile_list_length": {"max": int(length.max()), "mean": round(float(length.float().mean()), 1)},
               "walked_depth_per_tile": {"max": int(depth.max()), "mean": round(float(depth.float().mean()), 1),
                                         "share_of_walk_in_the_deepest_1pct_of_tiles":
                                             round(float(depth.sort().values[-max(1, depth.numel() // 100):].sum() / depth.sum().clamp_min(1)), 4)},
               "pixel_splat_pairs": {"listed": 256 * int(st.num_rendered), "to_last_contributor": int(nc.sum())}}
        asked, seg_len, per_tile = dgr.segment_counts(st)
        if seg_len:
            out["depth_segments"] = {"entries_per_segment": seg_len, "boundaries_recorded": asked,
                                     "most_in_one_tile": int(per_tile.max())}
        del st
        return out


def cpu_baseline(cloud, W, H, deg):
    """Forward render of the same workload by the C oracle, on the host cores.  A PORT: the reference has
    no CPU rasterizer (its only implementation is the CUDA one), so this is the build's own C / OpenMP
    restatement of the algorithm, not PyTorch-CPU; its depth sort runs on one core."""
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
            "sample": f"{reps} forward renders (preprocess + binning + blend) of the full workload, {dt:.2f} s each, by "
                      "the C/OpenMP oracle (not PyTorch-CPU: the reference has no CPU rasterizer); blend and "
                      "preprocess run on all cores, the sort on one"}


if __name__ == "__main__":
    main()
