"""Optimiser step and data-parallel gradient exchange of the reference's training loop.

The reference steps ``torch.optim.Adam`` over six parameter tensors with one
learning rate each and ``eps = 1e-15`` (``scene/gaussian_model.py:154-163``,
``train.py:126-128``) and decays the position learning rate log-linearly
(``:164-175``, ``utils/general_utils.py:29-62``).  Here the update of all six
tensors is ONE fused HIP launch (``gsr_adam_step``), and, when several ranks
each render their own view of a replicated model, their gradients meet in one
flat buffer that is summed over the ranks (reduce-scatter + all-gather over RCCL) before the step.  There is no
CPU path for the update: the parameters must live on the GPU.
"""
from __future__ import annotations

import ctypes as C
import math
from dataclasses import dataclass
from typing import Dict, List, Optional

import torch
import torch.distributed as dist

import diff_gaussian_rasterization as _dgr

_lib = _dgr._lib
MAX_TENSORS = 8


class _AdamTensor(C.Structure):
    _fields_ = [("param", C.c_void_p), ("grad", C.c_void_p), ("exp_avg", C.c_void_p), ("exp_avg_sq", C.c_void_p),
                ("n", C.c_size_t), ("lr", C.c_float), ("vec4", C.c_int)]


_lib.gsr_adam_step.argtypes = [C.c_int, C.POINTER(_AdamTensor), C.c_double, C.c_double, C.c_double, C.c_int, C.c_float,
                               C.c_void_p]


def expon_lr(step: int, lr_init: float, lr_final: float, lr_delay_steps: int = 0, lr_delay_mult: float = 1.0,
             max_steps: int = 1_000_000) -> float:
    """Log-linear learning-rate decay with an optional eased-in start (utils/general_utils.py:29-62)."""
    if step < 0 or (lr_init == 0.0 and lr_final == 0.0):
        return 0.0
    if lr_delay_steps > 0:
        delay = lr_delay_mult + (1 - lr_delay_mult) * math.sin(0.5 * math.pi * min(max(step / lr_delay_steps, 0), 1))
    else:
        delay = 1.0
    t = min(max(step / max_steps, 0), 1)
    return delay * math.exp(math.log(lr_init) * (1 - t) + math.log(lr_final) * t)


@dataclass
class OptimizationParams:
    """The optimiser-related fields of arguments/__init__.py:71-90."""
    position_lr_init: float = 0.00016
    position_lr_final: float = 0.0000016
    position_lr_delay_mult: float = 0.01
    position_lr_max_steps: int = 30_000
    feature_lr: float = 0.0025
    opacity_lr: float = 0.05
    scaling_lr: float = 0.005
    rotation_lr: float = 0.001


EXCHANGES = ("scatter_gather", "direct", "all_reduce")


def _sum_flat_over_ranks(flat: torch.Tensor, group, method: str) -> None:
    """In place: ``flat`` (1-D fp32, the same length on every rank) becomes its sum over the ranks of ``group``.

    ``scatter_gather`` (the default): reduce-scatter, then all-gather.  Rank r ends the first step with the sum of
    slice r of every rank's buffer and the second hands the slices round: each rank sends and receives 2 (n-1)/n
    of the buffer, spread over its links to all n-1 peers, where one ring would push the same bytes through one
    link (8 GPUs joined point to point by 7 xGMI links each; 236 MB of gradients at a million Gaussians).
    ``direct``: the same two steps spelled out with point-to-point traffic only, an all-to-all of the slices,
    a local sum, an all-gather: no ring even if the library would build one for a reduce-scatter.
    ``all_reduce``: one collective, the library's choice of algorithm.
    The first two need the length to be a multiple of the number of ranks (a GradArena pads for that); otherwise
    the call falls back to ``all_reduce``.  ``gloo`` only ever sees host tensors."""
    if method not in EXCHANGES:
        raise ValueError(f"exchange method must be one of {EXCHANGES}")
    world = dist.get_world_size(group)
    if world <= 1:
        return
    work = flat
    on_host = dist.get_backend(group) == "gloo" and flat.is_cuda
    if on_host:
        work = flat.cpu()
    n = work.numel()
    if method == "all_reduce" or n % world != 0:
        dist.all_reduce(work, op=dist.ReduceOp.SUM, group=group)
    else:
        piece = n // world
        if method == "scatter_gather":
            mine = torch.empty((piece,), dtype=work.dtype, device=work.device)
            dist.reduce_scatter_tensor(mine, work, op=dist.ReduceOp.SUM, group=group)
        else:
            theirs = torch.empty_like(work)                         # slice `rank` of every rank's buffer
            dist.all_to_all_single(theirs, work, group=group)
            mine = theirs.view(world, piece).sum(dim=0)
        dist.all_gather_into_tensor(work, mine, group=group)
    if on_host:
        flat.copy_(work)


class GradArena:
    """One flat fp32 buffer holding the gradients of all parameter tensors back to back (each
    start 16-byte aligned, the whole padded to a multiple of 1024 floats so that it splits evenly over
    any power-of-two number of ranks), with a view per tensor.  Backward passes write into the views, the
    data-parallel exchange works on ``flat`` as a whole, and the optimiser reads the views."""

    def __init__(self, params: Dict[str, torch.Tensor]):
        self.offsets, total = {}, 0
        for name, p in params.items():
            self.offsets[name] = total
            total += (p.numel() + 3) // 4 * 4
        first = next(iter(params.values()))
        self.flat = torch.zeros(((total + 1023) // 1024 * 1024,), dtype=torch.float32, device=first.device)
        self.views = {name: self.flat[o:o + params[name].numel()].view_as(params[name])
                      for name, o in self.offsets.items()}

    def attach(self, params: Dict[str, torch.Tensor]) -> None:
        """Makes every parameter's ``.grad`` the arena view, so autograd accumulates in place."""
        for name, p in params.items():
            p.grad = self.views[name]

    def zero_(self) -> None:
        self.flat.zero_()

    def all_reduce(self, group=None, average: bool = True, method: str = "scatter_gather") -> float:
        """Sums the gradients over the ranks of ``group`` (``method``: see _sum_flat_over_ranks; two collectives on
        the one flat buffer by default) and returns the factor the
        optimiser step still has to apply: 1 / world size when every rank back-propagated its own
        un-normalised loss (``average``, the default: sum -> mean over ranks), 1.0 when the losses were
        already scaled so that their sum over ranks IS the wanted gradient (``average=False``: what
        ``multiview_step`` does, it divides every view's loss by the global view count).
        Without an initialised process group this is a no-op with factor 1."""
        if not (dist.is_available() and dist.is_initialized()):
            return 1.0
        world = dist.get_world_size(group)
        _sum_flat_over_ranks(self.flat, group, method)
        return 1.0 / world if average else 1.0


def all_reduce_gradients(params: Dict[str, torch.Tensor], group=None, average: bool = True,
                         method: str = "all_reduce") -> float:
    """Sums ``param.grad`` of every tensor over the ranks and returns the factor left for the optimiser step to
    apply: 1 / world size, or 1.0 with ``average=False`` (losses already normalised over all ranks' views, as
    ``multiview_step`` does; see GradArena.all_reduce).  ``method="all_reduce"``: one collective per tensor, all in
    flight together (the 45-coefficient SH gradient is 3/4 of the bytes; coalescing the small ones buys nothing
    next to it).  ``"scatter_gather"`` / ``"direct"``: the gradients are packed into one flat buffer, exchanged as
    GradArena does, and unpacked (two extra copies of the gradients; gradients that already sit in a GradArena
    skip them: use its ``all_reduce``)."""
    if not (dist.is_available() and dist.is_initialized()):
        return 1.0
    world = dist.get_world_size(group)
    if world > 1 and method != "all_reduce":
        have = {k: p for k, p in params.items() if p.grad is not None}
        if have:
            arena = GradArena({k: p.grad for k, p in have.items()})
            for k, p in have.items():
                arena.views[k].copy_(p.grad)
            _sum_flat_over_ranks(arena.flat, group, method)
            for k, p in have.items():
                p.grad.copy_(arena.views[k])
    elif world > 1:
        host_side = dist.get_backend(group) == "gloo"
        pending = []
        for p in params.values():
            if p.grad is None:
                continue
            if host_side and p.grad.is_cuda:
                h = p.grad.cpu()                            # gloo only ever sees host tensors
                dist.all_reduce(h, op=dist.ReduceOp.SUM, group=group)
                p.grad.copy_(h)
            else:
                pending.append(dist.all_reduce(p.grad, op=dist.ReduceOp.SUM, group=group, async_op=True))
        for w in pending:
            w.wait()
    return 1.0 / world if average else 1.0


class GaussianAdam:
    """torch.optim.Adam as the reference configures it (scene/gaussian_model.py:154-163), fused.

    ``params``: name -> tensor, in the reference's group names (xyz, f_dc, f_rest, opacity, scaling,
    rotation); ``lrs``: name -> learning rate.  State (exp_avg, exp_avg_sq, step) is what
    torch.optim.Adam keeps per parameter; ``state_dict`` / ``load_state_dict`` exchange it in that
    optimiser's layout."""

    def __init__(self, params: Dict[str, torch.Tensor], lrs: Dict[str, float], betas=(0.9, 0.999),
                 eps: float = 1e-15):
        if len(params) > MAX_TENSORS:
            raise ValueError(f"at most {MAX_TENSORS} tensors per fused step")
        for name, p in params.items():
            if p.dtype != torch.float32 or not p.is_contiguous():
                raise ValueError(f"{name}: parameters must be contiguous float32 tensors")
        self.params, self.lrs = dict(params), dict(lrs)
        self.betas, self.eps = betas, eps
        self.step_count = 0
        self.exp_avg = {k: torch.zeros_like(p) for k, p in params.items()}
        self.exp_avg_sq = {k: torch.zeros_like(p) for k, p in params.items()}

    @classmethod
    def for_model(cls, model, opt: Optional[OptimizationParams] = None, spatial_lr_scale: float = 1.0):
        """The six groups of training_setup (scene/gaussian_model.py:154-161)."""
        opt = opt or OptimizationParams()
        params = {"xyz": model._xyz, "f_dc": model._features_dc, "f_rest": model._features_rest,
                  "opacity": model._opacity, "scaling": model._scaling, "rotation": model._rotation}
        lrs = {"xyz": opt.position_lr_init * spatial_lr_scale, "f_dc": opt.feature_lr, "f_rest": opt.feature_lr / 20.0,
               "opacity": opt.opacity_lr, "scaling": opt.scaling_lr, "rotation": opt.rotation_lr}
        o = cls(params, lrs)
        o.schedule = (opt.position_lr_init * spatial_lr_scale, opt.position_lr_final * spatial_lr_scale,
                      opt.position_lr_delay_mult, opt.position_lr_max_steps)
        return o

    def update_learning_rate(self, iteration: int) -> float:
        """scene/gaussian_model.py:169-175: only the position group is scheduled."""
        lr_init, lr_final, mult, max_steps = self.schedule
        lr = expon_lr(iteration, lr_init, lr_final, lr_delay_mult=mult, max_steps=max_steps)
        self.lrs["xyz"] = lr
        return lr

    # ---- the optimiser surgery of densification (scene/gaussian_model.py:258-327): the number of Gaussians
    # changes between steps, the moments of the survivors stay ------------------------------------------------
    @torch.no_grad()
    def replace_tensor(self, name: str, tensor: torch.Tensor) -> torch.nn.Parameter:
        """replace_tensor_to_optimizer (:258-271): a new value for one tensor, its moments start again at zero,
        the step count stays (torch.optim.Adam keeps ``state["step"]`` there too)."""
        p = torch.nn.Parameter(tensor.detach().contiguous().requires_grad_(True))
        self.params[name] = p
        self.exp_avg[name] = torch.zeros_like(p)
        self.exp_avg_sq[name] = torch.zeros_like(p)
        return p

    @torch.no_grad()
    def prune(self, keep: torch.Tensor) -> Dict[str, torch.nn.Parameter]:
        """_prune_optimizer (:273-289): keeps the rows selected by the boolean mask ``keep`` of every tensor and
        of both moments; returns the new parameters by group name."""
        for name, p in self.params.items():
            self.exp_avg[name] = self.exp_avg[name][keep].contiguous()
            self.exp_avg_sq[name] = self.exp_avg_sq[name][keep].contiguous()
            self.params[name] = torch.nn.Parameter(p.detach()[keep].contiguous().requires_grad_(True))
        return dict(self.params)

    @torch.no_grad()
    def cat_tensors(self, extension: Dict[str, torch.Tensor]) -> Dict[str, torch.nn.Parameter]:
        """cat_tensors_to_optimizer (:307-327): appends rows to every tensor; the new rows' moments are zero."""
        if set(extension) != set(self.params):
            raise ValueError(f"extension must name every tensor: {sorted(self.params)}")
        for name, p in self.params.items():
            ext = extension[name].detach().to(p.dtype)
            self.exp_avg[name] = torch.cat((self.exp_avg[name], torch.zeros_like(ext)), dim=0)
            self.exp_avg_sq[name] = torch.cat((self.exp_avg_sq[name], torch.zeros_like(ext)), dim=0)
            self.params[name] = torch.nn.Parameter(torch.cat((p.detach(), ext), dim=0).contiguous().requires_grad_(True))
        return dict(self.params)

    # ---- checkpoints: torch.optim.Adam's own layout (what the reference saves, scene/gaussian_model.py:73) ----
    def state_dict(self) -> dict:
        """The dictionary ``torch.optim.Adam(...).state_dict()`` returns for the reference's six groups
        (scene/gaussian_model.py:154-163): per parameter ``step`` / ``exp_avg`` / ``exp_avg_sq``, per group the
        hyper-parameters, ``name`` and the parameter's index."""
        names = list(self.params)
        state = {}
        if self.step_count > 0:
            for i, n in enumerate(names):
                state[i] = {"step": torch.tensor(float(self.step_count)), "exp_avg": self.exp_avg[n],
                            "exp_avg_sq": self.exp_avg_sq[n]}
        groups = [{"lr": float(self.lrs[n]), "name": n, "betas": tuple(self.betas), "eps": self.eps, "weight_decay": 0,
                   "amsgrad": False, "maximize": False, "foreach": None, "capturable": False, "differentiable": False,
                   "fused": None, "params": [i]} for i, n in enumerate(names)]
        return {"state": state, "param_groups": groups}

    @torch.no_grad()
    def load_state_dict(self, sd: dict) -> None:
        """Takes the moments, the step count and the learning rates from a ``torch.optim.Adam`` state dictionary
        whose groups carry the reference's names (or come in the same order)."""
        groups = sd["param_groups"]
        if len(groups) != len(self.params):
            raise ValueError(f"{len(groups)} parameter groups for {len(self.params)} tensors")
        steps = set()
        for i, (g, own) in enumerate(zip(groups, list(self.params))):
            name = g.get("name", own)
            if name not in self.params:
                raise ValueError(f"unknown parameter group {name!r}")
            if len(g["params"]) != 1:
                raise ValueError("one tensor per group, as the reference sets them up")
            p = self.params[name]
            self.lrs[name] = float(g["lr"])
            st = sd["state"].get(g["params"][0])
            if st is None:
                self.exp_avg[name], self.exp_avg_sq[name] = torch.zeros_like(p), torch.zeros_like(p)
                continue
            for key, store in (("exp_avg", self.exp_avg), ("exp_avg_sq", self.exp_avg_sq)):
                t = st[key]
                if tuple(t.shape) != tuple(p.shape):
                    raise ValueError(f"{name}.{key}: shape {tuple(t.shape)} does not match the parameter's {tuple(p.shape)}")
                store[name] = t.detach().to(device=p.device, dtype=torch.float32).contiguous().clone()
            steps.add(int(float(st["step"])))
        if len(steps) > 1:
            raise ValueError(f"the groups were stepped {sorted(steps)} times: one fused step cannot continue that")
        self.step_count = steps.pop() if steps else 0
        if groups and "betas" in groups[0]:
            self.betas, self.eps = tuple(groups[0]["betas"]), float(groups[0]["eps"])

    @torch.no_grad()
    def step(self, grads: Optional[Dict[str, torch.Tensor]] = None, grad_scale: float = 1.0) -> None:
        """One Adam update of every tensor that has a gradient (``grads[name]`` or ``param.grad``)."""
        for name, p in self.params.items():
            if not p.is_cuda:
                raise ValueError(f"{name}: the fused update runs on the GPU only, the parameter is on {p.device}")
        self.step_count += 1
        arr = (_AdamTensor * MAX_TENSORS)()
        n = 0
        keep: List[torch.Tensor] = []
        for name, p in self.params.items():
            g = grads[name] if grads is not None else p.grad
            if g is None:
                continue
            g = _dgr._f32(g, f"grad of {name}")
            if g.numel() != p.numel():
                raise ValueError(f"{name}: gradient has {g.numel()} elements, parameter {p.numel()}")
            keep.append(g)
            arr[n] = _AdamTensor(p.data_ptr(), g.data_ptr(), self.exp_avg[name].data_ptr(),
                                 self.exp_avg_sq[name].data_ptr(), p.numel(), float(self.lrs[name]), 0)
            n += 1
        _dgr._run("adam_step", _lib.gsr_adam_step, n, arr, float(self.betas[0]), float(self.betas[1]),
                  float(self.eps), int(self.step_count), float(grad_scale), _dgr._stream())

    def zero_grad(self, set_to_none: bool = True) -> None:
        """train.py:128."""
        for p in self.params.values():
            if set_to_none:
                p.grad = None
            elif p.grad is not None:
                p.grad.zero_()
