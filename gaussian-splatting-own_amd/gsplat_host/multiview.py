"""Multi-GPU: independent views sharded one process per GPU.

``render()`` is a pure function of (camera, Gaussians)
(reference ``gaussian_renderer/__init__.py:18-100``; the reference itself walks
views one at a time, ``train.py:76-86``, ``render.py:31-32``), so a batch of
views splits across ranks with the Gaussians replicated and NO collective on
the render path.  The only exchange is the scalar loss: one fp32 all-reduce
(RCCL over xGMI when the process group is ``nccl``), latency-bound.

The reference has no distributed code; this file is new design, not parity
work.
"""
from __future__ import annotations

from typing import Callable, List, Sequence

import torch
import torch.distributed as dist


def views_for_rank(n_views: int, rank: int, world: int) -> List[int]:
    """Round-robin: view v is rendered by rank v mod world."""
    if world <= 0 or not 0 <= rank < world:
        raise ValueError(f"bad rank/world {rank}/{world}")
    return list(range(rank, n_views, world))


def mean_loss_over_ranks(local_sum: torch.Tensor, n_views: int) -> torch.Tensor:
    """Sum of the ranks' local loss sums divided by the global view count.
    ``gloo`` only ever sees host tensors (device tensors are copied first)."""
    t = local_sum.detach().clone().reshape(1)
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        if dist.get_backend() == "gloo" and t.is_cuda:
            h = t.cpu()
            dist.all_reduce(h)
            t = h.to(t.device)
        else:
            dist.all_reduce(t)
    return t[0] / n_views


def multiview_step(cameras: Sequence, render_view: Callable, loss_fn: Callable, targets: Sequence,
                   rank: int = 0, world: int = 1, backward: bool = True):
    """Renders this rank's share of ``cameras``, back-propagates each view's
    loss (scaled by 1/len(cameras), so that gradients are those of the mean
    loss over ALL views once they are summed over ranks: exchange them with
    ``all_reduce_gradients(..., average=False)`` / ``GradArena.all_reduce(average=False)``,
    nothing is left for the optimiser step to scale), and returns
    (mean loss over all views and ranks, list of this rank's view indices)."""
    mine = views_for_rank(len(cameras), rank, world)
    total = None
    for v in mine:
        image = render_view(cameras[v])
        loss = loss_fn(image, targets[v])
        if backward and loss.requires_grad:
            (loss / len(cameras)).backward()
        total = loss.detach() if total is None else total + loss.detach()
    if total is None:
        dev = targets[0].device if len(targets) else "cpu"
        total = torch.zeros((), device=dev)
    return mean_loss_over_ranks(total, len(cameras)), mine
