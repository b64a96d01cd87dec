"""Single-node data parallelism for the PEFT fine-tune step.

One process per GPU.  Every rank holds the full frozen backbone (never communicated) and a
replica of the trainable parameters; the global batch is split contiguously into ``world``
equal shards; the ONLY exchange per step is a sum all-reduce of the flat f32 gradient buffer
(KAdaptation ViT-B/32 + 100-class head: 101,476 floats = 406 KB) over RCCL/xGMI
(``torch.distributed`` backend "nccl" on ROCm; "gloo" in the CPU tests), after which every rank
applies the same SGD update with the gradient scaled by 1/world.

Caveats that define DP "parity" (they follow from the reference's semantics, not from this
engine): BatchNorm batch statistics and the raw-reshape scramble of the delta both
depend on the LOCAL batch, so DP=8x128 equals the mean of 8 independent reference steps at
bs=128, not one reference step at bs=1024; parameters whose reference ``.grad`` is None
(KAdaptation's v_proj_adapter1_*) keep zero slots in the flat buffer and are never updated.
"""
from __future__ import annotations

import torch
import torch.distributed as dist


def shard_bounds(global_batch: int, rank: int, world: int):
    """Contiguous equal shards (1024 -> 8 x 128); the global batch must divide evenly."""
    if global_batch % world:
        raise ValueError(f"global batch {global_batch} is not divisible by world size {world}")
    per = global_batch // world
    return rank * per, (rank + 1) * per


def shard_batch(images: torch.Tensor, labels: torch.Tensor, rank: int, world: int):
    lo, hi = shard_bounds(images.shape[0], rank, world)
    return images[lo:hi], labels[lo:hi]


def all_reduce_flat(flat_grads: torch.Tensor, group=None) -> float:
    """Sum-all-reduce the flat gradient buffer in place; returns the scale (1/world) that the
    optimizer step must apply (it is folded into the fused SGD kernel, not a separate pass)."""
    if not (dist.is_available() and dist.is_initialized()):
        return 1.0
    world = dist.get_world_size(group)
    if world > 1:
        dist.all_reduce(flat_grads, op=dist.ReduceOp.SUM, group=group)
    return 1.0 / world


def average_bn_buffers(running_mean: torch.Tensor, running_var: torch.Tensor, group=None):
    """BatchNorm running statistics are per-rank; average them before a checkpoint/validation
    (the reference has no opinion: it never runs multi-process)."""
    if not (dist.is_available() and dist.is_initialized()):
        return
    world = dist.get_world_size(group)
    if world > 1:
        for t in (running_mean, running_var):
            dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
            t.div_(world)


def sgd_momentum_(params, grads, momentum_buf, mask, lr, momentum, weight_decay, grad_scale, first_step, nesterov=False):
    """Host statement of the fused SGD kernel (misc.hip: sgd_kernel) on flat tensors; used by the
    CPU tests of the DP path.  Mirrors torch.optim.SGD(momentum, nesterov=False), skipping
    parameters that never receive a gradient."""
    d = grads * grad_scale + weight_decay * params
    buf = d if first_step else momentum * momentum_buf + d
    sel = mask.bool() if mask is not None else torch.ones_like(params, dtype=torch.bool)
    momentum_buf[sel] = buf[sel]
    upd = d + momentum * buf if nesterov else buf
    params[sel] -= lr * upd[sel]
