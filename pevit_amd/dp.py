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


class FlatAllReduce:
    """pevit_allreduce_flat (include/pevit_hip.h, csrc/allreduce.hip): sum all-reduce of a flat f32 device buffer by one-shot
    pushes into IPC-mapped peer mailboxes + a deterministic local reduction -- no collective-library kernel competes with the
    one-tile-per-CU GEMMs of the overlapped backward (profiles/r04_dp_evidence.md).  The IPC handles travel through the
    process group's object collective once, at construction (any backend: gloo in the tests, nccl in production)."""

    def __init__(self, max_floats: int, group=None, device=None):
        import ctypes as C
        from . import _lib
        self._C, self._lib_mod = C, _lib
        self.lib = _lib.load()
        self.group = group
        self.rank, self.world = dist.get_rank(group), dist.get_world_size(group)
        if device is not None:
            torch.cuda.set_device(device)
        self._ar = C.c_void_p()
        _lib.check(self.lib.pevit_ar_create(C.byref(self._ar), self.rank, self.world, int(max_floats)), "pevit_ar_create")
        hb = self.lib.pevit_ar_handle_bytes()
        mine = C.create_string_buffer(hb)
        _lib.check(self.lib.pevit_ar_export(self._ar, mine), "pevit_ar_export")
        handles = [None] * self.world
        dist.all_gather_object(handles, bytes(mine.raw), group=group)
        for p, h in enumerate(handles):
            if p != self.rank:
                _lib.check(self.lib.pevit_ar_import(self._ar, p, C.create_string_buffer(h, hb)), "pevit_ar_import")
        dist.barrier(group=group)                       # every mailbox is open everywhere before the first push
        self.fine_grained = bool(self.lib.pevit_ar_fine_grained(self._ar))

    def error_word(self):
        """Device address of the error word (for engine / pevit_set_external_poison)."""
        return self.lib.pevit_ar_error_word(self._ar)

    def all_reduce(self, buf: torch.Tensor, stream=None):
        """In place, asynchronous on ``stream`` (default: the current stream).  Same stream, same call order on every rank."""
        if buf.dtype != torch.float32 or not buf.is_contiguous() or not buf.is_cuda:
            raise self._lib_mod.PevitError("FlatAllReduce.all_reduce: a contiguous float32 device tensor is required")
        s = self._C.c_void_p((stream or torch.cuda.current_stream()).cuda_stream)
        self._lib_mod.check(self.lib.pevit_allreduce_flat(self._ar, s, self._C.c_void_p(buf.data_ptr()), buf.numel()),
                            "pevit_allreduce_flat")

    def check(self, stream=None):
        """Raise if a reduction gave up (synchronises ``stream``: pass the stream the all-reduces run on).  The word stays raised --
        and the engine's optimizer updates withheld -- until every rank has gone through ``resync()`` (then ``sync_replicas()``:
        the peers that did receive every flag applied an update this rank withheld)."""
        s = self._C.c_void_p((stream or torch.cuda.current_stream()).cuda_stream)
        rc = self.lib.pevit_ar_error(self._ar, s)
        if rc != 0:
            raise self._lib_mod.PevitError({1: "pevit_allreduce_flat: a peer's contribution never arrived (that bucket was left unreduced; "
                                               "the optimizer update of the step was withheld if the engine holds the error word)",
                                            2: "pevit_allreduce_flat: the ranks passed different sizes for the same all-reduce"}.get(
                                               rc, "pevit_ar_error failed"))

    def resync(self):
        """After an error (check() raised): bring every rank's protocol state back to the start.  Collective."""
        torch.cuda.synchronize()
        dist.barrier(group=self.group)
        self._lib_mod.check(self.lib.pevit_ar_reset(self._ar, self._C.c_void_p(torch.cuda.current_stream().cuda_stream)), "pevit_ar_reset")
        dist.barrier(group=self.group)

    def close(self, barrier: bool = True):
        """Peers may still have pushes in flight into this mailbox: every rank drains its device and meets the others first
        (a collective: call it on every rank; the finalizer skips the meeting, it cannot know the other ranks are there)."""
        if getattr(self, "_ar", None):
            try:
                torch.cuda.synchronize()
                if barrier and dist.is_initialized():
                    dist.barrier(group=self.group)
            except Exception:
                pass                                   # interpreter shutdown / a torn-down group: free anyway
            self.lib.pevit_ar_destroy(self._ar)
            self._ar = None

    def __del__(self):
        try:
            self.close(barrier=False)
        except Exception:
            pass


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
