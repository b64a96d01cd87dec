"""pevit_allreduce_flat (csrc/allreduce.hip, dp.FlatAllReduce): the DP gradient exchange without a collective-library kernel.

Two, four and eight processes share the one GPU of the test box (the handles travel over a gloo group): IPC export / open of the
mailboxes, the push + flag protocol over several epochs (both parities, odd lengths, one element), the rank-ordered reduction,
the error paths (ranks that disagree on the size; a peer that never arrives; resync afterwards; the optimizer update withheld
while the error word is raised) and the three overlapped buckets of ``engine.forward_backward_dp`` -- bit-identical to the
process group's all-reduce (a two-term f32 sum has one value) and to the hand-summed two-shard step.  The push is a kernel with a system-scope release (round 5; the
copy-engine push of round 4 lost flags in an eight-process soak).  Two GPUs over xGMI have never run."""
import os
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT

pytestmark = pytest.mark.gpu


def _contrib(it, r, n):
    return torch.randn(n, generator=torch.Generator().manual_seed(100 * it + r))


def _raw_worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from pevit_amd import dp
    torch.cuda.set_device(0)
    ar = dp.FlatAllReduce(101476)
    res = []
    for it, n in enumerate((101476, 1, 7, 4097, 101475, 64, 101476, 33)):
        buf = _contrib(it, rank, n).cuda()
        ar.all_reduce(buf)
        torch.cuda.synchronize()
        want = torch.zeros(n)
        for r in range(world):                                           # rank order: ((0 + c0) + c1) + ... in f32, on every rank
            want = want + _contrib(it, r, n)
        assert torch.equal(buf.cpu(), want), (it, n, float((buf.cpu() - want).abs().max()))
        res.append(buf.cpu())
    ar.check()
    with pytest.raises(Exception):
        ar.all_reduce(torch.zeros(200000, device="cuda"))                # beyond the mailbox capacity: refused, not truncated
    # the refusal did not advance the epoch: the next exchange still works
    buf = torch.full((16,), float(rank + 1), device="cuda")
    ar.all_reduce(buf); torch.cuda.synchronize()
    assert torch.equal(buf.cpu(), torch.full((16,), float(world * (world + 1) // 2)))
    # ranks that disagree on the size: every rank sees the mismatch, nothing is summed, the error word says 2
    n_bad = 64 if rank == 0 else 32
    buf = torch.ones(n_bad, device="cuda")
    ar.all_reduce(buf); torch.cuda.synchronize()
    assert torch.equal(buf.cpu(), torch.ones(n_bad))
    with pytest.raises(Exception, match="different sizes"):
        ar.check()
    ar.resync()
    buf = torch.full((5,), 2.0, device="cuda")
    ar.all_reduce(buf); torch.cuda.synchronize()
    assert torch.equal(buf.cpu(), torch.full((5,), 2.0 * world))
    ar.check()
    torch.save(res, os.path.join(out_dir, f"raw{rank}.pt"))
    dist.barrier()
    ar.close()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 4, 8])
def test_flat_allreduce_processes_on_one_device(world, tmp_path):
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    port = 29800 + (os.getpid() % 2000) + world
    mp.spawn(_raw_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    r0 = torch.load(tmp_path / "raw0.pt")
    for r in range(1, world):
        b = torch.load(tmp_path / f"raw{r}.pt")
        assert len(b) == 8 and all(torch.equal(x, y) for x, y in zip(r0, b))  # the replicas hold identical bits


def _timeout_worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from pevit_amd import dp
    torch.cuda.set_device(0)
    ar = dp.FlatAllReduce(1024)
    buf = torch.full((8,), 3.0, device="cuda")
    if rank == 0:
        ar.all_reduce(buf)                       # rank 1 never pushes: the reducing launch gives up after its bounded wait
        torch.cuda.synchronize()
        assert torch.equal(buf.cpu(), torch.full((8,), 3.0))             # left as it was, everywhere or nowhere
        with pytest.raises(Exception, match="never arrived"):
            ar.check()
    ar.resync()
    buf = torch.full((8,), 1.0 + rank, device="cuda")
    ar.all_reduce(buf); torch.cuda.synchronize()
    assert torch.equal(buf.cpu(), torch.full((8,), 3.0))
    ar.check()
    dist.barrier()
    ar.close()
    dist.destroy_process_group()


def test_flat_allreduce_gives_up_on_a_missing_peer_and_resyncs(tmp_path):
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    mp.spawn(_timeout_worker, args=(2, 29700 + (os.getpid() % 2000), str(tmp_path)), nprocs=2, join=True)


def test_external_error_word_withholds_the_update_and_marks_the_loss():
    """pevit_set_external_poison: while a caller-owned device word is non-zero (the all-reduce's error word under DP) the fused SGD
    kernel leaves parameters and momentum alone, counts the withheld update, and the loss of that step reads NaN."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import ctypes as C
    from pevit_amd import _lib
    from test_gpu_dp import _make_engine
    eng, t = _make_engine("tiny_kadaptation", 4)
    img, lab = t["images"].cuda().contiguous(), t["labels"].cuda().contiguous()
    word = torch.zeros(1, dtype=torch.int32, device="cuda")
    _lib.check(eng.lib.pevit_set_external_poison(eng._ctx, C.c_void_p(word.data_ptr())), "pevit_set_external_poison")
    eng.train_step(img, lab, lr=0.01)
    torch.cuda.synchronize()
    p1 = eng.params.clone()
    word.fill_(1)
    _, loss = eng.train_step(img, lab, lr=0.01)
    torch.cuda.synchronize()
    assert torch.equal(eng.params, p1) and bool(torch.isnan(loss).all())
    word.zero_()
    _, loss = eng.train_step(img, lab, lr=0.01)
    torch.cuda.synchronize()
    assert not torch.equal(eng.params, p1) and bool(torch.isfinite(loss).all())
    _lib.check(eng.lib.pevit_set_external_poison(eng._ctx, None), "pevit_set_external_poison")


def _step_worker(rank, world, port, case, flat, out_dir, mode="staged"):
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from test_gpu_dp import LR, MOM, STEPS, WD, _batch, _make_engine
    eng, t = _make_engine(case, 4)
    eng.dp_exchange_mode = mode
    eng.sync_replicas()
    if flat:
        eng.use_flat_allreduce()
    img, lab = _batch(t, rank, world)
    img, lab = img.cuda().contiguous(), lab.cuda().contiguous()
    for _ in range(STEPS):
        eng.train_step(img, lab, lr=LR, momentum=MOM, weight_decay=WD, world_size=world)
    torch.cuda.synchronize()
    if flat:
        eng.check_streamk()                      # covers the exchange's error word too (on its own stream)
    torch.save({"p": eng.params.cpu(), "g": eng.grads.cpu(), "m": eng.momentum.cpu()}, os.path.join(out_dir, f"{'flat' if flat else 'pg'}{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("mode", ["staged", "single", "pipelined"])
@pytest.mark.parametrize("case", ["tiny_kadaptation", "tiny_adapter"])
def test_dp_step_through_allreduce_flat_equals_the_process_group_route(case, mode, tmp_path):
    """engine.forward_backward_dp with its buckets (three overlapped ones, "staged"; or the whole flat buffer behind the fused call,
    "single") exchanged by pevit_allreduce_flat on a side stream (use_flat_allreduce) against the same step through
    torch.distributed.all_reduce: parameters, momentum and gradients agree bit for bit on both ranks after three steps."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    world = 2
    for flat in (False, True):
        port = 29900 + (os.getpid() % 1000) + (500 if flat else 0) + {"staged": 0, "single": 250, "pipelined": 125}[mode]
        mp.spawn(_step_worker, args=(world, port, case, flat, str(tmp_path), mode), nprocs=world, join=True)
    for r in range(world):
        a, b = torch.load(tmp_path / f"flat{r}.pt"), torch.load(tmp_path / f"pg{r}.pt")
        for k in ("p", "g", "m"):
            assert torch.equal(a[k], b[k]), (r, k)
    f0, f1 = torch.load(tmp_path / "flat0.pt"), torch.load(tmp_path / "flat1.pt")
    assert all(torch.equal(f0[k], f1[k]) for k in ("p", "g", "m"))
