"""pevit_allreduce_flat (csrc/allreduce.hip, dp.FlatAllReduce): the DP gradient exchange without a collective-library kernel.

Two processes share the one GPU of the test box (the handles travel over a gloo group): IPC export / open of the mailboxes, the
push + flag protocol over several epochs (both parities, odd lengths, one element), the rank-ordered reduction, and the three
overlapped buckets of ``engine.forward_backward_dp`` -- bit-identical to the process group's all-reduce (a two-term f32 sum has
one value) and to the hand-summed two-shard step.  Two GPUs over xGMI have never run."""
import os
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT

pytestmark = pytest.mark.gpu


def _raw_worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from pevit_amd import dp
    torch.cuda.set_device(0)
    ar = dp.FlatAllReduce(101476)
    res = []
    for it, n in enumerate((101476, 1, 7, 4097, 101475, 64, 101476, 33)):
        g = torch.Generator().manual_seed(100 * it + rank)
        mine = torch.randn(n, generator=g)
        other = torch.randn(n, generator=torch.Generator().manual_seed(100 * it + (1 - rank)))
        buf = mine.cuda()
        ar.all_reduce(buf)
        torch.cuda.synchronize()
        want = (mine + other) if rank == 0 else (other + mine)          # rank order: contribution 0 + contribution 1
        assert torch.equal(buf.cpu(), want), (it, n, float((buf.cpu() - want).abs().max()))
        res.append(buf.cpu())
    ar.check()
    with pytest.raises(Exception):
        ar.all_reduce(torch.zeros(200000, device="cuda"))                # beyond the mailbox capacity: refused, not truncated
    torch.save(res, os.path.join(out_dir, f"raw{rank}.pt"))
    dist.barrier()
    ar.close()
    dist.destroy_process_group()


def test_flat_allreduce_two_processes_one_device(tmp_path):
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    world, port = 2, 29800 + (os.getpid() % 2000)
    mp.spawn(_raw_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    a, b = torch.load(tmp_path / "raw0.pt"), torch.load(tmp_path / "raw1.pt")
    assert len(a) == 8 and all(torch.equal(x, y) for x, y in zip(a, b))       # the replicas hold identical bits


def _step_worker(rank, world, port, case, flat, out_dir):
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from test_gpu_dp import LR, MOM, STEPS, WD, _batch, _make_engine
    eng, t = _make_engine(case, 4)
    eng.sync_replicas()
    if flat:
        eng.use_flat_allreduce()
    img, lab = _batch(t, rank, world)
    img, lab = img.cuda().contiguous(), lab.cuda().contiguous()
    for _ in range(STEPS):
        eng.train_step(img, lab, lr=LR, momentum=MOM, weight_decay=WD, world_size=world)
    torch.cuda.synchronize()
    if flat:
        eng._flat_ar.check()
    torch.save({"p": eng.params.cpu(), "g": eng.grads.cpu(), "m": eng.momentum.cpu()}, os.path.join(out_dir, f"{'flat' if flat else 'pg'}{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("case", ["tiny_kadaptation", "tiny_adapter"])
def test_dp_step_through_allreduce_flat_equals_the_process_group_route(case, tmp_path):
    """engine.forward_backward_dp with its three buckets exchanged by pevit_allreduce_flat on a side stream (use_flat_allreduce)
    against the same step through torch.distributed.all_reduce: parameters, momentum and gradients agree bit for bit on both
    ranks after three steps."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    world = 2
    for flat in (False, True):
        port = 29900 + (os.getpid() % 1000) + (500 if flat else 0)
        mp.spawn(_step_worker, args=(world, port, case, flat, str(tmp_path)), nprocs=world, join=True)
    for r in range(world):
        a, b = torch.load(tmp_path / f"flat{r}.pt"), torch.load(tmp_path / f"pg{r}.pt")
        for k in ("p", "g", "m"):
            assert torch.equal(a[k], b[k]), (r, k)
    f0, f1 = torch.load(tmp_path / "flat0.pt"), torch.load(tmp_path / "flat1.pt")
    assert all(torch.equal(f0[k], f1[k]) for k in ("p", "g", "m"))
