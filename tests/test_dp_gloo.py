"""Data-parallel path on CPU: world_size 2, gloo.  Two ranks run the oracle on their shard of the
fixture batch, pack gradients into the engine's flat layout, exchange them with the product's
``dp.all_reduce_flat`` and apply the SGD update; rank 0 checks the result against the mean of
the two shard gradients computed in-process (DP "parity" per SURVEY 8e: the mean of independent
per-shard reference steps)."""
import os
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT, golden_param_dict, load_golden


def _flat_grads(tr, spec_names):
    chunks = []
    for n in spec_names:
        g = tr.p[n].grad
        chunks.append(torch.zeros_like(tr.p[n]).flatten() if g is None else g.detach().flatten())
    chunks += [tr.head_w.grad.flatten(), tr.head_b.grad.flatten()]
    return torch.cat(chunks)


def _shard_grads(case, rank, world):
    from oracle import ref_cpu
    from pevit_amd import dp
    from pevit_amd.engine import adapter_param_spec
    meta, t = load_golden(case)
    sd = golden_param_dict(meta, t)
    tr = ref_cpu.OracleTrainer(sd, meta["method"], meta["classes"])
    with torch.no_grad():
        tr.head_w.copy_(t["head_w"]); tr.head_b.copy_(t["head_b"])
    img, lab = dp.shard_batch(t["images"], t["labels"], rank, world)
    tr.loss_and_grads(img, lab)
    names = [n for n, s, trn in adapter_param_spec(meta["method"], 128, 2, meta["lora_r"]) if trn]
    flat_p = torch.cat([tr.p[n].detach().flatten() for n in names] + [tr.head_w.detach().flatten(), tr.head_b.detach().flatten()])
    mask = torch.cat([torch.full((tr.p[n].numel(),), 0 if tr.p[n].grad is None else 1, dtype=torch.uint8) for n in names]
                     + [torch.ones(tr.head_w.numel() + tr.head_b.numel(), dtype=torch.uint8)])
    return _flat_grads(tr, names), flat_p, mask


def _worker(rank, world, port, case, out_dir):
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
    torch.set_num_threads(2)
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from pevit_amd import dp
    g, p, mask = _shard_grads(case, rank, world)
    scale = dp.all_reduce_flat(g)
    mom = torch.zeros_like(p)
    dp.sgd_momentum_(p, g, mom, mask, lr=0.01, momentum=0.9, weight_decay=1e-4, grad_scale=scale, first_step=True)
    torch.save({"g": g, "p": p, "scale": scale}, os.path.join(out_dir, f"rank{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("case", ["tiny_kadaptation", "tiny_lora"])
def test_two_rank_allreduce_matches_mean_of_shard_steps(case, tmp_path):
    world, port = 2, 29500 + (os.getpid() % 2000)
    mp.spawn(_worker, args=(world, port, case, str(tmp_path)), nprocs=world, join=True)
    r0 = torch.load(tmp_path / "rank0.pt"); r1 = torch.load(tmp_path / "rank1.pt")
    assert r0["scale"] == 0.5
    assert torch.equal(r0["g"], r1["g"]) and torch.equal(r0["p"], r1["p"])       # replicas stay identical
    from pevit_amd import dp
    g0, p0, mask = _shard_grads(case, 0, world)
    g1, _, _ = _shard_grads(case, 1, world)
    assert torch.allclose(r0["g"], g0 + g1, rtol=1e-3, atol=1e-5)     # thread-count dependent f32 sums
    mom = torch.zeros_like(p0)
    dp.sgd_momentum_(p0, 0.5 * (g0 + g1), mom, mask, 0.01, 0.9, 1e-4, 1.0, True)
    assert torch.allclose(r0["p"], p0, rtol=1e-4, atol=1e-6)
    dead = mask == 0
    if dead.any():            # KAdaptation: v_proj_adapter1_* never move
        _, p_init, _ = _shard_grads(case, 0, world)
        assert torch.equal(r0["p"][dead], p_init[dead])


def test_shard_bounds():
    from pevit_amd import dp
    assert [dp.shard_bounds(1024, r, 8) for r in (0, 7)] == [(0, 128), (896, 1024)]
    with pytest.raises(ValueError):
        dp.shard_bounds(10, 0, 4)
