"""Data parallelism of the HIP engine with more than one rank.

Two processes share the one GPU of the test box (RCCL refuses two ranks on one device, so the process group is gloo
carrying DEVICE tensors); each drives ``HipEngine.train_step(world_size=2)`` -- the code bench.py runs for --gpus N > 1 --
on its contiguous shard of a fixture batch.  Checked: replicas stay bit-identical, the result equals the mean of the two
independent single-rank shard steps (the DP rule of SURVEY 8e: BatchNorm statistics and the delta scramble follow the LOCAL
batch), the 1/world scale is applied by the fused SGD kernel, parameters whose reference .grad is None never move, and
``sync_replicas`` removes a deliberate initial divergence."""
import os
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT, golden_param_dict, load_golden

pytestmark = pytest.mark.gpu
STEPS, LR, MOM, WD = 3, 0.05, 0.9, 1e-3


def _make_engine(case, batch):
    from pevit_amd.engine import HipEngine
    from pevit_amd.synth import ARCHS
    meta, t = load_golden(case)
    eng = HipEngine(ARCHS[meta["arch"]], meta["method"], meta["classes"], batch, lora_rank=meta["lora_r"])
    eng.load_state_dict(golden_param_dict(meta, t))
    v = eng.param_views()
    with torch.no_grad():
        v["layers.0.weight"].copy_(t["head_w"]); v["layers.0.bias"].copy_(t["head_b"])
    return eng, t


def _batch(t, rank, world):
    """the fixture's 4 images twice over -> 8 images, 4 per rank (BatchNorm needs more than one sample per shard)"""
    from pevit_amd import dp
    img = torch.cat([t["images"], t["images"].flip(0) * 0.5]); lab = torch.cat([t["labels"], t["labels"].flip(0)])
    return dp.shard_batch(img, lab, rank, world)


def _worker(rank, world, port, case, out_dir, mode):
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    eng, t = _make_engine(case, 4)
    eng.dp_exchange_mode = mode
    if rank == 1:
        eng.params.mul_(1.5); eng.running_mean.add_(3.0)          # a diverged replica ...
    eng.sync_replicas()                                            # ... is brought back to rank 0's state
    img, lab = _batch(t, rank, world)
    img, lab = img.cuda().contiguous(), lab.cuda().contiguous()
    losses = []
    for _ in range(STEPS):
        _, loss = eng.train_step(img, lab, lr=LR, momentum=MOM, weight_decay=WD, world_size=world)
        losses.append(float(loss))
    torch.cuda.synchronize()
    torch.save({"p": eng.params.cpu(), "g": eng.grads.cpu(), "m": eng.momentum.cpu(), "losses": losses,
                "rm": eng.running_mean.cpu()}, os.path.join(out_dir, f"rank{rank}.pt"))
    eng.average_bn_buffers()
    torch.save(eng.running_mean.cpu(), os.path.join(out_dir, f"rm_avg{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


# mode: "single" = the fused call + one all-reduce of the flat gradient buffer (the default since round 5); "staged" = the backward in
# two halves with three overlapped buckets; "pipelined" = single with the exchange + SGD on a second stream under the next step's stem
@pytest.mark.parametrize("mode", ["single", "staged", "pipelined"])
@pytest.mark.parametrize("case", ["tiny_kadaptation", "tiny_lora", "tiny_adapter"])
def test_two_rank_hip_step_equals_mean_of_shard_steps(case, mode, tmp_path):
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    world, port = 2, 29600 + (os.getpid() % 2000) + {"single": 0, "staged": 11, "pipelined": 23}[mode]
    mp.spawn(_worker, args=(world, port, case, str(tmp_path), mode), nprocs=world, join=True)
    r0 = torch.load(tmp_path / "rank0.pt"); r1 = torch.load(tmp_path / "rank1.pt")
    for k in ("p", "g", "m"):
        assert torch.equal(r0[k], r1[k]), k                      # replicas bit-identical after 3 steps
    assert not torch.equal(r0["rm"], r1["rm"])                  # BatchNorm statistics are local ...
    a0, a1 = torch.load(tmp_path / "rm_avg0.pt"), torch.load(tmp_path / "rm_avg1.pt")
    assert torch.equal(a0, a1) and torch.allclose(a0, 0.5 * (r0["rm"] + r1["rm"]))       # ... until averaged
    # the same computation without a process group: two engines, one per shard, gradients summed by hand and the fused
    # SGD kernel applied with grad_scale = 1/2 -- must agree bit for bit (a 2-term f32 sum has one order)
    e0, t = _make_engine(case, 4); e1, _ = _make_engine(case, 4)
    shards = [tuple(x.cuda().contiguous() for x in _batch(t, r, world)) for r in range(world)]
    p_init = e0.params.clone()
    for step in range(STEPS):
        l0 = float(e0.forward_backward(*shards[0])[1]); l1 = float(e1.forward_backward(*shards[1])[1])
        assert l0 == r0["losses"][step] and l1 == r1["losses"][step]
        total = e0.grads + e1.grads
        for e in (e0, e1):
            e.grads.copy_(total)
            e.sgd_step(LR, MOM, WD, 1.0 / world)
    torch.cuda.synchronize()
    assert torch.equal(e0.params.cpu(), r0["p"]) and torch.equal(e0.momentum.cpu(), r0["m"])
    dead = ~e0.grad_mask.bool().cpu()
    if dead.any():                                              # KAdaptation's v_proj_adapter1_*: never touched
        assert torch.equal(r0["p"][dead], p_init.cpu()[dead])
    assert float((r0["p"] - p_init.cpu()).abs().max()) > 0


@pytest.mark.parametrize("exchange", ["rccl", "flat"])
def test_bench_multi_rank_launch_contract(tmp_path, exchange):
    """bench.py exactly as the driver launches it for N > 1 (python -m torch.distributed.run --nnodes=1 --nproc-per-node N
    --master-addr 127.0.0.1 ... bench.py --gpus N --steps K --warmup W), two ranks on the one GPU of the test box (gloo
    instead of RCCL, both ranks on cuda:0): rank 0 prints ONE JSON line whose value is the whole-job aggregate."""
    import json
    import subprocess
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    port = 29700 + (os.getpid() % 2000) + (7 if exchange == "flat" else 0)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1",
           "--arch", "tiny-256", "--batch", "16", "--dist-backend", "gloo", "--share-device", "--exchange", exchange]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 3 and d["warmup"] == 1 and d["scaling"] == "weak"
    assert d["config"]["global_batch"] == 32 and d["config"]["parallelism"] == "dp2" and d["config"]["gradient_exchange"] == exchange
    assert abs(d["value"] - 32 * 3 / (d["ms_per_step"] * 3e-3)) < 1e-6 * d["value"]      # whole job: all ranks' images / time
    assert "cpu_baseline" not in d                                                          # rank 0 at N = 1 only
    assert d["roofline"]["achieved"] > 0
    # the line diagnoses itself: both schedules timed, per-rank step times, the exchange's device time
    c = d["config"]
    assert set(c["exchange_schedules"]) == {"single", "staged"} and c["gradient_exchange_schedule"]
    best = max(c["exchange_schedules"].values(), key=lambda v: v["images_per_sec"])
    assert abs(best["images_per_sec"] - d["value"]) < 1e-6 * d["value"]
    for v in c["exchange_schedules"].values():
        assert len(v["per_rank_ms_per_step"]) == 2 and v["exchange_us_per_step"] > 0
    assert len(c["per_rank_ms_per_step"]) == 2 and c["exchange_us_per_step"] > 0
