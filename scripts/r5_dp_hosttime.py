#!/usr/bin/env python3
"""Does the process group's all_reduce hold the HOST until the device reaches it?  Host time of each part of the single-exchange DP
step on a 1-rank RCCL group (ViT-B/32 + KAdaptation, batch 128): the fused forward/backward call, all_reduce(async_op=True), its
wait(), the SGD call -- with the device several milliseconds behind the host."""
import os, sys, time, socket
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import torch.distributed as dist
from pevit_amd.engine import HipEngine
from pevit_amd.synth import ARCHS, synth_batch, synth_state_dict

arch = ARCHS["ViT-B/32"]
eng = HipEngine(arch, "kadaptation", 100, 128)
eng.load_state_dict({k: v for k, v in synth_state_dict(arch, seed=0, text_tower=False).items() if k.startswith("visual.")})
images, labels = synth_batch(128, arch.resolution, 100, seed_img=1, seed_lbl=2)
images, labels = images.cuda(), labels.cuda()
with socket.socket() as sk:
    sk.bind(("127.0.0.1", 0)); port = sk.getsockname()[1]
dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1, device_id=torch.device("cuda:0"))
for _ in range(5):
    eng.train_step(images, labels, 0.01)
torch.cuda.synchronize()
acc = [0.0] * 4
N = 30
for _ in range(N):
    t0 = time.perf_counter(); eng.forward_backward(images, labels)
    t1 = time.perf_counter(); w = dist.all_reduce(eng.grads, async_op=True)
    t2 = time.perf_counter(); w.wait()
    t3 = time.perf_counter(); eng.sgd_step(0.01, 0.9, 1e-6, 1.0)
    t4 = time.perf_counter()
    for i, d in enumerate((t1 - t0, t2 - t1, t3 - t2, t4 - t3)):
        acc[i] += d
torch.cuda.synchronize()

# device time per step (events on the compute stream, median of 60) of variations of the step
side = torch.cuda.Stream()
def fused():
    eng.forward_backward(images, labels); eng.sgd_step(0.01, 0.9, 1e-6, 1.0)
def hop():          # the same with one round trip through another stream between the backward and the update, nothing on it
    eng.forward_backward(images, labels)
    cur = torch.cuda.current_stream()
    e1 = torch.cuda.Event(); e1.record(cur); side.wait_event(e1)
    e2 = torch.cuda.Event(); e2.record(side); cur.wait_event(e2)
    eng.sgd_step(0.01, 0.9, 1e-6, 1.0)
def hop_reuse():    # ... with two long-lived events instead of fresh ones
    eng.forward_backward(images, labels)
    cur = torch.cuda.current_stream()
    E1.record(cur); side.wait_event(E1); E2.record(side); cur.wait_event(E2)
    eng.sgd_step(0.01, 0.9, 1e-6, 1.0)
def single():
    eng.forward_backward(images, labels); dist.all_reduce(eng.grads, async_op=True).wait(); eng.sgd_step(0.01, 0.9, 1e-6, 1.0)
def single_sync():
    eng.forward_backward(images, labels); dist.all_reduce(eng.grads); eng.sgd_step(0.01, 0.9, 1e-6, 1.0)
E1, E2 = torch.cuda.Event(), torch.cuda.Event()
for name, fn in (("fused", fused), ("fused + empty round trip through a second stream (fresh events)", hop), ("... (two long-lived events)", hop_reuse),
                 ("all_reduce(async_op=True).wait()", single), ("all_reduce()", single_sync), ("fused again", fused)):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    mk = [torch.cuda.Event(enable_timing=True) for _ in range(61)]
    mk[0].record()
    for i in range(60):
        fn(); mk[i + 1].record()
    torch.cuda.synchronize()
    v = sorted(mk[i].elapsed_time(mk[i + 1]) for i in range(60))
    print(f"device ms per step, {name}: median {v[30]:.3f}  (p10 {v[6]:.3f}, p90 {v[54]:.3f})")
print("host ms per step: forward_backward call %.3f | all_reduce(async_op=True) %.3f | wait() %.3f | sgd_step %.3f" % tuple(a / N * 1e3 for a in acc))
dist.destroy_process_group()
