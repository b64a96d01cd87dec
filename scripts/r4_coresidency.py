#!/usr/bin/env python3
"""DP evidence that needs one GPU.
 (i)  the DP ROUTE (engine.forward_backward_dp: staged backward, stream-K off, three bucketed all-reduces on a 1-rank RCCL group)
      against the fused step;
 (ii) CO-RESIDENCY: k workgroups of a side-stream kernel hold CU slots during the timed steps, as the RCCL kernels of an
      overlapped all-reduce would -- without LDS (they only take issue slots; the 512-thread GEMM workgroups still fit beside
      them) and with 64 KiB of LDS each (a one-tile-per-CU GEMM workgroup, 144 KiB, no longer fits on that CU).
Prints a markdown table."""
import ctypes as C, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
from pevit_amd.engine import HipEngine
from pevit_amd.synth import ARCHS, reference_init_, synth_batch, synth_state_dict

B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
dev = torch.device("cuda", 0); torch.cuda.set_device(dev)
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
torch.distributed.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
arch = ARCHS["ViT-B/32"]
eng = HipEngine(arch, "kadaptation", 100, B, device=dev)
eng.load_state_dict(synth_state_dict(arch, seed=2, text_tower=False))
reference_init_(eng.param_views().items(), "kadaptation")
images, labels = synth_batch(B, 224, 100); images, labels = images.to(dev), labels.to(dev)
side = torch.cuda.Stream(dev)
STEPS = 40

def fused():
    eng.forward_backward(images, labels); eng.sgd_step(0.01, 0.9, 1e-6)
def dp_route():
    eng.forward_backward_dp(images, labels); eng.sgd_step(0.01, 0.9, 1e-6, 1.0)

def timed(fn, occupy=None):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    if occupy:
        k, lds = occupy
        assert eng.lib.pevit_debug_occupy(C.c_void_p(side.cuda_stream), k, lds, float(STEPS * 8000.0)) == 0
        time.sleep(0.002)
    marks = [torch.cuda.Event(enable_timing=True) for _ in range(STEPS + 1)]
    marks[0].record()
    for i in range(STEPS):
        fn(); marks[i + 1].record()
    marks[-1].synchronize()
    ms = sorted(marks[i].elapsed_time(marks[i + 1]) for i in range(STEPS))
    torch.cuda.synchronize()            # the occupier runs out on its own
    return ms[len(ms) // 2]

print(f"ViT-B/32 + KAdaptation, B = {B}, median of {STEPS} steps (HIP events), one MI355X\n")
base = timed(fused)
print("| configuration | ms / step | vs fused step |"); print("|---|---|---|")
print(f"| fused step (pevit_train_forward_backward + SGD) | {base:.3f} | 1.000 |")
eng.tune("gemm_streamk", 0); nosk = timed(fused)
print(f"| fused step, stream-K off (what DP uses) | {nosk:.3f} | {nosk / base:.3f} |")
dpr = timed(dp_route)
print(f"| DP route: forward_backward_dp, 1-rank RCCL group (staged backward, 3 async all-reduces) | {dpr:.3f} | {dpr / base:.3f} |")
base2 = timed(fused)
print(f"| fused step, stream-K off, again (drift check) | {base2:.3f} | {base2 / base:.3f} |")
print("\n| side-stream workgroups held during the steps | LDS each | ms / step (fused, stream-K off) | slowdown |"); print("|---|---|---|---|")
for lds in (0, 65536):
    for k in (4, 8, 16, 32, 64):
        t = timed(fused, (k, lds))
        print(f"| {k} | {lds // 1024} KiB | {t:.3f} | {t / base2:.3f} |", flush=True)
t = timed(dp_route, (16, 65536))
print(f"| 16 (DP route) | 64 KiB | {t:.3f} | {t / dpr:.3f} vs the DP route alone |")
torch.distributed.destroy_process_group()
