#!/usr/bin/env python3
"""Pieces of the host -> device feed of train_one (DeviceFeeder): gather into pinned memory (three ways, 1 and 4 threads), upload,
and the feeder end to end."""
import os, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from pevit_amd.evaluation._harness import DeviceFeeder
from pevit_amd.evaluation.dataloader import TensorLoader, _Tensors
bs, steps = int(sys.argv[1]) if len(sys.argv) > 1 else 128, 20
n = bs * steps
print("torch threads", torch.get_num_threads(), "cpus", os.cpu_count())
u8 = torch.randint(0, 256, (n, 3, 224, 224), dtype=torch.uint8); lbl = torch.arange(n) % 100
pins = [torch.empty((bs, 3, 224, 224), dtype=torch.uint8).pin_memory() for _ in range(4)]
sels = [torch.randperm(n)[:bs] for _ in range(4)]
dst = torch.empty((bs, 3, 224, 224), dtype=torch.uint8, device="cuda")
def t(f, k=10):
    f(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(k): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / k * 1e3
def g_take(i): np.take(u8.numpy().reshape(n, -1), sels[i].numpy(), axis=0, out=pins[i].numpy().reshape(bs, -1), mode="clip")
def g_rows(i):
    for r, j in enumerate(sels[i].tolist()): pins[i][r].copy_(u8[j])
def g_isel(i): torch.index_select(u8, 0, sels[i], out=pins[i])
def par(fn, w):
    def run():
        th = [threading.Thread(target=fn, args=(i,)) for i in range(w)]
        [x.start() for x in th]; [x.join() for x in th]
    return run
for name, fn in (("np.take", g_take), ("row copy_", g_rows), ("index_select", g_isel)):
    print(f"{name:14s} -> pinned: 1 thread {t(lambda: fn(0)):.2f} ms / batch of {bs};  4 threads, 4 batches {t(par(fn, 4), 5):.2f} ms")
print(f"pinned -> device         {t(lambda: dst.copy_(pins[0], non_blocking=True)):.2f} ms ({pins[0].numel() / 1e6:.1f} MB)")
loader = TensorLoader(_Tensors(u8, lbl), batch_size=bs, shuffle=True)
for rep in range(2):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for a, b in DeviceFeeder(loader, 0):
        pass
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f"DeviceFeeder alone: {dt / steps * 1e3:.2f} ms / batch -> {n / dt:.0f} images/s")
