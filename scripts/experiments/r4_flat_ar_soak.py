#!/usr/bin/env python3
"""Soak of pevit_allreduce_flat: two processes sharing one GPU, 1000 all-reduces of the three DP bucket sizes back to back
(no host synchronisation in between), every result checked against the known sum; reports the error word and the mean time per
all-reduce (both ranks on ONE device: a protocol check, not an xGMI measurement)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch, torch.distributed as dist, torch.multiprocessing as mp

def worker(rank, world, port, iters):
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from pevit_amd import dp
    torch.cuda.set_device(0)
    ar = dp.FlatAllReduce(101476)
    sizes = (51300, 23040, 27136)                       # head | upper half | rules + lower half of KAdaptation ViT-B/32
    bufs = [[torch.full((n,), float(rank + 1), device="cuda") for n in sizes] for _ in range(2)]
    bad = 0
    torch.cuda.synchronize(); dist.barrier(); t0 = time.perf_counter()
    for it in range(iters):
        cur = bufs[it & 1]
        for b in cur:
            b.fill_(float(rank + 1) * (it % 7 + 1))
            ar.all_reduce(b)
        if it % 50 == 49:                                # check a window, then go on
            torch.cuda.synchronize()
            want = 3.0 * (it % 7 + 1)
            bad += sum(int((b != want).sum()) for b in cur)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    ar.check()
    if rank == 0:
        print(f"{iters * len(sizes)} all-reduces on 2 ranks sharing one device: {bad} wrong elements, error word clear, "
              f"{dt / (iters * len(sizes)) * 1e6:.1f} us per all-reduce including the fill kernel (host-paced)")
    dist.barrier(); ar.close(); dist.destroy_process_group()

if __name__ == "__main__":
    mp.spawn(worker, args=(2, 29871, 1000), nprocs=2, join=True)
