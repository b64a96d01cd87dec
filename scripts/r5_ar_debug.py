#!/usr/bin/env python3
"""pevit_allreduce_flat with W processes on one device, R rounds of mixed sizes: which contribution is wrong when a sum is wrong?
usage: python scripts/r5_ar_debug.py W [rounds]   (env PEVIT_AR_COARSE=1: plain allocation)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch, torch.distributed as dist, torch.multiprocessing as mp


def contrib(it, r, n):
    return torch.randn(n, generator=torch.Generator().manual_seed(100 * it + r))


def worker(rank, world, port, rounds):
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from pevit_amd import dp
    torch.cuda.set_device(0)
    ar = dp.FlatAllReduce(101476)
    if rank == 0:
        print("fine-grained mailbox:", ar.fine_grained, flush=True)
    sizes = (101476, 1, 7, 4097, 101475, 64, 101476, 33)
    bad = 0
    for it in range(rounds):
        n = sizes[it % len(sizes)]
        buf = contrib(it, rank, n).cuda()
        ar.all_reduce(buf)
        torch.cuda.synchronize()
        got = buf.cpu()
        want = torch.zeros(n)
        for r in range(world):
            want = want + contrib(it, r, n)
        if not torch.equal(got, want):
            bad += 1
            # which peers' contributions are stale: compare with the sum that uses round it-2's (same parity) values for a subset
            d = got - want
            who = []
            for r in range(world):
                if r == rank: continue
                for back in (2, 4, 6):
                    if it - back >= 0:
                        old = contrib(it - back, r, sizes[(it - back) % len(sizes)])
                        m = min(n, old.numel())
                        alt = d[:m] - (old[:m] - contrib(it, r, n)[:m])
                        if float(alt.abs().max()) < float(d[:m].abs().max()) * 0.5: who.append((r, back))
            print(f"rank {rank} round {it} n {n}: max diff {float(d.abs().max()):.3f}, elements wrong {int((d != 0).sum())}/{n}, stale-looking (peer, rounds back): {who}", flush=True)
    try:
        ar.check()
    except Exception as e:
        print(f"rank {rank}: {e}", flush=True)
    print(f"rank {rank}: {bad} wrong of {rounds}", flush=True)
    dist.barrier(); ar.close(); dist.destroy_process_group()


if __name__ == "__main__":
    W = int(sys.argv[1]); R = int(sys.argv[2]) if len(sys.argv) > 2 else 24
    mp.spawn(worker, args=(W, 29500 + os.getpid() % 400, R), nprocs=W, join=True)
