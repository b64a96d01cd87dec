#!/usr/bin/env python3
"""Round 6, VERDICT r5 item 7 (the reference's real workload: batch 64, ~90 short sweep runs on the same frozen backbone).

K independent fine-tune runs -- K engine contexts, each with its own parameters / workspace, each on its OWN stream -- stepped in
lockstep by one host thread.  A batch-64 step fills 120 of the 256 CUs in its large GEMMs (160x128 / 320x256 tiles at M = 3200);
two runs on two streams let the hardware place the second run's workgroups on the CUs the first leaves idle, with no kernel change
and no cross-stream dependency (the expensive thing on this platform is a cross-stream WAIT, not a second stream).

    python scripts/r6_dual_stream.py [--batch 64] [--runs 1,2,3,4] [--steps 60] [--method kadaptation]

Prints one JSON line per K: aggregate images/s over all runs, per-run ms/step; and checks that every concurrent run's parameters
are bit-identical to the same run stepped alone (same kernels, own buffers).
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def make_engine(arch, method, batch, seed, sd, share_from=None):
    from pevit_amd.engine import HipEngine
    from pevit_amd.synth import reference_init_
    eng = HipEngine(arch, method, 100, batch, lora_rank=8 if method == "lora" else 4)
    eng.load_state_dict(sd)
    views = eng.param_views()
    reference_init_(views.items(), method, seed=7 + seed)
    g = torch.Generator().manual_seed(5 + seed)
    with torch.no_grad():
        bound = arch.embed_dim ** -0.5
        views["layers.0.weight"].copy_(((torch.rand(views["layers.0.weight"].shape, generator=g) * 2 - 1) * bound).cuda())
        views["layers.0.bias"].copy_(((torch.rand(views["layers.0.bias"].shape, generator=g) * 2 - 1) * bound).cuda())
    return eng


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--runs", default="1,2,3,4")
    ap.add_argument("--steps", type=int, default=60)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--method", default="kadaptation")
    ap.add_argument("--arch", default="ViT-B/32")
    args = ap.parse_args()
    from pevit_amd.synth import ARCHS, synth_batch, synth_state_dict
    arch = ARCHS[args.arch]
    sd = synth_state_dict(arch, seed=2, text_tower=False)
    if args.method == "compacter":
        sd["visual.transformer.phm_rule"] = torch.rand((4, 4, 4), generator=torch.Generator().manual_seed(4)) * 2 - 1
    kmax = max(int(k) for k in args.runs.split(","))
    engines = [make_engine(arch, args.method, args.batch, r, sd) for r in range(kmax)]
    batches = []
    for r in range(kmax):
        im, lb = synth_batch(args.batch, arch.resolution, 100, seed_img=2 * r, seed_lbl=2 * r + 1)
        batches.append((im.cuda(), lb.cuda()))
    lrs = [0.01 * (r + 1) for r in range(kmax)]               # sweep runs differ in lr / l2
    streams = [torch.cuda.Stream() for _ in range(kmax)]
    init = [e.params.clone() for e in engines]

    def reset():
        for e, p in zip(engines, init):
            e.reset_run()
            e.params.copy_(p)
        torch.cuda.synchronize()

    def run(k, steps):
        for _ in range(steps):
            for r in range(k):
                with torch.cuda.stream(streams[r]):
                    engines[r].train_step(*batches[r], lr=lrs[r], momentum=0.9, weight_decay=1e-6)

    # every run alone, on its own stream: the reference result for the bit-identity check
    solo = []
    for r in range(kmax):
        reset()
        with torch.cuda.stream(streams[r]):
            for _ in range(args.steps + args.warmup):
                engines[r].train_step(*batches[r], lr=lrs[r], momentum=0.9, weight_decay=1e-6)
        torch.cuda.synchronize()
        solo.append(engines[r].params.clone())
    for k in [int(x) for x in args.runs.split(",")]:
        reset()
        run(k, args.warmup)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        run(k, args.steps)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        same = all(torch.equal(engines[r].params, solo[r]) for r in range(k))
        for e in engines[:k]:
            e.check_streamk()
        print(json.dumps({"concurrent_runs": k, "batch_per_run": args.batch, "method": args.method, "arch": args.arch,
                          "aggregate_images_per_sec": k * args.batch * args.steps / dt, "ms_per_lockstep": dt / args.steps * 1e3,
                          "ms_per_run_step": dt / args.steps * 1e3 / k, "bit_identical_to_solo": same}), flush=True)
        assert same, "a concurrent run differs from the same run stepped alone"


if __name__ == "__main__":
    main()
