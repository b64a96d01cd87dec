#!/usr/bin/env python3
"""Round 6: wall-clock of a weight-decay sweep (hyperparameter_sweep: 7 coarse + 8 refinement runs of train_task, each 10 epochs
of a 5-shot set at batch 64 + validation per epoch -- the reference's real workload, kadaptation_clip.py:188-243) with
TRAIN.SWEEP_CONCURRENCY = 1 (one run at a time, rounds 3-5) and 2 / 3 (runs at a time, each on its own stream and engine context).
Seconds per run = sweep wall-clock / 15; the first sweep at each setting pays for the additional backbones and is not the one
reported.  usage: python scripts/r6_sweep_concurrency.py [--method kadaptation]"""
import argparse, json, os, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import importlib
import torch
from pevit_amd.config import vitb32_clip_config
from pevit_amd.evaluation import _harness
from pevit_amd.evaluation.dataloader import construct_dataloader
from pevit_amd.synth import synth_state_dict

ap = argparse.ArgumentParser()
ap.add_argument("--method", default="kadaptation")
ap.add_argument("--ks", default="1,2,3")
args = ap.parse_args()
mod = importlib.import_module("pevit_amd.evaluation." + {"kadaptation": "kadaptation_clip", "lora": "lora_clip", "adapter": "adapter_tuning_clip",
                                                          "compacter": "compacter_clip"}[args.method])
tmp = tempfile.mkdtemp()
ckpt = os.path.join(tmp, "vitb32_synth.pt")
torch.save(synth_state_dict("ViT-B/32", seed=2, text_tower=True), ckpt)
cfg = vitb32_clip_config()
cfg.MODEL.NAME = ckpt
cfg.DATASET.DATASET, cfg.DATASET.NUM_CLASSES, cfg.DATASET.NUM_SAMPLES_PER_CLASS = "synthetic", 100, 5
cfg.DATASET.SYNTHETIC_SIZES = (1000, 256)          # 5-shot of 100 classes = 500 images -> 400 train / 100 val
cfg.TRAIN.LR, cfg.TRAIN.END_EPOCH = 0.01, 10
cfg.TEST.METRIC = "accuracy"
train, val, test = construct_dataloader(cfg)
for k in [int(x) for x in args.ks.split(",")]:
    cfg.defrost(); cfg.TRAIN.SWEEP_CONCURRENCY = k
    res = []
    for rep in range(3):
        torch.manual_seed(0)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        wd, score = mod.hyperparameter_sweep(train, val, cfg)
        torch.cuda.synchronize(); res.append((time.perf_counter() - t0, wd, score))
    print(json.dumps({"method": args.method, "sweep_concurrency": k, "sweep_seconds": [r[0] for r in res], "seconds_per_run": min(r[0] for r in res[1:]) / 15,
                      "best_wd": [r[1] for r in res], "score": [r[2] for r in res], "deterministic": res[1][1:] == res[2][1:],
                      "idle_backbones_kept": len(_harness._BACKBONES)}), flush=True)
