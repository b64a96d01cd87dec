#!/usr/bin/env python3
"""Where the wall-clock of one sweep run (train_task with backbone reuse) goes: cProfile of the third run + GPU-busy time."""
import cProfile, os, pstats, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
from pevit_amd.config import vitb32_clip_config
from pevit_amd.evaluation import kadaptation_clip as mod, _harness
from pevit_amd.evaluation.dataloader import construct_dataloader
from pevit_amd.synth import synth_state_dict
tmp = tempfile.mkdtemp(); ckpt = os.path.join(tmp, "vitb32_synth.pt")
torch.save(synth_state_dict("ViT-B/32", seed=2, text_tower=True), ckpt)
cfg = vitb32_clip_config(); cfg.MODEL.NAME = ckpt
cfg.DATASET.DATASET, cfg.DATASET.NUM_CLASSES, cfg.DATASET.NUM_SAMPLES_PER_CLASS = "synthetic", 100, 5
cfg.DATASET.SYNTHETIC_SIZES = (1000, 256)
cfg.TRAIN.LR, cfg.TRAIN.WD, cfg.TRAIN.END_EPOCH = 0.01, 1e-6, 10
cfg.TEST.METRIC = "accuracy"
train, val, test = construct_dataloader(cfg)
for _ in range(2):
    mod.train_task(train, val, cfg, sweep_run=True)
torch.cuda.synchronize()
pr = cProfile.Profile()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
t0 = time.perf_counter(); e0.record(); pr.enable()
mod.train_task(train, val, cfg, sweep_run=True)
pr.disable(); e1.record(); torch.cuda.synchronize(); dt = time.perf_counter() - t0
print(f"run: {dt:.3f} s wall, {e0.elapsed_time(e1) / 1e3:.3f} s between stream events")
pstats.Stats(pr).sort_stats("cumulative").print_stats(38)
