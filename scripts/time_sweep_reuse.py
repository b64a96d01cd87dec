#!/usr/bin/env python3
"""Wall-clock of consecutive train_task() runs (what a hyper-parameter sweep does ~90 times per dataset):
first run builds the model from the checkpoint file, later runs re-use the constructed backbone."""
import os, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
from pevit_amd.config import vitb32_clip_config
from pevit_amd.evaluation import kadaptation_clip as mod, _harness
from pevit_amd.evaluation.dataloader import construct_dataloader
from pevit_amd.synth import synth_state_dict

tmp = tempfile.mkdtemp()
ckpt = os.path.join(tmp, "vitb32_synth.pt")
torch.save(synth_state_dict("ViT-B/32", seed=2, text_tower=True), ckpt)
cfg = vitb32_clip_config()
cfg.MODEL.NAME = ckpt
cfg.DATASET.DATASET, cfg.DATASET.NUM_CLASSES, cfg.DATASET.NUM_SAMPLES_PER_CLASS = "synthetic", 100, 5
cfg.DATASET.SYNTHETIC_SIZES = (1000, 256)          # 5-shot of 100 classes = 500 images -> 400 train / 100 val
cfg.TRAIN.LR, cfg.TRAIN.WD, cfg.TRAIN.END_EPOCH = 0.01, 1e-6, 10
cfg.TEST.METRIC = "accuracy"
train, val, test = construct_dataloader(cfg)
for reuse in (True, False):
    _harness._BACKBONES.clear()
    times = []
    for run in range(4):
        if not reuse:
            _harness._BACKBONES.clear()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        mod.train_task(train, val, cfg, sweep_run=True)
        torch.cuda.synchronize(); times.append(time.perf_counter() - t0)
    print(("with" if reuse else "without") + " backbone reuse: " + ", ".join(f"{t:.2f} s" for t in times) +
          f"   ({len(train.dataset)} train images, bs 64, {cfg.TRAIN.END_EPOCH} epochs + validation each epoch)")
