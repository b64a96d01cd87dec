#!/usr/bin/env python3
"""Forward-only throughput (what validate() and feature extraction run): visual tower + head, no saved activations."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
from pevit_amd.engine import HipEngine
from pevit_amd.synth import ARCHS, reference_init_, synth_batch, synth_state_dict
for arch_name, method, B in (("ViT-B/32", "kadaptation", 128), ("ViT-B/32", "kadaptation", 512), ("ViT-B/16", "compacter", 64), ("ViT-L/14", "kadaptation", 32)):
    arch = ARCHS[arch_name]
    sd = synth_state_dict(arch, seed=2, text_tower=False)
    if method == "compacter":
        sd["visual.transformer.phm_rule"] = torch.rand((4, 4, 4)) * 2 - 1
    eng = HipEngine(arch, method, 100, B); eng.load_state_dict(sd)
    reference_init_(eng.param_views().items(), method)
    images, labels = synth_batch(B, arch.resolution, 100); images = images.cuda()
    for _ in range(5):
        feat = eng.visual_forward(images, save=False); eng.head_forward_backward(feat, None, bn_training=False)
    torch.cuda.synchronize(); n = 30; t0 = time.perf_counter()
    for _ in range(n):
        feat = eng.visual_forward(images, save=False); eng.head_forward_backward(feat, None, bn_training=False)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / n
    print(f"{arch_name} + {method}, batch {B}: forward {dt * 1e3:.3f} ms  {B / dt:,.0f} images/s")
    del eng
