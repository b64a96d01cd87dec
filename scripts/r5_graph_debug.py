#!/usr/bin/env python3
"""Which tensor differs between an eager step and a graph replay, and from which replay on?"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
from pevit_amd.engine import HipEngine, adapter_param_spec
from pevit_amd.synth import ARCHS, randomize_adapters, synth_batch, synth_state_dict
archn = sys.argv[1] if len(sys.argv) > 1 else "tiny-128"
arch, method, B, C = ARCHS[archn], "kadaptation", (8 if archn.startswith("tiny") else 128), (10 if archn.startswith("tiny") else 100)
sd = {k: v for k, v in synth_state_dict(arch, seed=2, text_tower=False).items() if k.startswith("visual.")}
ad = [(n, torch.zeros(s)) for n, s, _ in adapter_param_spec(method, arch.width, arch.layers)]
randomize_adapters(ad, seed=3); sd.update(dict(ad))
images, labels = synth_batch(B, arch.resolution, C); images, labels = images.cuda(), labels.cuda()
engs = []
for _ in range(3):
    e = HipEngine(arch, method, C, B); e.load_state_dict(sd); engs.append(e)
a, b, c = engs
for t in sys.argv[2:]:
    k, v = t.split("="); [e.tune(k, int(v)) for e in engs]
for e in engs: e.train_step(images, labels, lr=0.05, momentum=0.9, weight_decay=1e-4)
replay = b.capture_train_step(images, labels, lr=0.05, momentum=0.9, weight_decay=1e-4)
names = ("params", "momentum", "grads", "running_mean", "running_var", "_logits", "_loss")
for step in range(5):
    a.train_step(images, labels, lr=0.05, momentum=0.9, weight_decay=1e-4)
    c.train_step(images, labels, lr=0.05, momentum=0.9, weight_decay=1e-4)
    replay(); torch.cuda.synchronize()
    out = []
    for n in names:
        x, y, z = getattr(a, n), getattr(b, n), getattr(c, n)
        out.append(f"{n}: graph {'=' if torch.equal(x, y) else 'DIFF %.2e (%d el)' % (float((x - y).abs().max()), int((x != y).sum()))} eager2 {'=' if torch.equal(x, z) else 'DIFF %.2e' % float((x - z).abs().max())}")
    print(step, " | ".join(out), flush=True)
    if step == 1:
        d = (a.grads != b.grads).nonzero().flatten()
        if d.numel():
            offs = {n: (o, o + v.numel()) for (n, v), o in zip(a.grad_views().items(), [0] * 0)}
            off = 0
            for n, v in a.grad_views().items():
                k = int(((d >= off) & (d < off + v.numel())).sum())
                if k: print("   differing grads in", n, k, "of", v.numel())
                off += v.numel()
