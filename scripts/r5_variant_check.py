#!/usr/bin/env python3
"""Is a tuning variant bit-identical to the stock step?  usage: r5_variant_check.py key=value [key=value ...]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
from pevit_amd.engine import HipEngine, adapter_param_spec
from pevit_amd.synth import ARCHS, randomize_adapters, synth_batch, synth_state_dict
arch, method, B, C = ARCHS["ViT-B/32"], "kadaptation", 128, 100
sd = {k: v for k, v in synth_state_dict(arch, seed=2, text_tower=False).items() if k.startswith("visual.")}
ad = [(n, torch.zeros(s)) for n, s, _ in adapter_param_spec(method, arch.width, arch.layers)]
randomize_adapters(ad, seed=3); sd.update(dict(ad))
images, labels = synth_batch(B, arch.resolution, C); images, labels = images.cuda(), labels.cuda()
res = []
for variant in (False, True):
    e = HipEngine(arch, method, C, B); e.load_state_dict(sd)
    for t in sys.argv[1:]:
        k, v = t.split("="); e.tune(k, int(v) if variant else 0)
    for _ in range(2): lg, ls = e.train_step(images, labels, lr=0.01)
    torch.cuda.synchronize()
    res.append((lg.clone(), ls.clone(), e.grads.clone(), e.params.clone()))
    del e
print("variant", sys.argv[1:], "bit-identical:", all(torch.equal(a, b) for a, b in zip(*res)),
      "max |dlogits|", float((res[0][0] - res[1][0]).abs().max()), "max |dgrads|", float((res[0][2] - res[1][2]).abs().max()))
