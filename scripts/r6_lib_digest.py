#!/usr/bin/env python3
"""Digest of what two fine-tune steps leave behind (logits, loss, gradients, parameters), per method, with whatever library is at
pevit_amd/libpevit_hip.so: two builds whose digests agree are bit-identical on these steps.  usage: r6_lib_digest.py [batch]
(scripts/gpu_lib_ab.sh runs it for the stock library and every pevit_amd/variants/*.so)"""
import hashlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
from pevit_amd.engine import HipEngine, adapter_param_spec
from pevit_amd.synth import ARCHS, randomize_adapters, synth_batch, synth_state_dict
B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
arch, C = ARCHS["ViT-B/32"], 100
out = []
for method in ("kadaptation", "adapter"):
    sd = {k: v for k, v in synth_state_dict(arch, seed=2, text_tower=False).items() if k.startswith("visual.")}
    ad = [(n, torch.zeros(s)) for n, s, _ in adapter_param_spec(method, arch.width, arch.layers)]
    randomize_adapters(ad, seed=3); sd.update(dict(ad))
    images, labels = synth_batch(B, arch.resolution, C); images, labels = images.cuda(), labels.cuda()
    e = HipEngine(arch, method, C, B); e.load_state_dict(sd)
    for _ in range(2):
        lg, ls = e.train_step(images, labels, lr=0.01, momentum=0.9, weight_decay=1e-4)
    torch.cuda.synchronize()
    h = hashlib.sha256()
    for t in (lg, ls, e.grads, e.params):
        h.update(t.detach().cpu().numpy().tobytes())
    out.append(f"{method}:{h.hexdigest()[:16]}")
    del e
print("digest B=%d " % B + " ".join(out))
