#!/usr/bin/env python3
"""Deviation of the opt-in fp8 x fp8 forward path (weight_format="fp8-act") from the fp8-weights engine (which is bit-identical to
the bf16 engine on the de-quantised weights), same inputs: one block (seam), and whole steps.  -> markdown on stdout."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import test_gpu_emulation as T
from conftest import max_rel, rel_err
from pevit_amd.engine import HipEngine
from pevit_amd.synth import VitArch, synth_batch

print("| case | what | fp8-act vs fp8 weights |\n|---|---|---|")
for arch_name, B in [("ViT-B/32-2L", 128), ("ViT-B/32", 8), ("ViT-L/14", 8)]:
    arch, sd = T._case(arch_name, "kadaptation", 4)
    images, labels = synth_batch(B, arch.resolution, 10, seed_img=3, seed_lbl=4)
    res = {}
    for wf in ("fp8", "fp8-act"):
        eng = HipEngine(arch, "kadaptation", 10, B, weight_format=wf)
        eng.load_state_dict(sd)
        g = torch.Generator().manual_seed(5)
        v = eng.param_views()
        with torch.no_grad():
            v["layers.0.weight"].copy_(((torch.rand(v["layers.0.weight"].shape, generator=g) * 2 - 1) / arch.embed_dim ** 0.5).cuda())
        x = torch.randn(arch.tokens, B, arch.width, generator=torch.Generator().manual_seed(9)).cuda()
        y = eng.transformer_forward(x, save=False).cpu()
        logits, loss = eng.forward_backward(images.cuda(), labels.cuda())
        torch.cuda.synchronize()
        res[wf] = (y, logits.clone().cpu(), float(loss), {k: t.clone().cpu() for k, t in eng.grad_views().items()})
        del eng
    a, b = res["fp8-act"], res["fp8"]
    errs = sorted(rel_err(a[3][k], b[3][k]) for k in b[3] if float(b[3][k].abs().max()) > 0)
    print(f"| {arch_name} bs {B} | tower output (seam, all {arch.layers} blocks), rel L2 | {rel_err(a[0], b[0]):.2e} |")
    print(f"| | logits, max / max | {max_rel(a[1], b[1]):.2e} |")
    print(f"| | loss, abs ({b[2]:.4f}) | {abs(a[2] - b[2]):.2e} |")
    print(f"| | gradients, rel L2: median / worst | {errs[len(errs)//2]:.2e} / {errs[-1]:.2e} |", flush=True)
