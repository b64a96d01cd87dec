#!/usr/bin/env python3
"""Where do the HIP engine and the rounding-point emulation part ways?  Transformer seam and visual features, engine vs
emulation vs f32 oracle (relative L2)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import test_gpu_emulation as T
from oracle import emul_bf16, ref_cpu
from pevit_amd.engine import HipEngine
from pevit_amd.synth import synth_batch

def rel(a, b): return float((a.double() - b.double()).norm() / b.double().norm())
for arch_name, method, r, B in [("tiny-128", "lora", 4, 4), ("tiny-128", "kadaptation", 4, 4), ("ViT-B/32-2L", "kadaptation", 4, 16), ("tiny-128", "adapter", 4, 4)]:
    arch, sd = T._case(arch_name, method, r)
    eng = HipEngine(arch, method, 10, B, lora_rank=r)
    eng.load_state_dict(sd)
    g = torch.Generator().manual_seed(9)
    x = torch.randn(arch.tokens, B, arch.width, generator=g)
    with torch.no_grad():
        y_em = emul_bf16.transformer_forward(x, sd, arch.layers, arch.heads, method)
        y_or = ref_cpu.transformer_forward(x, sd, arch.layers, arch.heads, method)
        y1_em = emul_bf16.block(x, sd, 0, arch.heads, method, emul_bf16.make_wcache(sd))
    y_en = eng.transformer_forward(x.cuda(), save=False).cpu()
    print(f"{arch_name} {method}: seam  engine-vs-emul {rel(y_en, y_em):.3e}   oracle-vs-emul {rel(y_or, y_em):.3e}   engine-vs-oracle {rel(y_en, y_or):.3e}")
    images, labels = synth_batch(B, arch.resolution, 10, seed_img=3, seed_lbl=4)
    with torch.no_grad():
        f_em = emul_bf16.visual_forward(images, sd, method)
        f_or = ref_cpu.visual_forward(images, sd, method)
    f_en = eng.visual_forward(images.cuda(), save=False).cpu()
    print(f"     feat  engine-vs-emul {rel(f_en, f_em):.3e}   oracle-vs-emul {rel(f_or, f_em):.3e}   engine-vs-oracle {rel(f_en, f_or):.3e}")
