#!/usr/bin/env python3
"""How far is the HIP step from (a) the f32 oracle and (b) the oracle with bf16-rounded contraction operands
(oracle.ref_cpu.operand_rounding), per fixture: logits (max-rel) and gradients (rel-L2, worst tensor)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
torch.set_num_threads(16)
from conftest import golden_param_dict, load_golden, max_rel, rel_err
from oracle import ref_cpu
from pevit_amd.engine import HipEngine, adapter_param_spec
from pevit_amd.synth import ARCHS, randomize_adapters, synth_batch, synth_state_dict


def oracle(sd, method, classes, images, labels, hw, hb, emulate):
    tr = ref_cpu.OracleTrainer(sd, method, classes)
    with torch.no_grad():
        tr.head_w.copy_(hw); tr.head_b.copy_(hb)
    if emulate:
        with ref_cpu.operand_rounding(torch.bfloat16):
            lg, ls = tr.loss_and_grads(images, labels)
    else:
        lg, ls = tr.loss_and_grads(images, labels)
    g = {n: tr.p[n].grad for n in tr.names if tr.p[n].grad is not None}
    g["layers.0.weight"] = tr.head_w.grad; g["layers.0.bias"] = tr.head_b.grad
    return lg, float(ls), g


def report(tag, eng, sd, method, classes, images, labels, hw, hb):
    logits, loss = eng.forward_backward(images.cuda(), labels.cuda())
    torch.cuda.synchronize()
    gv = {k: v.cpu() for k, v in eng.grad_views().items()}
    lf, lossf, gf = oracle(sd, method, classes, images, labels, hw, hb, False)
    le, losse, ge = oracle(sd, method, classes, images, labels, hw, hb, True)
    def worst(a, b):
        return max(rel_err(a[k], b[k]) for k in b)
    print(f"{tag:28s} logits hip-f32 {max_rel(logits.cpu(), lf):.2e} emu-f32 {max_rel(le, lf):.2e} hip-emu {max_rel(logits.cpu(), le):.2e} | "
          f"loss hip-f32 {abs(float(loss)-lossf):.2e} hip-emu {abs(float(loss)-losse):.2e} | "
          f"grads(worst) hip-f32 {worst(gv, gf):.2e} emu-f32 {worst(ge, gf):.2e} hip-emu {worst(gv, ge):.2e}", flush=True)


for case in ["tiny_kadaptation", "tiny_lora", "tiny_lora_r8", "tiny_adapter", "tiny_compacter"]:
    meta, t = load_golden(case)
    arch = ARCHS[meta["arch"]]
    eng = HipEngine(arch, meta["method"], meta["classes"], meta["batch"], lora_rank=meta["lora_r"])
    sd = golden_param_dict(meta, t)
    eng.load_state_dict(sd)
    v = eng.param_views()
    with torch.no_grad():
        v["layers.0.weight"].copy_(t["head_w"]); v["layers.0.bias"].copy_(t["head_b"])
    report(case, eng, sd, meta["method"], meta["classes"], t["images"], t["labels"], t["head_w"], t["head_b"])

for arch_name, method, r, B in [("ViT-B/32", "kadaptation", 4, 8), ("ViT-B/32", "lora", 8, 8), ("ViT-B/32", "adapter", 4, 8)]:
    arch = ARCHS[arch_name]
    sd = {k: v for k, v in synth_state_dict(arch, seed=2, text_tower=False).items() if k.startswith("visual.")}
    ad = [(n, torch.zeros(s)) for n, s, _ in adapter_param_spec(method, arch.width, arch.layers, r)]
    randomize_adapters(ad, seed=3); sd.update(dict(ad))
    C = 10
    images, labels = synth_batch(B, arch.resolution, C, seed_img=3, seed_lbl=4)
    g = torch.Generator().manual_seed(5); D = arch.embed_dim
    hw = (torch.rand((C, D), generator=g) * 2 - 1) / D ** 0.5; hb = (torch.rand((C,), generator=g) * 2 - 1) / D ** 0.5
    eng = HipEngine(arch, method, C, B, lora_rank=r); eng.load_state_dict(sd)
    v = eng.param_views()
    with torch.no_grad():
        v["layers.0.weight"].copy_(hw); v["layers.0.bias"].copy_(hb)
    report(f"{arch_name} {method} bs{B}", eng, sd, method, C, images, labels, hw, hb)
