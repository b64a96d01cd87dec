#!/usr/bin/env python3
"""Diagnostic (GPU box): error of the HIP step vs the f32 oracle, next to the error that bf16
operand rounding alone produces in the oracle, on the parity cases.  Calibrates test tolerances."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from conftest import golden_param_dict, load_golden, max_rel, rel_err
from oracle import ref_cpu
from pevit_amd.engine import HipEngine, adapter_param_spec
from pevit_amd.synth import ARCHS, randomize_adapters, synth_batch, synth_state_dict


def oracle_run(p, method, classes, images, labels, head, emul):
    tr = ref_cpu.OracleTrainer(p, method, classes, lr=0.01, wd=1e-4)
    with torch.no_grad():
        tr.head_w.copy_(head[0]); tr.head_b.copy_(head[1])
    if emul:
        with ref_cpu.operand_rounding(torch.bfloat16):
            logits, loss = tr.loss_and_grads(images, labels)
    else:
        logits, loss = tr.loss_and_grads(images, labels)
    return tr, logits, loss


def case(name, arch_name, method, p, classes, images, labels, head, lora_r=4):
    arch = ARCHS[arch_name]
    t0, l0, loss0 = oracle_run(p, method, classes, images, labels, head, False)
    t1, l1, loss1 = oracle_run(p, method, classes, images, labels, head, True)
    eng = HipEngine(arch, method, classes, images.shape[0], lora_rank=lora_r)
    eng.load_state_dict(p)
    v = eng.param_views()
    with torch.no_grad():
        v["layers.0.weight"].copy_(head[0]); v["layers.0.bias"].copy_(head[1])
    lg, ls = eng.forward_backward(images.cuda(), labels.cuda())
    torch.cuda.synchronize()
    lg = lg.cpu()
    print(f"== {name}: loss f32 {float(loss0):.5f} emul {float(loss1):.5f} hip {float(ls):.5f}")
    print(f"   logits max-rel: hip-vs-f32 {max_rel(lg, l0):.4f}  emul-vs-f32 {max_rel(l1, l0):.4f}  hip-vs-emul {max_rel(lg, l1):.4f}")
    gv = eng.grad_views()
    worst = [0, 0, 0]; rows = []
    for n in t0.names:
        if t0.p[n].grad is None:
            continue
        a = rel_err(gv[n].cpu(), t0.p[n].grad); b = rel_err(t1.p[n].grad, t0.p[n].grad); c = rel_err(gv[n].cpu(), t1.p[n].grad)
        rows.append((a, b, c, n))
        worst = [max(worst[0], a), max(worst[1], b), max(worst[2], c)]
    print(f"   grads worst rel-L2: hip-vs-f32 {worst[0]:.4f}  emul-vs-f32 {worst[1]:.4f}  hip-vs-emul {worst[2]:.4f}")
    rows.sort(reverse=True)
    for a, b, c, n in rows[:4]:
        print(f"      {n}: {a:.4f} {b:.4f} {c:.4f}")
    a = rel_err(gv["layers.0.weight"].cpu(), t0.head_w.grad); b = rel_err(t1.head_w.grad, t0.head_w.grad)
    print(f"   head.weight grad: hip-vs-f32 {a:.4f} emul-vs-f32 {b:.4f}")
    del eng


def long_cases():
    for method, arch_name, B in (("kadaptation", "tiny-n197", 8), ("lora", "tiny-n257", 8), ("compacter", "tiny-n197", 9),
                                 ("adapter", "tiny-n257", 8)):
        arch = ARCHS[arch_name]
        sd = {k: v for k, v in synth_state_dict(arch, seed=21, text_tower=False).items() if k.startswith("visual.")}
        ad = [(n, torch.zeros(s)) for n, s, tr in adapter_param_spec(method, arch.width, arch.layers)]
        randomize_adapters(ad, seed=6)
        for n, v in ad:
            if n.endswith("phm_rule"):
                v.copy_(torch.rand(v.shape, generator=torch.Generator().manual_seed(8)) * 2 - 1)
        sd.update(dict(ad))
        images, labels = synth_batch(B, arch.resolution, 10, seed_img=3, seed_lbl=4)
        g = torch.Generator().manual_seed(5)
        D = arch.embed_dim
        head = ((torch.rand((10, D), generator=g) * 2 - 1) / D ** 0.5, (torch.rand((10,), generator=g) * 2 - 1) / D ** 0.5)
        case(f"{method} {arch_name} B={B}", arch_name, method, sd, 10, images, labels, head)


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "long":
        return long_cases()
    for c in ("tiny_kadaptation", "tiny_lora_r8", "tiny_lora"):
        meta, t = load_golden(c)
        p = golden_param_dict(meta, t)
        case(c, meta["arch"], meta["method"], p, meta["classes"], t["images"], t["labels"], (t["head_w"], t["head_b"]), meta["lora_r"])
    for B in (8, 32):
        arch = ARCHS["ViT-B/32"]
        sd = {k: v for k, v in synth_state_dict(arch, seed=2, text_tower=False).items() if k.startswith("visual.")}
        ad = [(n, torch.zeros(s)) for n, s, _ in adapter_param_spec("kadaptation", 768, 12)]
        randomize_adapters(ad, seed=3); sd.update(dict(ad))
        images, labels = synth_batch(B, 224, 100)
        g = torch.Generator().manual_seed(5)
        head = ((torch.rand((100, 512), generator=g) * 2 - 1) / 512 ** 0.5, (torch.rand((100,), generator=g) * 2 - 1) / 512 ** 0.5)
        case(f"full ViT-B/32 kadaptation bs={B}", "ViT-B/32", "kadaptation", sd, 100, images, labels, head)


if __name__ == "__main__":
    main()
