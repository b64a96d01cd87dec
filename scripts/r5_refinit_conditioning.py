#!/usr/bin/env python3
"""How well conditioned are the gradient tensors of a *_refinit fixture?  CPU only (oracle/emul_bf16.py): the distance of the
rounding-point emulation from the RECORDED REFERENCE (tests/golden/*_refinit) with (a) every storage point on, (b) ONE KIND of
storage point on and everything else f32, (c) one point off at a time.  profiles/r05_parity_refinit.md quotes the table.
usage: python scripts/r5_refinit_conditioning.py full_b32_adapter_refinit [--ablate]"""
import sys, json, time
import os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from conftest import load_golden, rel_err, proj_rel_err, max_rel
from oracle import emul_bf16 as E, ref_cpu as R
from pevit_amd.synth import ARCHS, reference_init_, synth_batch, synth_state_dict
torch.set_num_threads(8)
tag = sys.argv[1]
meta, t = load_golden(tag)
method, arch = meta["method"], ARCHS[meta["arch"]]
sd = synth_state_dict(arch, seed=2, text_tower=False)
p = {k: v for k, v in sd.items() if k.startswith("visual.")}
shapes = R.adapter_param_shapes(method, arch.width, arch.layers, meta["lora_r"])
named = [(n, torch.zeros(shapes[n])) for n in meta["trainable_names"]]
if "init_checksum" in meta: reference_init_(named, method, seed=7)
p.update(dict(named))
for k, v in t.items():
    if k.startswith("adapter/"): p[k[8:]] = v.float().view(shapes[k[8:]])
images, labels = synth_batch(meta["batch"], arch.resolution, meta["classes"])
for cls in (E.EmulTrainer,):
    tr = cls(p, method, meta["classes"], lr=meta["lr"], wd=meta["wd"])
    with torch.no_grad(): tr.head_w.copy_(t["head_w"]); tr.head_b.copy_(t["head_b"])
    lg, ls = tr.loss_and_grads(images, labels)
    print(cls.__name__, "logits", max_rel(lg, t["logits0"]), "loss", float(ls), float(t["loss0"]))
    errs = {}
    for n in tr.names:
        g = tr.p[n].grad
        k = "backbone." + n
        if "grad/" + k in t: errs[n] = rel_err(g, t["grad/" + k].view_as(g))
        elif "grad_proj/" + k in t: errs[n] = proj_rel_err(g, meta["proj_index"][k], t["grad_proj/" + k], t["grad_norm/" + k])
    for n, e in sorted(errs.items(), key=lambda kv: -kv[1])[:12]: print("   ", n[-50:], round(e, 4))
    import collections
    by = collections.defaultdict(list)
    for n, e in errs.items(): by[n.split(".")[-2] + "." + n.split(".")[-1]].append(e)
    for k, v in by.items(): print("  kind", k, "max", round(max(v), 4), "mean", round(sum(v) / len(v), 4))
def run_off(off):
    E.POINTS_OFF.clear(); E.POINTS_OFF.update(off)
    tr = E.EmulTrainer(p, method, meta["classes"], lr=meta["lr"], wd=meta["wd"])
    with torch.no_grad(): tr.head_w.copy_(t["head_w"]); tr.head_b.copy_(t["head_b"])
    lg, ls = tr.loss_and_grads(images, labels)
    errs = {}
    for n in tr.names:
        g = tr.p[n].grad; k = "backbone." + n
        if "grad/" + k in t: errs[n] = rel_err(g, t["grad/" + k].view_as(g))
        elif "grad_proj/" + k in t: errs[n] = proj_rel_err(g, meta["proj_index"][k], t["grad_proj/" + k], t["grad_norm/" + k])
    w = max(errs.items(), key=lambda kv: kv[1])
    return max_rel(lg, t["logits0"]), w, sum(errs.values()) / len(errs)

print("---- only ONE kind of rounding on (everything else f32)")
for on in (["w"], ["img"], ["xn"], ["h", "gelu"], []):
    r = run_off([x for x in E.POINTS if x not in on])
    print("on:", on, "logits %.4f worst %.4f (%s) mean %.4f" % (r[0], r[1][1], r[1][0][-40:], r[2]), flush=True)
if "--ablate" in sys.argv:
    print("---- one storage point off at a time")
    for pt in [[]] + [[x] for x in E.POINTS] + [["dyb", "dh", "dx", "ds", "dqkv", "p_bwd", "u", "Q_bwd", "t_bwd"]]:
        r = run_off(pt)
        print(pt, "logits %.4f worst %.4f (%s) mean %.4f" % (r[0], r[1][1], r[1][0][-40:], r[2]), flush=True)
