#!/usr/bin/env python3
"""Measured errors of the bf16 production path against the reference fixtures (tests/test_gpu_tower.py cases), so that the
gates there can be set to 2x what is measured."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from conftest import load_golden, max_rel, rel_err
import test_gpu_tower as TT
for case in ["tiny_kadaptation", "tiny_lora", "tiny_lora_r8", "tiny_adapter", "tiny_compacter"]:
    meta, t = load_golden(case)
    eng, sd = TT.make_engine(meta, t)
    images, labels = t["images"].cuda(), t["labels"].cuda()
    logits, loss = eng.forward_backward(images, labels, bn_training=True)
    torch.cuda.synchronize()
    none = {n[len("backbone."):] for n in meta["grad_is_none"]}
    errs = {}
    for name, g in eng.grad_views().items():
        key = "grad/" + (name if name.startswith("layers.") else "backbone." + name)
        if name not in none:
            errs[name] = rel_err(g.cpu(), t[key])
    w = max(errs.items(), key=lambda kv: kv[1])
    print(f"{case}: step logits {max_rel(logits.cpu(), t['logits0']):.2e} loss {abs(float(loss) - float(t['loss0'])):.2e} worst grad {w[1]:.2e} ({w[0][-40:]}) median {sorted(errs.values())[len(errs)//2]:.2e}")
    if case == "tiny_lora_r8":
        continue
    eng, sd = TT.make_engine(meta, t)
    losses = []
    for _ in range(meta["steps"]):
        _, loss = eng.train_step(images, labels, lr=meta["lr"], momentum=0.9, weight_decay=meta["wd"])
        losses.append(float(loss))
    dl = max(abs(a - b) for a, b in zip(losses, meta["losses"]))
    perr = {}
    for name, p in eng.param_views().items():
        key = "final/" + (name if name.startswith("layers.") else "backbone." + name)
        if name not in none:
            perr[name] = rel_err(p.cpu(), t[key])
    w = max(perr.items(), key=lambda kv: kv[1])
    print(f"    trajectory: max |loss diff| {dl:.2e}  worst final param {w[1]:.2e} ({w[0][-40:]})  bn mean {rel_err(eng.running_mean.cpu(), t['bn_mean']):.2e} var {rel_err(eng.running_var.cpu(), t['bn_var']):.2e}")
