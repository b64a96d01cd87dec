#!/usr/bin/env python3
"""Which bf16 storage points carry the engine's distance from the f32 reference?  CPU only (oracle/emul_bf16.py).

Case: full ViT-B/32 + KAdaptation, bs 8, random x160 adapters -- the configuration of tests/golden/full_b32_kadaptation, where the
production path measures logits 6.5e-2 / worst gradient 1.1e-1 against the f32 reference (profiles/r03_parity_errors.md).  The
rounding-point emulation reproduces that distance; here every storage point is switched OFF one at a time (its value passes in
f32) and the distance to the f32 oracle is measured again.  A second table does the same at the reference initialisation.

    python scripts/r4_rounding_ablation.py [--refinit] > profiles/r04_rounding_ablation.md"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
from oracle import emul_bf16 as E, ref_cpu as R
from pevit_amd.engine import adapter_param_spec
from pevit_amd.synth import ARCHS, randomize_adapters, synth_batch, synth_state_dict

torch.set_num_threads(8)
arch, method, B, C = ARCHS["ViT-B/32"], "kadaptation", 8, 100
POINTS = [p for p in E.POINTS if p != "bottleneck"]


def case(refinit):
    sd = {k: v for k, v in synth_state_dict(arch, seed=2, text_tower=False).items() if k.startswith("visual.")}
    if refinit:
        sd.update(R.init_adapter_params(method, arch.width, arch.layers))
    else:
        ad = [(n, torch.zeros(s)) for n, s, _ in adapter_param_spec(method, arch.width, arch.layers)]
        randomize_adapters(ad, seed=3); sd.update(dict(ad))
    return sd


def run(cls, sd, images, labels, head):
    tr = cls(sd, method, C)
    with torch.no_grad():
        tr.head_w.copy_(head[0]); tr.head_b.copy_(head[1])
    lg, ls = tr.loss_and_grads(images, labels)
    g = {n: tr.p[n].grad.detach().clone() for n in tr.names if tr.p[n].grad is not None}
    g["layers.0.weight"], g["layers.0.bias"] = tr.head_w.grad.clone(), tr.head_b.grad.clone()
    return lg.detach(), float(ls), g


def dist(a, ref):
    lg = float((a[0] - ref[0]).abs().max() / ref[0].abs().max())
    errs = {n: float((a[2][n] - ref[2][n]).norm() / (ref[2][n].norm() + 1e-30)) for n in ref[2] if float(ref[2][n].norm()) > 0}
    worst = max(errs.items(), key=lambda kv: kv[1])
    return lg, abs(a[1] - ref[1]), worst, sum(errs.values()) / len(errs)


for refinit in ([False, True] if "--both" in sys.argv else [("--refinit" in sys.argv)]):
    sd = case(refinit)
    images, labels = synth_batch(B, arch.resolution, C)
    g = torch.Generator().manual_seed(5)
    head = ((torch.rand((C, arch.embed_dim), generator=g) * 2 - 1) / arch.embed_dim ** 0.5, (torch.rand((C,), generator=g) * 2 - 1) / arch.embed_dim ** 0.5)
    t0 = time.time()
    ref = run(R.OracleTrainer, sd, images, labels, head)
    E.POINTS_OFF.clear()
    base = dist(run(E.EmulTrainer, sd, images, labels, head), ref)
    print(f"## ViT-B/32 + KAdaptation, bs 8, {'reference initialisation' if refinit else 'random x160 adapters (seed 3)'}\n")
    print("Distance of the rounding-point emulation from the f32 oracle with ONE storage point switched off (f32 there), all others on.\n")
    print("| storage point off | logits max-rel | loss abs | worst gradient rel-L2 (tensor) | mean gradient rel-L2 | logits vs all-on |")
    print("|---|---|---|---|---|---|")
    print(f"| (none: every point on) | {base[0]:.3e} | {base[1]:.2e} | {base[2][1]:.3e} ({base[2][0].split('resblocks.')[-1]}) | {base[3]:.3e} | 1.00 |", flush=True)
    rows = []
    for pt in POINTS:
        E.POINTS_OFF.clear(); E.POINTS_OFF.add(pt)
        d = dist(run(E.EmulTrainer, sd, images, labels, head), ref)
        rows.append((pt, d))
        print(f"| {pt} | {d[0]:.3e} | {d[1]:.2e} | {d[2][1]:.3e} ({d[2][0].split('resblocks.')[-1]}) | {d[3]:.3e} | {d[0] / base[0]:.2f} |", flush=True)
    # the three points with the largest effect on the logits, switched off together
    top = [pt for pt, d in sorted(rows, key=lambda r: r[1][0])[:3]]
    E.POINTS_OFF.clear(); E.POINTS_OFF.update(top)
    d = dist(run(E.EmulTrainer, sd, images, labels, head), ref)
    print(f"| {' + '.join(top)} (the three largest, together) | {d[0]:.3e} | {d[1]:.2e} | {d[2][1]:.3e} ({d[2][0].split('resblocks.')[-1]}) | {d[3]:.3e} | {d[0] / base[0]:.2f} |")
    FWD_ACT = ["img", "xn", "qkv", "q_delta", "p", "attn_out", "h", "gelu", "cls"]
    BWD = ["dyb", "dh", "dx", "ds", "dqkv", "p_bwd", "u", "Q_bwd", "t_bwd"]
    for label, grp in (("all nine BACKWARD points off (forward as the engine stores it)", BWD),
                       ("all nine forward ACTIVATION points off (operands: frozen weights and adapter panels bf16)", FWD_ACT),
                       ("frozen weights + adapter panels f32 (w, P, Q_fwd off; activations bf16)", ["w", "P", "Q_fwd"])):
        E.POINTS_OFF.clear(); E.POINTS_OFF.update(grp)
        d = dist(run(E.EmulTrainer, sd, images, labels, head), ref)
        print(f"| {label} | {d[0]:.3e} | {d[1]:.2e} | {d[2][1]:.3e} ({d[2][0].split('resblocks.')[-1]}) | {d[3]:.3e} | {d[0] / base[0]:.2f} |")
    E.POINTS_OFF.clear(); E.POINTS_OFF.update(POINTS)
    d = dist(run(E.EmulTrainer, sd, images, labels, head), ref)
    print(f"| every point off (the emulation in f32: summation order only) | {d[0]:.3e} | {d[1]:.2e} | {d[2][1]:.3e} | {d[3]:.3e} | {d[0] / base[0]:.3f} |")
    E.POINTS_OFF.clear()
    print(f"\n({time.time() - t0:.0f} s on {torch.get_num_threads()} CPU threads)\n", flush=True)
