#!/usr/bin/env python3
"""Generate the golden fixtures in tests/golden/ by running the REFERENCE's own
model code (imported, read-only, from /root/reference) on seeded inputs.

Runs only in the build container (the GPU box has no /root/reference); the
outputs (*.npz, *.json) are committed and are *data*: inputs and expected
outputs.  No reference source is copied.

Import recipe: ``vision_benchmark.evaluation.__init__`` pulls in
timm/torchvision/nltk, so the four model files are loaded individually under a
stub package ``refeval`` whose ``__path__`` points at the reference directory.

The harness pieces that cannot be imported (Classifier / train_one need
nltk+vision_datasets at import time) are reproduced with the stock torch
modules the reference itself instantiates: ``BatchNorm1d(D, affine=False)``,
``Linear(D, C)``, ``CrossEntropyLoss``, ``SGD(momentum=0.9)`` with the two
param groups of optim/build.py:81-84.

Usage:  python tests/golden/make_golden.py [--full]
"""
import argparse
import importlib.util
import json
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
REF = "/root/reference/vision_benchmark/evaluation"

from pevit_amd.synth import ARCHS, synth_state_dict, synth_batch, randomize_adapters  # noqa: E402

BUILDERS = {
    "kadaptation": ("model", "build_model"),
    "lora": ("lora_model", "build_lora_model"),
    "adapter": ("adapter_model", "build_adapter_model"),
    "compacter": ("compacter_model", "build_compacter_model"),
}


def load_ref(modname):
    if "refeval" not in sys.modules:
        pkg = types.ModuleType("refeval")
        pkg.__path__ = [REF]
        sys.modules["refeval"] = pkg
    full = f"refeval.{modname}"
    if full in sys.modules:
        return sys.modules[full]
    spec = importlib.util.spec_from_file_location(full, os.path.join(REF, modname + ".py"))
    mod = importlib.util.module_from_spec(spec)
    sys.modules[full] = mod
    spec.loader.exec_module(mod)
    return mod


def trainable_rule(method, name):
    # kadaptation_clip.py:118-122 / lora_clip.py / adapter_tuning_clip.py / compacter_clip.py:122
    if method == "kadaptation":
        return ("adapter" in name) or ("phm_rule" in name) or ("attn.b" in name)
    if method in ("lora", "adapter"):
        return "adapter" in name
    return "compacter" in name


class RefClassifier(torch.nn.Module):
    """Classifier.forward of the harness with the imported backbone."""

    def __init__(self, backbone, dim, classes, head_w, head_b):
        super().__init__()
        self.backbone = backbone
        self.channel_bn = torch.nn.BatchNorm1d(dim, affine=False)
        self.layers = torch.nn.Sequential(torch.nn.Linear(dim, classes))
        self.layers[0].weight.data.copy_(head_w)
        self.layers[0].bias.data.copy_(head_b)

    def forward(self, img):
        feature = self.backbone.encode_image(img).to(img.dtype)
        return self.layers(self.channel_bn(feature))


def build_ref(method, sd, lora_r=4):
    modname, fn = BUILDERS[method]
    mod = load_ref(modname)
    model = getattr(mod, fn)({k: v.clone() for k, v in sd.items()})
    if method == "lora" and lora_r != 4:
        # BASELINE config 3 asks for r=8; the reference hard-codes 4 (lora_model.py:461).
        for blk in model.visual.transformer.resblocks:
            a = blk.attn
            E = a.embed_dim
            a.lora_attn_dim = lora_r
            a.q_proj_adapter1 = torch.nn.Linear(E, lora_r, bias=False)
            a.q_proj_adapter2 = torch.nn.Linear(lora_r, E, bias=False)
            a.v_proj_adapter1 = torch.nn.Linear(E, lora_r, bias=False)
            a.v_proj_adapter2 = torch.nn.Linear(lora_r, E, bias=False)
    return model


def head_init(dim, classes, seed=5):
    g = torch.Generator(device="cpu"); g.manual_seed(seed)
    bound = 1.0 / np.sqrt(dim)
    w = (torch.rand((classes, dim), generator=g) * 2 - 1) * bound
    b = (torch.rand((classes,), generator=g) * 2 - 1) * bound
    return w, b


def run_case(method, arch_name, batch, classes, lora_r=4, steps=3, lr=0.01, wd=1e-4,
             store_tensors=True, reference_init=False):
    arch = ARCHS[arch_name]
    sd = synth_state_dict(arch, seed=2, text_tower=(arch_name.startswith("tiny")))
    if store_tensors:
        sd = {k: (v.half().float() if v.dim() > 0 else v) for k, v in sd.items()}
    model = build_ref(method, sd, lora_r)
    all_names = [n for n, _ in model.named_parameters()]
    for n, p in model.named_parameters():
        p.requires_grad = trainable_rule(method, n)
    train_named = [(n, p) for n, p in model.named_parameters() if p.requires_grad]
    init_vals = {n: p.detach().clone() for n, p in train_named}   # reference init
    if not reference_init:       # (reference_init: the adapters stay exactly as build_model left them, model.py:533-554,987-999)
        randomize_adapters(train_named, seed=3)
    adapters = {n: p.detach().clone() for n, p in train_named}
    # tensors the reference adds but never trains (Compacter's shared phm_rule)
    frozen_extra = {n: p.detach().clone() for n, p in model.named_parameters()
                    if n not in sd and n not in adapters}
    head_w, head_b = head_init(arch.embed_dim, classes)
    clf = RefClassifier(model, arch.embed_dim, classes, head_w, head_b)
    images, labels = synth_batch(batch, arch.resolution, classes)

    # trainable list in Classifier.named_parameters order: backbone.* then layers.0.*
    params = [p for p in clf.parameters() if p.requires_grad]
    opt = torch.optim.SGD([{"params": params}, {"params": [], "weight_decay": 0.0}],
                          lr=lr, momentum=0.9, weight_decay=wd, nesterov=False)
    crit = torch.nn.CrossEntropyLoss()

    out = {}
    # ---- step 0: forward, loss, grads
    opt.zero_grad()
    feat0 = model.encode_image(images).detach()
    logits = clf(images)
    loss = crit(logits, labels)
    loss.backward()
    grads = {}
    for n, p in clf.named_parameters():
        if p.requires_grad:
            grads[n] = None if p.grad is None else p.grad.detach().clone()
    out["feat"] = feat0
    out["logits0"] = logits.detach().clone()
    out["loss0"] = loss.detach().clone()
    opt.step()
    losses = [float(loss)]
    for _ in range(steps - 1):
        opt.zero_grad()
        lg = clf(images)
        ls = crit(lg, labels)
        ls.backward()
        opt.step()
        losses.append(float(ls))
    final = {n: p.detach().clone() for n, p in clf.named_parameters() if p.requires_grad}

    meta = dict(
        method=method, arch=arch_name, batch=batch, classes=classes, lora_r=lora_r,
        steps=steps, lr=lr, wd=wd, losses=losses,
        all_names=all_names,
        trainable_names=[n for n, _ in train_named],
        grad_is_none=[n for n, g in grads.items() if g is None],
        n_adapter_params=int(sum(p.numel() for _, p in train_named)),
        n_trainable_params=int(sum(p.numel() for p in params)),
        n_backbone_params=int(sum(p.numel() for p in model.parameters())),
        n_visual_params=int(sum(p.numel() for p in model.visual.parameters())),
        torch=torch.__version__,
    )
    tensors = {}
    if store_tensors:
        tensors["images"] = images.numpy()
        tensors["labels"] = labels.numpy()
        tensors["head_w"] = head_w.numpy(); tensors["head_b"] = head_b.numpy()
        for n, v in list(adapters.items()) + list(frozen_extra.items()):
            tensors["adapter/" + n] = v.numpy()
        for n, v in init_vals.items():
            tensors["init/" + n] = v.numpy()
        tensors["feat"] = out["feat"].numpy()
        tensors["logits0"] = out["logits0"].numpy()
        tensors["loss0"] = out["loss0"].numpy()
        for n, g in grads.items():
            if g is not None:
                tensors["grad/" + n] = g.numpy()
        for n, v in final.items():
            tensors["final/" + n] = v.numpy()
        tensors["bn_mean"] = clf.channel_bn.running_mean.numpy()
        tensors["bn_var"] = clf.channel_bn.running_var.numpy()
    elif reference_init:
        # full-size at the reference initialisation: the regime every reference run is in (both Kronecker factors
        # start at zero, so only attn.b and the head ever receive a non-zero gradient).  Stored: the head (seeded), the
        # reference's own initial values of what it draws from torch's RNG (the shared phm_rule factors), logits / loss /
        # features of step 0, every NON-ZERO gradient in full, and the trained tensors after `steps` SGD steps.
        meta["sd_checksum"] = {k: [float(v.double().sum()), float((v.double() ** 2).sum())]
                               for k, v in list(sd.items())[:12]}
        tensors["head_w"] = head_w.numpy(); tensors["head_b"] = head_b.numpy()
        for n, v in adapters.items():
            if float(v.abs().max()) != 0.0:
                tensors["adapter/" + n] = v.numpy()
        tensors["logits0"] = out["logits0"].numpy(); tensors["loss0"] = out["loss0"].numpy(); tensors["feat"] = out["feat"].numpy()
        meta["zero_grad_tensors"] = [n for n, g in grads.items() if g is not None and float(g.abs().max()) == 0.0]
        for n, g in grads.items():
            if g is not None and float(g.abs().max()) != 0.0:
                tensors["grad/" + n] = g.numpy()
        for n, v in final.items():
            if n in adapters and torch.equal(v, adapters[n]) and float(v.abs().max()) == 0.0:
                continue                                   # still exactly zero: nothing to store
            tensors["final/" + n] = v.numpy()
        tensors["bn_mean"] = clf.channel_bn.running_mean.numpy()
        tensors["bn_var"] = clf.channel_bn.running_var.numpy()
    else:
        # full-size: summaries only (the state-dict is regenerated from its seed;
        # a checksum guards generator drift)
        meta["sd_checksum"] = {k: [float(v.double().sum()), float((v.double() ** 2).sum())]
                               for k, v in list(sd.items())[:12]}
        tensors["logits0"] = out["logits0"].numpy()
        tensors["loss0"] = out["loss0"].numpy()
        tensors["feat"] = out["feat"].numpy()
        for n, v in frozen_extra.items():
            tensors["adapter/" + n] = v.numpy()
        meta["grad_norms"] = {n: (None if g is None else float(g.double().norm())) for n, g in grads.items()}
        meta["final_norms"] = {n: float(v.double().norm()) for n, v in final.items()}
    return meta, tensors


def param_count_table():
    """Adapter-parameter counts for every (arch, method) at reference init (README.md:84-87)."""
    table = {}
    for arch_name in ("ViT-B/32", "ViT-B/16", "ViT-L/14"):
        arch = ARCHS[arch_name]
        sd = synth_state_dict(arch, seed=2, text_tower=False)
        for method in BUILDERS:
            model = build_ref(method, sd)
            n_adapter = sum(p.numel() for n, p in model.named_parameters() if trainable_rule(method, n))
            n_visual = sum(p.numel() for p in model.visual.parameters())
            table[f"{arch_name}|{method}"] = dict(n_adapter=int(n_adapter), n_visual=int(n_visual))
            del model
    # the full-CLIP counts the survey quotes (B/32 KAdaptation with the real-size text tower)
    sd = synth_state_dict(ARCHS["ViT-B/32"], seed=2, text_tower=True)
    model = build_ref("kadaptation", sd)
    table["ViT-B/32|kadaptation|full"] = dict(
        n_backbone=int(sum(p.numel() for p in model.parameters())),
        n_visual=int(sum(p.numel() for p in model.visual.parameters())))
    return table


def text_case():
    """encode_text of the reference CLIP (model.py:1153-1168) on seeded tokens + the zero-shot head
    reduction of feature.py:513-520 (normalise, mean over templates, normalise, stack on dim 1)."""
    arch = ARCHS["tiny-128"]
    sd = synth_state_dict(arch, seed=2, text_tower=True)
    sd = {k: (v.half().float() if v.dim() > 0 else v) for k, v in sd.items()}
    model = build_ref("kadaptation", sd)
    g = torch.Generator(device="cpu"); g.manual_seed(11)
    classes, templates = 10, 3
    tokens = torch.randint(1, arch.vocab_size - 1, (classes, templates, arch.context_length), generator=g)
    # an end-of-text marker (the arg-max token, model.py:1166) at a random position per prompt
    eot = torch.randint(2, arch.context_length, (classes, templates), generator=g)
    for c in range(classes):
        for t in range(templates):
            tokens[c, t, eot[c, t]] = arch.vocab_size - 1
            tokens[c, t, eot[c, t] + 1:] = 0
    feats, cols = [], []
    with torch.no_grad():
        for c in range(classes):
            e = model.encode_text(tokens[c])
            feats.append(e.clone())
            e = e / e.norm(dim=-1, keepdim=True)
            m = e.mean(dim=0)
            cols.append(m / m.norm())
    return dict(tokens=tokens.numpy(), text_features=torch.stack(feats).numpy(),
                zeroshot_weights=torch.stack(cols, dim=1).numpy())


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--full", action="store_true", help="also run the full-size ViT-B/32 bs=8 cases")
    ap.add_argument("--counts", action="store_true", help="also regenerate the parameter-count table")
    ap.add_argument("--text-only", action="store_true", help="only (re)generate tiny_text.npz")
    ap.add_argument("--refinit", action="store_true",
                    help="only generate full_b32_kadaptation_refinit: ViT-B/32 + KAdaptation at the reference initialisation, bs 8, 3 SGD steps")
    ap.add_argument("--other-archs", action="store_true",
                    help="only generate the full-size summaries for the ViT-B/16 and ViT-L/14 configurations of BASELINE.json")
    args = ap.parse_args()
    if args.text_only:
        np.savez_compressed(os.path.join(HERE, "tiny_text.npz"), **text_case())
        print("tiny_text written")
        return
    if args.refinit:
        torch.manual_seed(0)
        torch.set_num_threads(8)
        meta, tensors = run_case("kadaptation", "ViT-B/32", batch=8, classes=100, steps=3, store_tensors=False, reference_init=True)
        np.savez_compressed(os.path.join(HERE, "full_b32_kadaptation_refinit.npz"), **tensors)
        with open(os.path.join(HERE, "full_b32_kadaptation_refinit.json"), "w") as f:
            json.dump(meta, f, indent=1)
        print("full_b32_kadaptation_refinit ok; losses", meta["losses"], "non-zero grads:", sorted(k for k in tensors if k.startswith("grad/"))[:4], "...")
        return
    if args.other_archs:
        for method, arch_name, tag, lora_r in (("compacter", "ViT-B/16", "full_b16_compacter", 4),
                                                ("kadaptation", "ViT-L/14", "full_l14_kadaptation", 4),
                                                ("lora", "ViT-B/32", "full_b32_lora_r8", 8)):
            meta, tensors = run_case(method, arch_name, batch=8, classes=100, lora_r=lora_r, steps=1, store_tensors=False)
            np.savez_compressed(os.path.join(HERE, f"{tag}.npz"), **tensors)
            with open(os.path.join(HERE, f"{tag}.json"), "w") as f:
                json.dump(meta, f, indent=1)
            print(tag, "ok; loss", meta["losses"])
        return
    torch.manual_seed(0)
    torch.set_num_threads(8)
    # the tiny checkpoint is stored once (fp16-exact values) and shared by all tiny cases
    tiny = synth_state_dict(ARCHS["tiny-128"], seed=2, text_tower=True)
    np.savez_compressed(os.path.join(HERE, "tiny_sd.npz"),
                        **{k: (v.half().numpy() if v.dim() > 0 else v.numpy()) for k, v in tiny.items()})
    for method in BUILDERS:
        meta, tensors = run_case(method, "tiny-128", batch=4, classes=10)
        np.savez_compressed(os.path.join(HERE, f"tiny_{method}.npz"), **tensors)
        with open(os.path.join(HERE, f"tiny_{method}.json"), "w") as f:
            json.dump(meta, f, indent=1)
        print(method, "tiny ok; losses", meta["losses"], "n_adapter", meta["n_adapter_params"])
    meta, tensors = run_case("lora", "tiny-128", batch=4, classes=10, lora_r=8)
    np.savez_compressed(os.path.join(HERE, "tiny_lora_r8.npz"), **tensors)
    with open(os.path.join(HERE, "tiny_lora_r8.json"), "w") as f:
        json.dump(meta, f, indent=1)
    np.savez_compressed(os.path.join(HERE, "tiny_text.npz"), **text_case())
    if args.counts:
        with open(os.path.join(HERE, "param_counts.json"), "w") as f:
            json.dump(param_count_table(), f, indent=1)
        print("param counts written")
    if args.full:
        for method in BUILDERS:
            meta, tensors = run_case(method, "ViT-B/32", batch=8, classes=100, steps=2, store_tensors=False)
            np.savez_compressed(os.path.join(HERE, f"full_b32_{method}.npz"), **tensors)
            with open(os.path.join(HERE, f"full_b32_{method}.json"), "w") as f:
                json.dump(meta, f, indent=1)
            print(method, "full ok; losses", meta["losses"])


if __name__ == "__main__":
    main()
