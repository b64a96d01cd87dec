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

from pevit_amd.synth import ARCHS, synth_state_dict, synth_batch, randomize_adapters, reference_init_  # noqa: E402

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
            # ... with the reference's own initialisation of these four (lora_model.py:466-475): A ~ N(0, 0.02), B = 0
            torch.nn.init.normal_(a.q_proj_adapter1.weight, std=0.02); a.q_proj_adapter2.weight.data.zero_()
            torch.nn.init.normal_(a.v_proj_adapter1.weight, std=0.02); a.v_proj_adapter2.weight.data.zero_()
    return model


def head_init(dim, classes, seed=5):
    g = torch.Generator(device="cpu"); g.manual_seed(seed)
    bound = 1.0 / np.sqrt(dim)
    w = (torch.rand((classes, dim), generator=g) * 2 - 1) * bound
    b = (torch.rand((classes,), generator=g) * 2 - 1) * bound
    return w, b


def sign_projections(v, index, k=32):
    """k seeded +-1 projections of a flattened tensor (float64): <s_j, v>.  For an error e = a - b the mean of
    (<s_j, a> - <s_j, b>)^2 over j is an unbiased estimate of |e|^2, so k numbers per tensor pin its relative L2 error to
    ~ +-sqrt(1/(2k)) and see any permutation / sign / scale error that a norm cannot.  Same generator in the tests
    (tests/conftest.py:sign_projections)."""
    g = torch.Generator(device="cpu"); g.manual_seed(100003 + index)
    s = torch.randint(0, 2, (k, v.numel()), generator=g, dtype=torch.int8).double() * 2 - 1
    return s @ v.double().flatten()


def run_case(method, arch_name, batch, classes, lora_r=4, steps=3, lr=0.01, wd=1e-4,
             store_tensors=True, reference_init=False, redraw=False, full_layers=None, keep_frozen_from=None):
    arch = ARCHS[arch_name]
    sd = synth_state_dict(arch, seed=2, text_tower=(arch_name.startswith("tiny")))
    if store_tensors:
        sd = {k: (v.half().float() if v.dim() > 0 else v) for k, v in sd.items()}
    model = build_ref(method, sd, lora_r)
    if keep_frozen_from and os.path.exists(keep_frozen_from):
        # tensors the reference draws from torch's GLOBAL generator and never trains (Compacter's shared phm_rule ~ U(-1, 1),
        # compacter_model.py:511-519) depend on everything that drew before them in the recording process: a re-recording of an
        # existing fixture keeps the draw it was recorded with, everything else is recomputed from the reference
        old = np.load(keep_frozen_from)
        with torch.no_grad():
            for n, p in model.named_parameters():
                if n not in sd and not trainable_rule(method, n) and "adapter/" + n in old.files:
                    p.copy_(torch.from_numpy(np.asarray(old["adapter/" + n])))
    all_names = [n for n, _ in model.named_parameters()]
    for n, p in model.named_parameters():
        p.requires_grad = trainable_rule(method, n)
    train_named = [(n, p) for n, p in model.named_parameters() if p.requires_grad]
    init_vals = {n: p.detach().clone() for n, p in train_named}   # reference init
    if not reference_init:       # (reference_init: the adapters stay exactly as build_model left them, model.py:533-554,987-999)
        randomize_adapters(train_named, seed=3)
    elif redraw:
        # 1.2 M non-zero initial values (bottleneck Adapter) would make the fixture several MB: the reference's draws are replaced by
        # draws of the SAME law (adapter_model.py:285-295 restated in pevit_amd/synth.py:reference_init_) from a seeded generator,
        # which the test regenerates; the law is checked against what build_*_model left (mean / std / zero pattern) right here
        law = {n: (float(p.detach().mean()), float(p.detach().std()) if p.numel() > 1 else 0.0, float(p.detach().abs().max()) == 0.0) for n, p in train_named}
        reference_init_(train_named, method, seed=7)
        for n, p in train_named:
            m, sd_, z = law[n]
            assert z == (float(p.detach().abs().max()) == 0.0), n
            if not z and p.numel() > 1000:
                assert abs(float(p.detach().std()) - sd_) < 0.05 * sd_ and abs(float(p.detach().mean()) - m) < 0.1 * sd_, (n, m, sd_)
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
    grads_last = {}
    for it in range(steps - 1):
        opt.zero_grad()
        lg = clf(images)
        ls = crit(lg, labels)
        ls.backward()
        if it == steps - 2:              # the gradients of the LAST recorded step (the adapters have moved: every low-rank factor carries signal)
            grads_last = {n: p.grad.detach().clone() for n, p in clf.named_parameters() if p.requires_grad and p.grad is not None}
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
        meta["init_source"] = "pevit_amd.synth.reference_init_(seed=7): same law as the reference's init, checked above" if redraw else "the reference's own draws (adapter/*)"
        meta["full_layers"] = full_layers
        order = {n: i for i, (n, _) in enumerate(clf.named_parameters())}

        def in_full(n):              # tensors stored in full: everything, or (full_layers given) the named blocks + whatever is not per block
            if full_layers is None or ".resblocks." not in n:
                return True
            return int(n.split(".resblocks.")[1].split(".")[0]) in full_layers

        def put(kind, n, v):
            if in_full(n):
                tensors[f"{kind}/{n}"] = v.numpy()
            else:                    # 32 seeded sign projections + the norm (make_golden.sign_projections)
                tensors[f"{kind}_proj/{n}"] = sign_projections(v, order[n]).numpy()
                tensors[f"{kind}_norm/{n}"] = np.float64(float(v.double().norm()))
        if redraw:
            meta["init_checksum"] = {n: [float(v.double().sum()), float((v.double() ** 2).sum())] for n, v in adapters.items()}
            for n in list(tensors):
                if n.startswith("adapter/"):
                    del tensors[n]
        for n, g in grads.items():
            if g is not None and float(g.abs().max()) != 0.0:
                put("grad", n, g)
        meta["zero_grad_last"] = [n for n, g in grads_last.items() if float(g.abs().max()) == 0.0]
        for n, g in grads_last.items():
            if float(g.abs().max()) != 0.0:
                put("grad_last", n, g)
        meta["unchanged"] = []
        for n, v in final.items():
            m = n[len("backbone."):] if n.startswith("backbone.") else n
            base = adapters[m] if m in adapters else (head_w if n.endswith("weight") else head_b)
            if torch.equal(v, base):
                meta["unchanged"].append(n)                # never moved (exact zeros that stay zero, dead parameters): nothing to store
                continue
            # what the steps CHANGED (final - initial): the final value of a tensor that barely moves would only compare its initial value
            put("delta", n, v - base)
        meta["proj_index"] = {n: order[n] for n in order if not in_full(n)}
        for n, v in frozen_extra.items():
            tensors["adapter/" + n] = v.numpy()
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
        # round 5: a norm cannot see a permutation / sign error -- 32 seeded sign projections per gradient tensor can (an unbiased
        # estimate of the relative L2 error, tests/conftest.py:proj_rel_err); index = position in Classifier.named_parameters()
        order = {n: i for i, (n, _) in enumerate(clf.named_parameters())}
        meta["proj_index"] = {n: order[n] for n, g in grads.items() if g is not None}
        for n, g in grads.items():
            if g is not None:
                tensors["grad_proj/" + n] = sign_projections(g, order[n]).numpy()
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
    ap.add_argument("--full-only", action="store_true", help="only the full-size ViT-B/32 bs=8 summaries (full_b32_<method>)")
    ap.add_argument("--counts", action="store_true", help="also regenerate the parameter-count table")
    ap.add_argument("--text-only", action="store_true", help="only (re)generate tiny_text.npz")
    ap.add_argument("--refinit", action="store_true",
                    help="only generate the *_refinit fixtures: full size at the reference initialisation, bs 8, 5 SGD steps "
                         "(ViT-B/32 KAdaptation / LoRA r=8 / Adapter, ViT-B/16 Compacter, ViT-L/14 KAdaptation)")
    ap.add_argument("--only", default="", help="with --refinit: only the fixtures whose name contains this")
    ap.add_argument("--other-archs", action="store_true",
                    help="only generate the full-size summaries for the ViT-B/16 and ViT-L/14 configurations of BASELINE.json")
    args = ap.parse_args()
    if args.text_only:
        np.savez_compressed(os.path.join(HERE, "tiny_text.npz"), **text_case())
        print("tiny_text written")
        return
    if args.refinit:
        # full size, bs 8, at the reference initialisation, 5 SGD steps (gradients of the first AND the last step recorded: by then
        # LoRA's B, the bottleneck up-projection and Compacter's factors have moved, so every low-rank gradient kernel carries signal)
        cases = (("kadaptation", "ViT-B/32", "full_b32_kadaptation_refinit", 4, False, None),
                 ("lora", "ViT-B/32", "full_b32_lora_r8_refinit", 8, True, [0, 5, 11]),
                 ("adapter", "ViT-B/32", "full_b32_adapter_refinit", 4, True, [0, 11]),
                 ("compacter", "ViT-B/16", "full_b16_compacter_refinit", 4, False, None),
                 ("kadaptation", "ViT-L/14", "full_l14_kadaptation_refinit", 4, False, None))
        for method, arch_name, tag, lora_r, redraw, full_layers in cases:
            if args.only and args.only not in tag:
                continue
            torch.manual_seed(0)
            torch.set_num_threads(8)
            meta, tensors = run_case(method, arch_name, batch=8, classes=100, lora_r=lora_r, steps=5, store_tensors=False,
                                     reference_init=True, redraw=redraw, full_layers=full_layers)
            np.savez_compressed(os.path.join(HERE, f"{tag}.npz"), **tensors)
            with open(os.path.join(HERE, f"{tag}.json"), "w") as f:
                json.dump(meta, f, indent=1)
            print(tag, "ok; losses", meta["losses"], "| non-zero grads step 0:", sum(k.startswith(("grad/", "grad_proj/")) for k in tensors),
                  "last step:", sum(k.startswith(("grad_last/", "grad_last_proj/")) for k in tensors), "| moved:", sum(k.startswith(("delta/", "delta_proj/")) for k in tensors),
                  "| %.2f MB" % (os.path.getsize(os.path.join(HERE, f"{tag}.npz")) / 1e6), flush=True)
        return
    if args.full_only:
        torch.manual_seed(0)
        torch.set_num_threads(8)
        for method in BUILDERS:
            meta, tensors = run_case(method, "ViT-B/32", batch=8, classes=100, steps=2, store_tensors=False,
                                     keep_frozen_from=os.path.join(HERE, f"full_b32_{method}.npz"))
            np.savez_compressed(os.path.join(HERE, f"full_b32_{method}.npz"), **tensors)
            with open(os.path.join(HERE, f"full_b32_{method}.json"), "w") as f:
                json.dump(meta, f, indent=1)
            print(method, "full ok; losses", meta["losses"])
        return
    if args.other_archs:
        for method, arch_name, tag, lora_r in (("compacter", "ViT-B/16", "full_b16_compacter", 4),
                                                ("kadaptation", "ViT-L/14", "full_l14_kadaptation", 4),
                                                ("lora", "ViT-B/32", "full_b32_lora_r8", 8)):
            meta, tensors = run_case(method, arch_name, batch=8, classes=100, lora_r=lora_r, steps=1, store_tensors=False,
                                     keep_frozen_from=os.path.join(HERE, f"{tag}.npz"))
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
