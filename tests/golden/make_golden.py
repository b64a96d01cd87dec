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
import contextlib
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


# ---- the bf16 floor, measured on the REFERENCE itself (round 6) --------------------------------------------------------------
# BASELINE configs 2-4 prescribe bf16 frozen weights.  How far that alone moves the reference's own outputs is recorded next to
# every *_refinit fixture (``floor`` in the .json) by running the imported reference a second and a third time from the same
# initial state:
#   leg "weights":  resblocks.*.{attn.in_proj_weight, attn.out_proj.weight, mlp.c_fc.weight, mlp.c_proj.weight}, conv1.weight and
#                   proj rounded to bf16 (stored back as f32), everything else and every operation in f32;
#   leg "operands": the same, plus both operands of every CONTRACTION the reference evaluates (linear, matmul, bmm, attention)
#                   rounded to bf16 on the way in (value rounded, gradient passed through), the gradient arriving at its output
#                   rounded to bf16 (what a dX product on the matrix core reads), and the pixels rounded in front of conv1 --
#                   f32 accumulation, f32 everything else (class bf16_operands).
# Neither leg involves a line of the engine or of oracle/: it is the reference's arithmetic with bf16 operands.
FROZEN_BF16 = (".attn.in_proj_weight", ".attn.out_proj.weight", ".mlp.c_fc.weight", ".mlp.c_proj.weight")


def round_frozen_weights_(model, fp8=False):
    """bf16 frozen weights; fp8: the four block weights as e4m3 codes x one power-of-two scale per output channel instead
    (BASELINE config 5; the format is stated in pevit_amd/fp8.py -- every such value is exact in bf16), stem weights bf16."""
    from pevit_amd import fp8 as fmt
    n = 0
    with torch.no_grad():
        for name, p in model.named_parameters():
            if not name.startswith("visual."):
                continue
            if name.endswith(FROZEN_BF16) and ".resblocks." in name:
                p.copy_(fmt.dequantize_rows(*fmt.quantize_rows(p)) if fp8 else p.bfloat16().float()); n += 1
            elif name in ("visual.conv1.weight", "visual.proj"):
                p.copy_(p.bfloat16().float()); n += 1
    return n


class _RoundValue(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        return x.bfloat16().float()

    @staticmethod
    def backward(ctx, g):
        return g


class _RoundGradient(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        return x.clone()          # (not a view: compacter_model.py:307 adds its bias in place)

    @staticmethod
    def backward(ctx, g):
        return g.bfloat16().float()


class bf16_operands:
    """Context: every CONTRACTION the reference evaluates sees bf16 operands (value rounded on the way in, gradient passed
    through) and the gradient arriving at its output is rounded to bf16 -- what any engine that feeds bf16 operands to a matrix
    core with f32 accumulation does, whatever its kernels look like:
      * ``linear``: the reference's own alias (model.py:256 / lora_model.py:256) and torch.nn.functional.linear (nn.Linear, the
        stock nn.MultiheadAttention of adapter_model.py:314 / compacter_model.py:481);
      * ``torch.matmul`` (x @ H of model.py:584, compacter_model.py:306; LoRA's two products, lora_model.py:492,514), the ``@``
        operator (the output projection ``x @ self.proj``, model.py:1049: the gradient entering the tower is rounded there) and
        ``torch.bmm`` / ``torch.baddbmm`` with an inner dimension > 1 (q k^T and p v, model.py:804-812; the rank-1 outer products
        of :567-579 are element-wise products and stay f32);
      * ``scaled_dot_product_attention`` (what the stock attention calls with need_weights=False): restated as
        softmax(q k^T / sqrt(d)) v with the two products rounded as above;
      * the pixels in front of conv1.
    Everything else -- softmax, LayerNorm, GELU, residual adds, the Kronecker expansions, the loss, SGD -- stays f32."""

    def __init__(self, model):
        self.model = model

    def __enter__(self):
        import math
        import torch.nn.functional as F
        raw_linear, raw_matmul, raw_bmm, raw_baddbmm = torch._C._nn.linear, torch.matmul, torch.bmm, torch.baddbmm
        RV, RG = _RoundValue.apply, _RoundGradient.apply

        def lin(x, w, b=None):
            return RG(raw_linear(RV(x), RV(w), b))

        def matmul(input, other, **kw):
            return RG(raw_matmul(RV(input), RV(other), **kw))

        def bmm(a, b, **kw):
            if a.shape[-1] == 1:
                return raw_bmm(a, b, **kw)
            return RG(raw_bmm(RV(a), RV(b), **kw))

        def baddbmm(m, a, b, **kw):
            return RG(raw_baddbmm(m, RV(a), RV(b), **kw))

        def sdpa(q, k, v, attn_mask=None, dropout_p=0.0, is_causal=False, scale=None, **kw):
            assert dropout_p == 0.0 and not is_causal
            s = RG(raw_matmul(RV(q), RV(k).transpose(-2, -1))) * (scale if scale is not None else 1.0 / math.sqrt(q.shape[-1]))
            if attn_mask is not None:
                s = s + attn_mask
            return RG(raw_matmul(RV(torch.softmax(s, dim=-1)), RV(v)))
        self.saved = [(F, "linear", F.linear), (F, "scaled_dot_product_attention", F.scaled_dot_product_attention),
                      (torch, "matmul", torch.matmul), (torch, "bmm", torch.bmm), (torch, "baddbmm", torch.baddbmm),
                      (torch.Tensor, "__matmul__", torch.Tensor.__matmul__)]
        for m in ("refeval.model", "refeval.lora_model"):
            if m in sys.modules:
                self.saved.append((sys.modules[m], "linear", sys.modules[m].linear))
        new = {"linear": lin, "scaled_dot_product_attention": sdpa, "matmul": matmul, "bmm": bmm, "baddbmm": baddbmm,
               "__matmul__": lambda a, b: matmul(a, b)}
        for obj, attr, _ in self.saved:
            setattr(obj, attr, new[attr])
        self.hook = self.model.visual.conv1.register_forward_pre_hook(lambda mod, a: (a[0].bfloat16().float(),))
        return self

    def __exit__(self, *exc):
        for obj, attr, old in self.saved:
            setattr(obj, attr, old)
        self.hook.remove()
        return False


def run_case(method, arch_name, batch, classes, lora_r=4, steps=3, lr=0.01, wd=1e-4,
             store_tensors=True, reference_init=False, redraw=False, full_layers=None, keep_frozen_from=None, bf16_leg=None, raw=None):
    arch = ARCHS[arch_name]
    sd = synth_state_dict(arch, seed=2, text_tower=(arch_name.startswith("tiny")))
    if store_tensors:
        sd = {k: (v.half().float() if v.dim() > 0 else v) for k, v in sd.items()}
    model = build_ref(method, sd, lora_r)
    if bf16_leg:                 # "weights" | "operands" | "fp8" (= operands with e4m3 block weights): the floor legs above
        assert round_frozen_weights_(model, fp8=(bf16_leg == "fp8")) == 4 * arch.layers + 2
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
    stack = contextlib.ExitStack()
    if bf16_leg in ("operands", "fp8"):
        stack.enter_context(bf16_operands(model))
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
    stack.close()
    if raw is not None:          # the untruncated tensors of this run (floor_case compares two runs of the reference with them)
        raw.update(logits0=out["logits0"], loss0=float(out["loss0"]), losses=list(losses),
                   grad={n: g for n, g in grads.items() if g is not None}, grad_last=grads_last,
                   delta={n: v - (adapters[n[len("backbone."):]] if n.startswith("backbone.") else (head_w if n.endswith("weight") else head_b))
                          for n, v in final.items()})

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


REFINIT_CASES = (("kadaptation", "ViT-B/32", "full_b32_kadaptation_refinit", 4, False, None),
                 ("lora", "ViT-B/32", "full_b32_lora_r8_refinit", 8, True, [0, 5, 11]),
                 ("adapter", "ViT-B/32", "full_b32_adapter_refinit", 4, True, [0, 11]),
                 ("compacter", "ViT-B/16", "full_b16_compacter_refinit", 4, False, None),
                 ("kadaptation", "ViT-L/14", "full_l14_kadaptation_refinit", 4, False, None))


def floor_case(method, arch_name, tag, lora_r, redraw, full_layers, legs=("weights", "operands"), run_kwargs=None):
    """The bf16 floor of one *_refinit fixture: three runs of the imported reference from the same initial state (f32 as recorded,
    bf16 frozen weights, bf16 frozen weights + bf16 linear operands), per-tensor deviation of the two bf16 legs from the f32 run.
    Written into the fixture's .json under "floor"; the .npz is not touched (and the f32 run must reproduce its loss trajectory
    bit for bit, or nothing is written)."""
    path = os.path.join(HERE, f"{tag}.json")
    with open(path) as f:
        meta = json.load(f)
    runs = {}
    for leg in (None,) + tuple(legs):
        torch.manual_seed(0)
        torch.set_num_threads(8)
        raw = {}
        kw = run_kwargs or dict(batch=8, classes=100, steps=5, store_tensors=False, reference_init=True, redraw=redraw, full_layers=full_layers)
        run_case(method, arch_name, lora_r=lora_r, bf16_leg=leg, raw=raw, **kw)
        runs[leg] = raw
        print(tag, leg or "f32", "losses", raw["losses"], flush=True)
    base = runs[None]
    assert base["losses"] == meta["losses"], (base["losses"], meta["losses"])      # this IS the recorded run

    def rel(a, b):
        a = a.double().flatten(); b = b.double().flatten()
        return float((a - b).norm() / (b.norm() + 1e-30))
    floor = dict(meta.get("floor", {}))
    floor.update({"recipe": "tests/golden/make_golden.py --refinit --bf16-weights: the imported reference run on the fixture's initial "
                       "state with (weights) the frozen block weights, conv1.weight and proj rounded to bf16, (operands) "
                       "additionally both operands of every contraction (linear, matmul, bmm, attention) and the gradient arriving at its "
                       "output rounded to bf16; "
                       "per-tensor relative L2 deviation from the reference's own f32 run (logits: max abs / max abs of the f32 "
                       "run; losses: abs).  (fp8, ViT-L/14 only: the operands leg with the four block weights as e4m3 codes x "
                       "power-of-two channel scales, pevit_amd/fp8.py)"})
    for leg in legs:
        r = runs[leg]
        d = {"logits": float((r["logits0"].double() - base["logits0"].double()).abs().max() / base["logits0"].double().abs().max()),
             "loss0": abs(r["loss0"] - base["loss0"]),
             "loss_traj": [abs(a - b) for a, b in zip(r["losses"], base["losses"])]}
        for kind in ("grad", "grad_last", "delta"):
            d[kind] = {n: rel(r[kind][n], v) for n, v in base[kind].items() if float(v.abs().max()) != 0.0}
            # one number for the whole step: all of its tensors as one vector
            num = sum(float((r[kind][n].double() - v.double()).pow(2).sum()) for n, v in base[kind].items())
            den = sum(float(v.double().pow(2).sum()) for v in base[kind].values())
            d[kind + "_all"] = (num / (den + 1e-300)) ** 0.5
        floor[leg] = d
    floor["contractions"] = "linear, matmul, @, bmm, baddbmm, scaled_dot_product_attention"      # what the operands leg rounds (round 6, final)
    meta["floor"] = floor
    with open(path, "w") as f:
        json.dump(meta, f, indent=1)
    for leg in legs:
        d = floor[leg]
        print(tag, "floor", leg, "logits %.3g loss0 %.3g traj %.3g" % (d["logits"], d["loss0"], max(d["loss_traj"])),
              "| worst grad %.3g grad_last %.3g delta %.3g" % tuple(max(d[k].values(), default=0.0) for k in ("grad", "grad_last", "delta")),
              "| whole-step grad %.3g grad_last %.3g delta %.3g" % tuple(d[k + "_all"] for k in ("grad", "grad_last", "delta")), flush=True)


RANDOM_ADAPTER_FIXTURES = ("tiny_kadaptation", "tiny_lora", "tiny_lora_r8", "tiny_adapter", "tiny_compacter",
                           "full_b32_kadaptation", "full_b32_lora", "full_b32_adapter", "full_b32_compacter",
                           "full_b32_lora_r8", "full_b16_compacter", "full_l14_kadaptation")


def floor_random_adapter_fixture(tag):
    """The same floor for a random-adapter fixture (tiny_*, full_*): the run is rebuilt from the fixture's own .json (method,
    arch, batch, classes, steps, lr, wd) exactly as main() recorded it -- the f32 leg must reproduce the recorded loss trajectory."""
    with open(os.path.join(HERE, f"{tag}.json")) as f:
        meta = json.load(f)
    tiny = tag.startswith("tiny")
    kw = dict(batch=meta["batch"], classes=meta["classes"], steps=meta["steps"], lr=meta["lr"], wd=meta["wd"], store_tensors=tiny,
              keep_frozen_from=os.path.join(HERE, f"{tag}.npz"))
    floor_case(meta["method"], meta["arch"], tag, meta["lora_r"], False, None, run_kwargs=kw)


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


def tiny_lora_r8():
    """tiny_lora_r8: the global generator is re-seeded right here (the r = 8 swap of build_ref draws its N(0, 0.02) `init/*` values
    from it), so that this fixture regenerates bit for bit on its own (--tiny-lora-r8) as well as inside the full run."""
    torch.manual_seed(0)
    meta, tensors = run_case("lora", "tiny-128", batch=4, classes=10, lora_r=8)
    np.savez_compressed(os.path.join(HERE, "tiny_lora_r8.npz"), **tensors)
    with open(os.path.join(HERE, "tiny_lora_r8.json"), "w") as f:
        json.dump(meta, f, indent=1)
    print("tiny_lora_r8 ok; losses", meta["losses"])


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--full", action="store_true", help="also run the full-size ViT-B/32 bs=8 cases")
    ap.add_argument("--full-only", action="store_true", help="only the full-size ViT-B/32 bs=8 summaries (full_b32_<method>)")
    ap.add_argument("--counts", action="store_true", help="also regenerate the parameter-count table")
    ap.add_argument("--text-only", action="store_true", help="only (re)generate tiny_text.npz")
    ap.add_argument("--refinit", action="store_true",
                    help="only generate the *_refinit fixtures: full size at the reference initialisation, bs 8, 5 SGD steps "
                         "(ViT-B/32 KAdaptation / LoRA r=8 / Adapter, ViT-B/16 Compacter, ViT-L/14 KAdaptation)")
    ap.add_argument("--bf16-weights", action="store_true",
                    help="with --refinit: do not re-record; run the reference again with bf16 frozen weights (and bf16 linear "
                         "operands) and store the per-tensor deviation from its own f32 run as `floor` in each *_refinit.json")
    ap.add_argument("--floor-random", action="store_true",
                    help="store the same reference-recorded bf16 floor in the random-adapter fixtures (tiny_*, full_*; --only filters)")
    ap.add_argument("--legs", default="", help="with --bf16-weights: comma-separated subset of weights,operands,fp8 (merged into the stored floor)")
    ap.add_argument("--only", default="", help="with --refinit: only the fixtures whose name contains this")
    ap.add_argument("--other-archs", action="store_true",
                    help="only generate the full-size summaries for the ViT-B/16 and ViT-L/14 configurations of BASELINE.json")
    ap.add_argument("--tiny-lora-r8", action="store_true", help="only (re)generate tiny_lora_r8.{npz,json}")
    args = ap.parse_args()
    if args.tiny_lora_r8:
        tiny_lora_r8()
        return
    if args.floor_random:
        for tag in RANDOM_ADAPTER_FIXTURES:
            if not args.only or args.only in tag:
                floor_random_adapter_fixture(tag)
        return
    if args.text_only:
        np.savez_compressed(os.path.join(HERE, "tiny_text.npz"), **text_case())
        print("tiny_text written")
        return
    if args.refinit:
        # full size, bs 8, at the reference initialisation, 5 SGD steps (gradients of the first AND the last step recorded: by then
        # LoRA's B, the bottleneck up-projection and Compacter's factors have moved, so every low-rank gradient kernel carries signal)
        for method, arch_name, tag, lora_r, redraw, full_layers in REFINIT_CASES:
            if args.only and args.only not in tag:
                continue
            if args.bf16_weights:
                legs = tuple(args.legs.split(",")) if args.legs else (("weights", "operands", "fp8") if "l14" in tag else ("weights", "operands"))
                floor_case(method, arch_name, tag, lora_r, redraw, full_layers, legs)
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
    tiny_lora_r8()
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
