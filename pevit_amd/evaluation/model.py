"""Host mirror of the reference's model entry points, backed by the HIP engine.

Keeps the importable names and the module/parameter naming of
``vision_benchmark/evaluation/model.py`` (and its lora / adapter / compacter siblings) of
eric-ai-lab/PEViT -- ``build_model(state_dict) -> CLIP`` (eval mode, fp32 parameters,
adapter keys keep their initial values unless the checkpoint has them: model.py:1210-1250),
``CLIP.encode_image``, ``.visual.input_resolution``, ``.visual.proj``, ``.logit_scale``,
``named_parameters()`` names (trainability is decided by substring match on
these names in the harness) -- while the vision tower itself executes in
``libpevit_hip.so``.  Nothing here computes the tower in PyTorch: on a machine without the
HIP library / a GPU, ``encode_image`` raises.

The text tower (``encode_text``, used once to initialise the zero-shot head) is plain host
PyTorch, as SURVEY 8f-4 prescribes; it is not on the accelerated path.
"""
from __future__ import annotations

import math
import os
import threading
import weakref
from collections import OrderedDict

import numpy as np
import torch
from torch import nn

from .. import _lib
from ..engine import HipEngine, adapter_param_spec
from ..synth import VitArch


# --------------------------------------------------------------------------- host-side pieces
class LayerNorm(nn.LayerNorm):
    """fp32 statistics regardless of input dtype (reference model.py:154-160)."""

    def forward(self, x):
        return super().forward(x.float()).type(x.dtype)


class QuickGELU(nn.Module):
    def forward(self, x):
        return x * torch.sigmoid(1.702 * x)


class _TextBlock(nn.Module):
    def __init__(self, width, heads, mask):
        super().__init__()
        self.attn = nn.MultiheadAttention(width, heads)
        self.ln_1 = LayerNorm(width)
        self.mlp = nn.Sequential(OrderedDict([("c_fc", nn.Linear(width, 4 * width)), ("gelu", QuickGELU()),
                                              ("c_proj", nn.Linear(4 * width, width))]))
        self.ln_2 = LayerNorm(width)
        self.attn_mask = mask

    def forward(self, x):
        m = self.attn_mask.to(dtype=x.dtype, device=x.device) if self.attn_mask is not None else None
        y = self.ln_1(x)
        x = x + self.attn(y, y, y, need_weights=False, attn_mask=m)[0]
        return x + self.mlp(self.ln_2(x))


class _TextTransformer(nn.Module):
    def __init__(self, width, layers, heads, mask):
        super().__init__()
        self.width, self.layers = width, layers
        self.resblocks = nn.Sequential(*[_TextBlock(width, heads, mask) for _ in range(layers)])

    def forward(self, x):
        return self.resblocks(x)


class _Params(nn.Module):
    """Named container: holds parameters / sub-containers under the reference's attribute names."""

    def add(self, name, tensor, trainable=False):
        head, _, rest = name.partition(".")
        if rest:
            if head not in self._modules:
                self.add_module(head, _Params())
            self._modules[head].add(rest, tensor, trainable)
        else:
            self.register_parameter(head, nn.Parameter(tensor, requires_grad=trainable))


# parameter registration order of one residual block, per method, exactly as the reference's
# named_parameters() lists it (checked against tests/golden/*.json: "all_names")
_FROZEN_BLOCK = ["attn.in_proj_weight", "attn.in_proj_bias", "attn.out_proj.weight", "attn.out_proj.bias",
                 "ln_1.weight", "ln_1.bias", "mlp.c_fc.weight", "mlp.c_fc.bias", "mlp.c_proj.weight",
                 "mlp.c_proj.bias", "ln_2.weight", "ln_2.bias"]


def _block_order(method, adapter_names):
    attn_direct = [n for n in adapter_names if n.startswith("attn.") and n.count(".") == 1]   # KAdaptation factors, b
    attn_child = [n for n in adapter_names if n.startswith("attn.") and n.count(".") > 1]     # LoRA Linear modules
    other = [n for n in adapter_names if not n.startswith("attn.")]                           # adapter.* / compacter.*
    order = list(other)                      # post-MLP adapter module is registered before self.attn
    order += ["attn.in_proj_weight", "attn.in_proj_bias"] + attn_direct
    order += ["attn.out_proj.weight", "attn.out_proj.bias"] + attn_child
    order += _FROZEN_BLOCK[4:]
    return order


class _EngineCache:
    """Sweep-level reuse: ``train_task`` builds a new Classifier for each of the ~90 sweep runs;
    the frozen, pre-packed bf16 backbone in the engine arena is identical every time, so an engine whose owner
    has been garbage-collected is handed to the next model with the same checkpoint fingerprint."""

    def __init__(self):
        self._items = []          # (key, engine, weakref-to-owner)
        self._lock = threading.Lock()      # the runs of a concurrent sweep (evaluation/_harness.py run_tasks) bind their engines from worker threads

    @staticmethod
    def _key(visual, device, num_classes):
        fp = getattr(visual, "_frozen_fingerprint", None)
        return None if fp is None else (fp, visual.method, visual.lora_rank, num_classes, str(device), visual.weight_format)

    def acquire(self, visual, device, num_classes, max_batch):
        key = self._key(visual, device, num_classes)
        if key is None:
            return None
        with self._lock:
            for i, (k, e, w) in enumerate(self._items):
                if k == key and w() is None:
                    self._items[i] = (k, e, weakref.ref(visual))      # owned from here on: a second thread cannot be handed the same engine
                    return e
        return None

    def register(self, visual, eng):
        key = self._key(visual, eng.device, eng.num_classes)
        if key is not None:
            with self._lock:
                self._items = [(k, e, w) for k, e, w in self._items if w() is not None][-3:]   # bound resident engines
                self._items.append((key, eng, weakref.ref(visual)))

    def clear(self):
        with self._lock:
            self._items.clear()


_ENGINES = _EngineCache()


def _check_generation(eng, generation):
    """The engine keeps the activations of ONE forward.  A backward through an older graph (gradient accumulation
    over two live graphs, two image batches in one loss, a grad-enabled validation forward in between) would silently
    differentiate the wrong activations: fail instead."""
    if eng.forward_generation != generation:
        raise _lib.PevitError("backward through a forward whose activations the engine no longer holds: another forward "
                              "ran in between (the HIP engine keeps one activation workspace, which every forward -- with or without "
                              "grad -- overwrites; call backward before the next forward of this tower)")


class _TransformerFn(torch.autograd.Function):
    """The operator seam of the reference, Transformer.forward(x: (N,B,E)) -> (N,B,E) (model.py:1013), on the engine."""

    @staticmethod
    def forward(ctx, x, visual, save, *params):
        eng = visual._engine
        y = eng.transformer_forward(x, save=save)
        ctx.visual, ctx.generation, ctx.need_dx = visual, eng.forward_generation, x.requires_grad
        return y

    @staticmethod
    def backward(ctx, dy):
        eng = ctx.visual._engine
        _check_generation(eng, ctx.generation)
        eng.grads[:eng.n_tower].zero_()
        dx = eng.transformer_backward(dy, need_dx=ctx.need_dx)
        views = eng.param_views(eng.grads.clone())
        mask = ctx.visual._has_grad
        grads = [views["visual." + n] if mask[n] else None for n in ctx.visual._trainable_names]
        return (dx, None, None, *grads)


class _BlockFn(torch.autograd.Function):
    """ResidualAttentionBlock.forward (model.py:972-975) of block ``index`` on the engine (pevit_blocks_forward / _backward)."""

    @staticmethod
    def forward(ctx, x, visual, index, save, *params):
        eng = visual._engine
        y = eng.blocks_forward(x, index, index + 1, save=save)
        ctx.visual, ctx.index, ctx.generation, ctx.need_dx = visual, index, eng.block_generation[index], x.requires_grad
        return y

    @staticmethod
    def backward(ctx, dy):
        eng = ctx.visual._engine
        if eng.block_generation.get(ctx.index) != ctx.generation:
            raise _lib.PevitError(f"backward through block {ctx.index} whose activations the engine no longer holds: another forward "
                                  "through this block ran in between")
        eng.grads[:eng.n_tower].zero_()
        dx = eng.blocks_backward(dy, ctx.index, ctx.index + 1, need_dx=ctx.need_dx)
        views = eng.param_views(eng.grads.clone())
        mask = ctx.visual._has_grad
        grads = [views["visual." + n] if mask[n] else None for n in ctx.visual._trainable_names]
        return (dx, None, None, None, *grads)


class _Block(_Params):
    """visual.transformer.resblocks[i]: parameters under the reference's names, callable like the reference's
    ResidualAttentionBlock (reference-side code that walks ``resblocks`` keeps working; the whole-tower call is one engine op)."""

    def forward(self, x):
        visual = self._owner()
        eng = visual.engine()
        eng.ensure_batch(x.shape[1])
        params = visual._trainable_params()
        save = torch.is_grad_enabled() and (x.requires_grad or any(p.requires_grad for p in params))
        return _BlockFn.apply(x.contiguous().float(), visual, self._index, save, *params)


class _Resblocks(_Params):
    """The reference's nn.Sequential of blocks: indexable, iterable, callable (applies the blocks in order)."""

    def __getitem__(self, i):
        return self._modules[str(i)]

    def __len__(self):
        return len(self._modules)

    def __iter__(self):
        return iter(self._modules.values())

    def forward(self, x):
        for blk in self:
            x = blk(x)
        return x


class _Transformer(_Params):
    """visual.transformer: owner of the shared phm_rule* factors and of resblocks (model.py:978-1014), callable like the
    reference's module.  ``kdropout`` mirrors MultiheadAttention.kdropout = Dropout(0.5) (model.py:516,582): the reference
    never leaves eval mode, where it is the identity; the engine has no dropout, so train mode is refused (see
    VisionTransformer.train)."""
    kdropout = 0.5

    def forward(self, x):
        visual = self._owner()
        eng = visual.engine()
        eng.ensure_batch(x.shape[1])
        params = visual._trainable_params()
        save = torch.is_grad_enabled() and (x.requires_grad or any(p.requires_grad for p in params))
        return _TransformerFn.apply(x.contiguous().float(), visual, save, *params)


class _VisualFn(torch.autograd.Function):
    """images -> features through the HIP engine; gradients of the trainable tensors come back as
    the engine's flat-buffer views (the reference gets them from autograd over aten ops)."""

    @staticmethod
    def forward(ctx, images, visual, save, *params):
        eng = visual._engine
        feat = eng.visual_forward(images, save=save)
        ctx.visual, ctx.n, ctx.generation = visual, len(params), eng.forward_generation
        return feat

    @staticmethod
    def backward(ctx, dfeat):
        eng = ctx.visual._engine
        _check_generation(eng, ctx.generation)
        eng.grads[:eng.n_tower].zero_()
        eng.visual_backward(dfeat)
        views = eng.param_views(eng.grads.clone())         # ONE copy; autograd takes views of it as the gradients
        mask = ctx.visual._has_grad
        grads = [views["visual." + n] if mask[n] else None for n in ctx.visual._trainable_names]
        return (None, None, None, *grads)


class VisionTransformer(nn.Module):
    """visual.* of the reference (model.py:1017-1051); forward is one engine call."""

    def __init__(self, arch: VitArch, method: str, lora_rank: int):
        super().__init__()
        self.input_resolution, self.output_dim = arch.resolution, arch.embed_dim
        self.arch, self.method, self.lora_rank = arch, method, lora_rank
        # storage of the frozen block weights in the engine: "bf16", or "fp8" (e4m3 codes + per-channel scales, BASELINE
        # config 5; KAdaptation / LoRA).  Set before the model moves to the GPU (config key MODEL.WEIGHT_FORMAT, or the
        # PEVIT_WEIGHT_FORMAT environment variable for scripts that cannot pass config keys)
        self.weight_format = os.environ.get("PEVIT_WEIGHT_FORMAT", "bf16")
        self._engine: HipEngine | None = None
        self._train_params = None
        E, L = arch.width, arch.layers
        scale = E ** -0.5
        spec = adapter_param_spec(method, E, L, lora_rank)
        self._trainable_names = [n[len("visual."):] for n, _, tr in spec if tr]
        shapes = {n[len("visual."):]: (s, tr) for n, s, tr in spec}
        self.register_parameter("class_embedding", nn.Parameter(scale * torch.randn(E), requires_grad=False))
        self.register_parameter("positional_embedding", nn.Parameter(scale * torch.randn(arch.tokens, E), requires_grad=False))
        self.register_parameter("proj", nn.Parameter(scale * torch.randn(E, arch.embed_dim), requires_grad=False))
        self.conv1 = _Params(); self.conv1.add("weight", torch.zeros(E, 3, arch.patch, arch.patch))
        self.ln_pre = _Params(); self.ln_pre.add("weight", torch.ones(E)); self.ln_pre.add("bias", torch.zeros(E))
        self.transformer = _Transformer()
        object.__setattr__(self.transformer, "_owner", weakref.ref(self))     # not a submodule edge: no cycle in the module tree
        for n, (s, tr) in shapes.items():              # shared rules live on the Transformer (model.py:987-999)
            if n.startswith("transformer.phm_rule"):
                self.transformer.add(n[len("transformer."):], torch.zeros(s), tr)
        self.transformer.add_module("resblocks", _Resblocks())
        frozen_shapes = {"attn.in_proj_weight": (3 * E, E), "attn.in_proj_bias": (3 * E,), "attn.out_proj.weight": (E, E),
                         "attn.out_proj.bias": (E,), "ln_1.weight": (E,), "ln_1.bias": (E,), "mlp.c_fc.weight": (4 * E, E),
                         "mlp.c_fc.bias": (4 * E,), "mlp.c_proj.weight": (E, 4 * E), "mlp.c_proj.bias": (E,),
                         "ln_2.weight": (E,), "ln_2.bias": (E,)}
        for i in range(L):
            pre = f"transformer.resblocks.{i}."
            mine = [n[len(pre):] for n in shapes if n.startswith(pre)]
            blk = _Block()
            object.__setattr__(blk, "_owner", weakref.ref(self))
            object.__setattr__(blk, "_index", i)
            for n in _block_order(method, mine):
                if n in frozen_shapes:
                    blk.add(n, torch.zeros(frozen_shapes[n]))
                else:
                    s, tr = shapes[pre + n]
                    blk.add(n, torch.zeros(s), tr)
            self.transformer.resblocks.add_module(str(i), blk)
        self.ln_post = _Params(); self.ln_post.add("weight", torch.ones(E)); self.ln_post.add("bias", torch.zeros(E))
        self._has_grad = {n: True for n in self._trainable_names}

    def train(self, mode: bool = True):
        if mode and self.method == "kadaptation":
            raise _lib.PevitError("the KAdaptation tower has kdropout = Dropout(0.5) on the Kronecker weights in train mode "
                                  "(model.py:516,582); the reference never enters it (build_model returns .eval(), "
                                  "kadaptation_clip.py never calls .train()) and the HIP engine does not implement it: keep the "
                                  "backbone in eval mode")
        return super().train(mode)

    # -- engine life cycle ---------------------------------------------------------------
    def _apply(self, fn, *a, **kw):
        out = super()._apply(fn, *a, **kw)
        eng = self._engine
        if eng is not None and self.class_embedding.device != eng.device:
            self._engine = None            # moved away: parameters are plain tensors again (values kept by fn)
        return out

    def engine(self) -> HipEngine:
        """The HIP context of this tower; created on first use once the module lives on a GPU (the head size
        of the surrounding Classifier, ``_num_classes``, is known by then)."""
        if self._engine is None:
            dev = self.class_embedding.device
            if dev.type != "cuda":
                raise _lib.PevitError("the vision tower runs only in the HIP engine: move the model to a GPU "
                                      "(model.cuda()) -- there is no PyTorch/CPU fallback")
            self.attach_engine(dev)
        return self._engine

    def attach_engine(self, device, num_classes: int | None = None, max_batch: int | None = None):
        """Create (or, for sweep runs over the same checkpoint, re-use) the HIP context on ``device``, load the
        frozen weights into it and re-seat every trainable Parameter as a view of the engine's flat buffer
        (the values the Parameters hold now are preserved).  Frozen tensors are packed (bf16, transposed copies)
        at this point: edit them afterwards and call ``attach_engine`` again -- nothing on the reference's path
        does."""
        num_classes = num_classes or getattr(self, "_num_classes", 1)
        max_batch = max_batch or getattr(self, "_max_batch", 128)
        object.__setattr__(self.transformer, "_owner", weakref.ref(self))     # (a deep copy carries the original's reference)
        sd = {"visual." + k: v for k, v in self.state_dict().items()}
        eng = _ENGINES.acquire(self, torch.device(device), num_classes, max_batch)
        if eng is None:
            eng = HipEngine(self.arch, self.method, num_classes, max_batch, lora_rank=self.lora_rank, device=device,
                            weight_format=self.weight_format)
            eng.load_state_dict(sd)                    # frozen weights + overlay of the adapter values in sd
            _ENGINES.register(self, eng)
        else:
            eng.reset_run()
            eng.ensure_batch(max_batch)
            eng.load_trainable(sd)
        views = eng.param_views()
        named = dict(self.named_parameters())
        for n in self._trainable_names:
            named[n].data = views["visual." + n]
        mask, off = eng.grad_mask_host, 0
        for n in self._trainable_names:
            k = named[n].numel()
            self._has_grad[n] = bool(mask[off:off + k].any())
            off += k
        if getattr(self, "_input_norm", None) is not None:
            eng.set_input_normalization(*self._input_norm)
        self._engine = eng
        return eng

    def merge_proj_into_head(self):
        """TRAIN.MERGE_ENCODER_AND_HEAD_PROJ (kadaptation_clip.py:146-158): the reference sets ``visual.proj = None`` and
        multiplies it into the head.  The engine always applies a projection, so here ``proj`` becomes the E x E identity (the
        tower then returns the ln_post'd class token, rounded to bf16 once) and the output width becomes E; returns the old
        (E, D) matrix for the caller to fold into its head.  Any existing engine is dropped."""
        import dataclasses
        old = self.proj.data.detach().clone()
        E = old.shape[0]
        self.proj = nn.Parameter(torch.eye(E, dtype=old.dtype, device=old.device), requires_grad=False)
        self.arch = dataclasses.replace(self.arch, embed_dim=E)
        self.output_dim = E
        fp = getattr(self, "_frozen_fingerprint", None)
        if fp is not None:
            self._frozen_fingerprint = (fp, "proj-merged")
        self._engine = None
        return old

    def _trainable_params(self):
        """The trainable Parameter objects in flat-buffer order (their identity survives device moves and
        re-seating, so the list is built once)."""
        if self._train_params is None:
            named = dict(self.named_parameters())
            self._train_params = [named[n] for n in self._trainable_names]
        return self._train_params

    def set_input_normalization(self, mean, std):
        """uint8 batches (B,3,R,R) are accepted once the preprocessing constants are known: the reference's ToTensor +
        Normalize(INPUT.MEAN, INPUT.STD) then runs inside the engine's patch gather (a quarter of the bytes to upload)."""
        self._input_norm = (tuple(float(v) for v in mean), tuple(float(v) for v in std))
        if self._engine is not None:
            self._engine.set_input_normalization(*self._input_norm)

    def forward(self, x):
        eng = self.engine()
        eng.ensure_batch(x.shape[0])
        params = self._trainable_params()
        save = torch.is_grad_enabled() and any(p.requires_grad for p in params)   # grad mode is off inside Function.forward
        if x.dtype == torch.uint8:
            if getattr(self, "_input_norm", None) is None:
                raise _lib.PevitError("uint8 images need the preprocessing constants: call visual.set_input_normalization(mean, std)")
            return _VisualFn.apply(x.contiguous(), self, save, *params)
        return _VisualFn.apply(x.contiguous().float(), self, save, *params)


class CLIP(nn.Module):
    """Same attribute surface as the reference CLIP (model.py:1054-1183)."""

    def __init__(self, embed_dim, image_resolution, vision_layers, vision_width, vision_patch_size, context_length,
                 vocab_size, transformer_width, transformer_heads, transformer_layers, method="kadaptation", lora_rank=4):
        super().__init__()
        self.context_length = context_length
        arch = VitArch("custom", vision_width, vision_layers, vision_patch_size, image_resolution, embed_dim,
                       text_width=transformer_width, text_layers=transformer_layers, context_length=context_length,
                       vocab_size=vocab_size)
        self.positional_embedding = nn.Parameter(torch.empty(context_length, transformer_width).normal_(std=0.01))
        self.text_projection = nn.Parameter(torch.empty(transformer_width, embed_dim).normal_(std=transformer_width ** -0.5))
        self.logit_scale = nn.Parameter(torch.ones([]) * np.log(1 / 0.07))
        self.visual = VisionTransformer(arch, method, lora_rank)
        self.transformer = _TextTransformer(transformer_width, transformer_layers, transformer_heads, self.build_attention_mask())
        self.vocab_size = vocab_size
        self.token_embedding = nn.Embedding(vocab_size, transformer_width)
        self.ln_final = LayerNorm(transformer_width)

    def build_attention_mask(self):
        mask = torch.empty(self.context_length, self.context_length)
        mask.fill_(float("-inf"))
        mask.triu_(1)
        return mask

    @property
    def dtype(self):
        return self.visual.conv1.weight.dtype

    def encode_image(self, image):
        if image.dtype == torch.uint8:          # raw pixels: normalised inside the engine (visual.set_input_normalization)
            return self.visual(image)
        return self.visual(image.type(self.dtype))

    def encode_text(self, text):
        x = self.token_embedding(text).type(self.dtype)
        x = x + self.positional_embedding.type(self.dtype)
        x = self.transformer(x.permute(1, 0, 2)).permute(1, 0, 2)
        x = self.ln_final(x).type(self.dtype)
        return x[torch.arange(x.shape[0]), text.argmax(dim=-1)] @ self.text_projection

    def forward(self, image, text):
        img, txt = self.encode_image(image), self.encode_text(text)
        img = img / img.norm(dim=-1, keepdim=True)
        txt = txt / txt.norm(dim=-1, keepdim=True)
        logits = self.logit_scale.exp() * img @ txt.t()
        return logits, logits.t()


def _init_adapters(model: CLIP, method: str):
    """Reference initialisation of the added tensors (model.py:533-554,987-999; lora_model.py:466-475;
    adapter_model.py:285-295; compacter_model.py:254-297,511-519)."""
    for name, p in model.visual.named_parameters():
        with torch.no_grad():
            if "phm_rule" in name and (name.endswith("_left") or name.endswith("_right")):
                p.uniform_(-0.01, 0.01)
            elif name.endswith("phm_rule"):
                p.uniform_(-1, 1)
            elif "adapter_norm_before" in name:
                p.fill_(1.0 if name.endswith("weight") else 0.0)
            elif method == "lora" and "adapter1.weight" in name:
                p.normal_(std=0.02)
            elif method == "adapter" and (name.endswith("adapter_down.1.weight") or name.endswith("adapter_up.weight")):
                p.normal_(mean=0.0, std=0.02)
            elif method == "compacter" and ("W_left" in name or "W_right" in name):
                for i in range(p.shape[0]):
                    nn.init.xavier_uniform_(p[i], gain=math.sqrt(2))
            # everything else added by the adapters starts at zero (already zero-filled)


def _fingerprint(sd) -> tuple:
    """Cheap identity of the frozen vision weights (shape + a few sampled values of three tensors)."""
    out = []
    for k in ("visual.conv1.weight", "visual.proj", "visual.transformer.resblocks.0.mlp.c_fc.weight"):
        t = sd[k].detach().float().flatten()
        idx = torch.linspace(0, t.numel() - 1, 16).long()
        out.append((tuple(sd[k].shape), tuple(round(float(v), 7) for v in t[idx])))
    return tuple(out)


def build_peft_model(state_dict: dict, method: str, lora_rank: int = 4) -> CLIP:
    if "visual.proj" not in state_dict:
        raise RuntimeError("only the ViT variants of CLIP are supported by the PEFT engines (the reference's "
                           "adapters are injected into VisionTransformer only)")
    vision_width = state_dict["visual.conv1.weight"].shape[0]
    vision_layers = len([k for k in state_dict if k.startswith("visual.") and k.endswith(".attn.in_proj_weight")])
    vision_patch_size = state_dict["visual.conv1.weight"].shape[-1]
    grid_size = round((state_dict["visual.positional_embedding"].shape[0] - 1) ** 0.5)
    image_resolution = vision_patch_size * grid_size
    embed_dim = state_dict["text_projection"].shape[1]
    context_length = state_dict["positional_embedding"].shape[0]
    vocab_size = state_dict["token_embedding.weight"].shape[0]
    transformer_width = state_dict["ln_final.weight"].shape[0]
    transformer_heads = transformer_width // 64
    transformer_layers = len(set(k.split(".")[2] for k in state_dict if k.startswith("transformer.resblocks")))
    model = CLIP(embed_dim, image_resolution, vision_layers, vision_width, vision_patch_size, context_length, vocab_size,
                 transformer_width, transformer_heads, transformer_layers, method=method, lora_rank=lora_rank)
    _init_adapters(model, method)
    for key in ("input_resolution", "context_length", "vocab_size"):
        state_dict.pop(key, None)
    own = model.state_dict()
    own.update({k: state_dict[k].float() for k in own if k in state_dict})      # checkpoint overlays, adapters keep init
    model.load_state_dict(own)
    model.visual._frozen_fingerprint = state_dict.get("__fingerprint__", None) or _fingerprint(own)
    return model.eval()


def build_model(state_dict: dict) -> CLIP:
    """KAdaptation CLIP (reference: evaluation/model.py:1210-1250)."""
    return build_peft_model(state_dict, "kadaptation")
