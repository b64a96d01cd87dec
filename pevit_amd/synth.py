"""Deterministic synthetic CLIP checkpoints in the OpenAI state-dict layout.

There is no network in the build/bench environment, so benchmarks and parity
tests run on random-initialised weights of the *architecture* the reference
loads (key names/shapes: /root/reference/vision_benchmark/evaluation/model.py:1210-1233,
SURVEY.md section 9.7).  Values follow the reference's own initialisers in spirit
(xavier-uniform in_proj, kaiming-uniform linears, width**-0.5 embeddings --
model.py:384-388,589,1024-1032) but LayerNorm affines and biases are perturbed
away from (1, 0) so that parity tests exercise them.
"""
from __future__ import annotations

import math
from collections import OrderedDict
from dataclasses import dataclass

import torch


@dataclass(frozen=True)
class VitArch:
    name: str
    width: int          # E
    layers: int         # L
    patch: int          # P
    resolution: int     # R
    embed_dim: int      # D (output of visual.proj)
    text_width: int = 512
    text_layers: int = 12
    context_length: int = 77
    vocab_size: int = 49408

    @property
    def heads(self) -> int:
        return self.width // 64

    @property
    def tokens(self) -> int:
        return (self.resolution // self.patch) ** 2 + 1


ARCHS = {
    # names as in clip_load._MODELS (clip_load.py:32-41)
    "ViT-B/32": VitArch("ViT-B/32", 768, 12, 32, 224, 512),
    "ViT-B/16": VitArch("ViT-B/16", 768, 12, 16, 224, 512),
    "ViT-L/14": VitArch("ViT-L/14", 1024, 24, 14, 224, 768, text_width=768),
    # ViT-B/32 geometry (width 768, 50 tokens, patch 32) cut to two blocks (whole-step parity tests at B = 128)
    "ViT-B/32-2L": VitArch("ViT-B/32-2L", 768, 2, 32, 224, 512),
    # small legal shapes for tests (head_dim stays 64, the only value CLIP uses)
    "tiny-128": VitArch("tiny-128", 128, 2, 16, 48, 64, text_width=64, text_layers=1,
                        context_length=8, vocab_size=32),
    "tiny-256": VitArch("tiny-256", 256, 3, 16, 64, 128, text_width=64, text_layers=1,
                        context_length=8, vocab_size=32),
    # token counts of ViT-B/16 (197) and ViT-L/14 (257, patch 14 -> padded im2col K) at test width
    "tiny-n197": VitArch("tiny-n197", 128, 2, 16, 224, 64, text_width=64, text_layers=1,
                         context_length=8, vocab_size=32),
    "tiny-n257": VitArch("tiny-n257", 128, 2, 14, 224, 64, text_width=64, text_layers=1,
                         context_length=8, vocab_size=32),
}


def _uniform(gen, shape, bound):
    return (torch.rand(shape, generator=gen, dtype=torch.float32) * 2 - 1) * bound


def _normal(gen, shape, std):
    return torch.randn(shape, generator=gen, dtype=torch.float32) * std


def _block(sd, prefix, width, gen):
    E = width
    xav = math.sqrt(6.0 / (E + 3 * E))
    sd[prefix + "attn.in_proj_weight"] = _uniform(gen, (3 * E, E), xav)
    sd[prefix + "attn.in_proj_bias"] = _normal(gen, (3 * E,), 0.02)
    kb = 1.0 / math.sqrt(E)
    sd[prefix + "attn.out_proj.weight"] = _uniform(gen, (E, E), kb)
    sd[prefix + "attn.out_proj.bias"] = _normal(gen, (E,), 0.02)
    sd[prefix + "ln_1.weight"] = 1.0 + _normal(gen, (E,), 0.1)
    sd[prefix + "ln_1.bias"] = _normal(gen, (E,), 0.1)
    sd[prefix + "mlp.c_fc.weight"] = _uniform(gen, (4 * E, E), kb)
    sd[prefix + "mlp.c_fc.bias"] = _uniform(gen, (4 * E,), kb)
    kb2 = 1.0 / math.sqrt(4 * E)
    sd[prefix + "mlp.c_proj.weight"] = _uniform(gen, (E, 4 * E), kb2)
    sd[prefix + "mlp.c_proj.bias"] = _uniform(gen, (E,), kb2)
    sd[prefix + "ln_2.weight"] = 1.0 + _normal(gen, (E,), 0.1)
    sd[prefix + "ln_2.bias"] = _normal(gen, (E,), 0.1)


def synth_state_dict(arch: VitArch | str, seed: int = 2, dtype=torch.float32,
                     text_tower: bool = True) -> "OrderedDict[str, torch.Tensor]":
    """Synthetic OpenAI-layout CLIP state-dict (CPU tensors).

    ``text_tower=False`` keeps only the handful of text-side keys that
    ``build_model`` needs to infer dimensions (one text block), which is what
    the throughput benchmark uses: the text encoder is not on the hot path.
    """
    if isinstance(arch, str):
        arch = ARCHS[arch]
    gen = torch.Generator(device="cpu")
    gen.manual_seed(seed)
    sd: "OrderedDict[str, torch.Tensor]" = OrderedDict()
    E, D = arch.width, arch.embed_dim
    scale = E ** -0.5
    sd["visual.class_embedding"] = _normal(gen, (E,), scale)
    sd["visual.positional_embedding"] = _normal(gen, (arch.tokens, E), scale)
    sd["visual.proj"] = _normal(gen, (E, D), scale)
    sd["visual.conv1.weight"] = _normal(gen, (E, 3, arch.patch, arch.patch),
                                        1.0 / math.sqrt(3 * arch.patch * arch.patch))
    sd["visual.ln_pre.weight"] = 1.0 + _normal(gen, (E,), 0.1)
    sd["visual.ln_pre.bias"] = _normal(gen, (E,), 0.1)
    sd["visual.ln_post.weight"] = 1.0 + _normal(gen, (E,), 0.1)
    sd["visual.ln_post.bias"] = _normal(gen, (E,), 0.1)
    for i in range(arch.layers):
        _block(sd, f"visual.transformer.resblocks.{i}.", E, gen)

    TW = arch.text_width
    n_text = arch.text_layers if text_tower else 1
    sd["positional_embedding"] = _normal(gen, (arch.context_length, TW), 0.01)
    sd["text_projection"] = _normal(gen, (TW, D), TW ** -0.5)
    sd["logit_scale"] = torch.tensor(math.log(1 / 0.07), dtype=torch.float32)
    sd["token_embedding.weight"] = _normal(gen, (arch.vocab_size if text_tower else 8, TW), 0.02)
    sd["ln_final.weight"] = 1.0 + _normal(gen, (TW,), 0.1)
    sd["ln_final.bias"] = _normal(gen, (TW,), 0.1)
    for i in range(n_text):
        _block(sd, f"transformer.resblocks.{i}.", TW, gen)
    if dtype != torch.float32:
        for k in sd:
            if sd[k].dim() >= 1 and not k.startswith(("visual.ln", "ln_final")):
                sd[k] = sd[k].to(dtype)
    return sd


def synth_batch(batch: int, resolution: int, num_classes: int, seed_img: int = 0, seed_lbl: int = 1):
    """CIFAR100-shaped batch after the reference preprocessing:
    images ~ N(0,1) fp32 (B,3,R,R), labels ~ U{0..C-1} int64."""
    g0 = torch.Generator(device="cpu"); g0.manual_seed(seed_img)
    g1 = torch.Generator(device="cpu"); g1.manual_seed(seed_lbl)
    images = torch.randn((batch, 3, resolution, resolution), generator=g0, dtype=torch.float32)
    labels = torch.randint(0, num_classes, (batch,), generator=g1, dtype=torch.int64)
    return images, labels


def randomize_adapters(named_params, seed: int = 3, scale: float = 1.0):
    """Give every trainable adapter tensor a non-degenerate seeded value.

    The reference initialisation leaves the Kronecker factors at exactly zero
    (model.py:533-539), which makes gradient parity vacuous.
    ``named_params`` is an iterable of (name, tensor); tensors are modified in
    place, in iteration order, from one generator stream.
    """
    gen = torch.Generator(device="cpu"); gen.manual_seed(seed)
    for name, p in named_params:
        if "phm_rule" in name and (name.endswith("_left") or name.endswith("_right")):
            std = 0.1                      # KAdaptation shared rule factors
        elif "phm_rule" in name:
            continue                       # Compacter rule: frozen U(-1,1), keep
        elif "norm" in name and name.endswith("weight"):
            v = 1.0 + torch.randn(p.shape, generator=gen) * 0.1
            p.data.copy_(v.to(p.dtype)); continue
        elif name.endswith("attn.b") or name.endswith(".b") or name.endswith("bias"):
            std = 0.05
        elif "adapter1_left" in name or "adapter1_right" in name:
            std = 0.05
        elif "W_left" in name or "W_right" in name:
            std = 0.2
        else:
            std = 0.02                     # LoRA A/B, bottleneck down/up
        v = torch.randn(p.shape, generator=gen, dtype=torch.float32) * (std * scale)
        p.data.copy_(v.to(p.dtype).to(p.device))


def reference_init_(named_params, method: str, seed: int = 7):
    """Adapter tensors at the reference's own initialisation, in place:
    KAdaptation: Kronecker factors 0, phm_rule factors U(-0.01,0.01), b = 0 (model.py:533-554,987-999);
    LoRA: A ~ N(0,0.02), B = 0 (lora_model.py:466-475); Adapter: N(0,0.02) weights, zero biases, LN (1,0)
    (adapter_model.py:285-295); Compacter: glorot-uniform(gain sqrt 2) W_left/W_right, zero biases, LN (1,0)."""
    g = torch.Generator(device="cpu"); g.manual_seed(seed)
    for name, p in named_params:
        shape = tuple(p.shape)
        if name.startswith("layers."):
            continue
        if "phm_rule" in name and (name.endswith("_left") or name.endswith("_right")):
            v = (torch.rand(shape, generator=g) * 2 - 1) * 0.01
        elif "norm" in name and name.endswith("weight"):
            v = torch.ones(shape)
        elif name.endswith(".b") or name.endswith("bias"):
            v = torch.zeros(shape)
        elif method == "kadaptation":
            v = torch.zeros(shape)
        elif method == "lora":
            v = torch.randn(shape, generator=g) * 0.02 if "adapter1" in name else torch.zeros(shape)
        elif method == "adapter":
            v = torch.randn(shape, generator=g) * 0.02
        else:   # compacter W_left (n, in, 1) / W_right (n, 1, out): xavier-uniform, gain sqrt(2), per slice
            fan_in, fan_out = shape[2], shape[1]
            bound = math.sqrt(2.0) * math.sqrt(6.0 / (fan_in + fan_out))
            v = (torch.rand(shape, generator=g) * 2 - 1) * bound
        p.data.copy_(v.to(p.dtype).to(p.device))
