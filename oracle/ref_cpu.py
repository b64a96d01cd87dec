"""CPU oracle for the PEViT hot path  --  TEST INFRASTRUCTURE, NOT PRODUCT CODE.

A plain PyTorch-eager fp32 restatement of the reference algorithm for the
fine-tune step of a CLIP vision tower with PEFT adapters.  Only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
this module; the product (``pevit_amd``) never does.

It follows the reference op-for-op, *including* the behaviours SURVEY.md
section 9 lists (dense Kronecker H materialised per call, V-delta built from
Wq, raw ``reshape`` of the (N,B,E) delta into (B*H,N,hd), MLP evaluated twice
in the bottleneck Adapter).  Written as functions over a flat ``{name: tensor}``
dict that uses the reference's parameter names, so the same dict can be fed to
the reference's ``load_state_dict`` when the fixtures are generated
(tests/golden/make_golden.py) -- that is how this oracle is pinned:
tests/test_oracle_golden.py checks it against tensors produced by importing
/root/reference/vision_benchmark/evaluation/{model,lora_model,adapter_model,
compacter_model}.py in the build container.

Reference lines restated (paths relative to
/root/reference/vision_benchmark/evaluation/):
  kron_sum               model.py:406-417, :575/:580 (.sum(0))
  kadapt_delta           model.py:563-584
  lora_delta             lora_model.py:490-514
  attention_site_mha     model.py:675,729-740,786-817 ; lora_model.py:718-733
  stock_mha              torch.nn.MultiheadAttention as used by
                         adapter_model.py:314, compacter_model.py:481
  bottleneck_adapter     adapter_model.py:264-282
  phm_linear             compacter_model.py:302-308
  compacter_adapter      compacter_model.py:432-448
  block_forward          model.py:972-975 ; adapter_model.py:330-336 ;
                         compacter_model.py:497-503
  visual_forward         model.py:1034-1051
  classifier_forward     kadaptation_clip.py:176-185 (BatchNorm1d affine=False + Linear)
  trainable_rule         kadaptation_clip.py:104-122 ; lora_clip.py / adapter_tuning_clip.py
                         ('adapter' in name) ; compacter_clip.py:122 ('compacter' in name)
  sgd_step               optim/build.py:120-127 (torch.optim.SGD momentum, no nesterov)
"""
from __future__ import annotations

import math
from collections import OrderedDict

import torch
import torch.nn.functional as F

METHODS = ("kadaptation", "lora", "adapter", "compacter")
KADAPT_SCALE = 128 / 4 * 5      # model.py:564  -> 160
PHM_DIM_KADAPT = 32             # model.py:485, :984
PHM_DIM_COMPACTER = 4           # compacter_model.py:398,417,512
BOTTLENECK = 64                 # adapter_model.py:305, compacter_model.py:472
LORA_ALPHA = 128                # lora_model.py:463


# --------------------------------------------------------------------------- #
# parameter inventory
# --------------------------------------------------------------------------- #
def adapter_param_shapes(method: str, width: int, layers: int, lora_r: int = 4):
    """Adapter parameters the reference *adds* to the OpenAI layout, keyed by the
    reference's names.  Order is irrelevant here (dict)."""
    E = width
    out = OrderedDict()
    t = "visual.transformer."
    if method == "kadaptation":
        n = PHM_DIM_KADAPT
        for r in ("1", "2"):
            out[f"{t}phm_rule{r}_left"] = (n, n, 1)
            out[f"{t}phm_rule{r}_right"] = (n, 1, n)
        for i in range(layers):
            a = f"{t}resblocks.{i}.attn."
            for m in ("q", "v"):
                out[f"{a}{m}_proj_adapter1_left"] = (n, E // n, 1)
                out[f"{a}{m}_proj_adapter1_right"] = (n, 1, E // n)
            out[f"{a}b"] = (E,)
    elif method == "lora":
        for i in range(layers):
            a = f"{t}resblocks.{i}.attn."
            for m in ("q", "v"):
                out[f"{a}{m}_proj_adapter1.weight"] = (lora_r, E)
                out[f"{a}{m}_proj_adapter2.weight"] = (E, lora_r)
    elif method == "adapter":
        for i in range(layers):
            a = f"{t}resblocks.{i}.adapter."
            out[f"{a}adapter_norm_before.weight"] = (E,)
            out[f"{a}adapter_norm_before.bias"] = (E,)
            out[f"{a}adapter_down.1.weight"] = (BOTTLENECK, E)
            out[f"{a}adapter_down.1.bias"] = (BOTTLENECK,)
            out[f"{a}adapter_up.weight"] = (E, BOTTLENECK)
            out[f"{a}adapter_up.bias"] = (E,)
    elif method == "compacter":
        n = PHM_DIM_COMPACTER
        out[f"{t}phm_rule"] = (n, n, n)
        for i in range(layers):
            a = f"{t}resblocks.{i}.compacter."
            out[f"{a}adapter_norm_before.weight"] = (E,)
            out[f"{a}adapter_norm_before.bias"] = (E,)
            out[f"{a}adapter_down.1.W_left"] = (n, E // n, 1)
            out[f"{a}adapter_down.1.W_right"] = (n, 1, BOTTLENECK // n)
            out[f"{a}adapter_down.1.b"] = (BOTTLENECK,)
            out[f"{a}adapter_up.W_left"] = (n, BOTTLENECK // n, 1)
            out[f"{a}adapter_up.W_right"] = (n, 1, E // n)
            out[f"{a}adapter_up.b"] = (E,)
    else:
        raise ValueError(method)
    return out


def is_trainable(method: str, name: str) -> bool:
    """requires_grad rule of the four harnesses, applied to *backbone* names
    (i.e. without the 'backbone.' prefix the Classifier adds)."""
    if method == "kadaptation":
        return ("adapter" in name) or ("phm_rule" in name) or ("attn.b" in name)
    if method in ("lora", "adapter"):
        return "adapter" in name
    if method == "compacter":
        return "compacter" in name
    raise ValueError(method)


def init_adapter_params(method, width, layers, lora_r=4, seed=0):
    """Adapter tensors at the *reference* initialisation (SURVEY 8a a4,a11-a13)."""
    g = torch.Generator(device="cpu"); g.manual_seed(seed)
    out = OrderedDict()
    for name, shape in adapter_param_shapes(method, width, layers, lora_r).items():
        if method == "kadaptation":
            if "phm_rule" in name:                      # model.py:987-999
                v = (torch.rand(shape, generator=g) * 2 - 1) * 0.01
            else:                                       # model.py:533-539,554
                v = torch.zeros(shape)
        elif method == "lora":                          # lora_model.py:466-475
            v = torch.randn(shape, generator=g) * 0.02 if "adapter1" in name else torch.zeros(shape)
        elif method == "adapter":                       # adapter_model.py:285-295
            if name.endswith("norm_before.weight"):
                v = torch.ones(shape)
            elif name.endswith("bias"):
                v = torch.zeros(shape)
            else:
                v = torch.randn(shape, generator=g) * 0.02
        else:                                           # compacter_model.py:254-288,511-519
            if name.endswith("phm_rule"):
                v = torch.rand(shape, generator=g) * 2 - 1
            elif name.endswith("norm_before.weight"):
                v = torch.ones(shape)
            elif name.endswith(".b") or name.endswith("bias"):
                v = torch.zeros(shape)
            else:  # glorot_uniform with gain sqrt(2) on each (in, out) slice
                v = torch.empty(shape)
                for i in range(shape[0]):
                    fan_in, fan_out = shape[2], shape[1]   # torch convention for 2-D (rows, cols)
                    bound = math.sqrt(2.0) * math.sqrt(6.0 / (fan_in + fan_out))
                    v[i] = (torch.rand(shape[1:], generator=g) * 2 - 1) * bound
        out[name] = v.float()
    return out


# --------------------------------------------------------------------------- #
# optional operand rounding
# --------------------------------------------------------------------------- #
# With OPERAND_DTYPE = torch.bfloat16 every contraction of the *frozen-backbone path* (conv,
# linear, bmm, x @ H, x @ proj) rounds its two operands to bf16 and accumulates in f32 -- the
# arithmetic class of a bf16-MFMA engine, with nothing else changed.  Tests use it to measure
# how far bf16 operand rounding ALONE moves logits / gradients away from the f32 reference on a
# given case, which calibrates the tolerance of the HIP-vs-reference comparisons.
OPERAND_DTYPE = None


class operand_rounding:
    def __init__(self, dtype):
        self.dtype = dtype

    def __enter__(self):
        global OPERAND_DTYPE
        self.prev, OPERAND_DTYPE = OPERAND_DTYPE, self.dtype

    def __exit__(self, *a):
        global OPERAND_DTYPE
        OPERAND_DTYPE = self.prev


class _RoundSTE(torch.autograd.Function):
    """round-to-dtype in forward, identity (also rounded) in backward."""

    @staticmethod
    def forward(ctx, x, dtype):
        ctx.dtype = dtype
        return x.to(dtype).to(x.dtype)

    @staticmethod
    def backward(ctx, g):
        return g.to(ctx.dtype).to(g.dtype), None


def _r(x):
    return x if OPERAND_DTYPE is None else _RoundSTE.apply(x, OPERAND_DTYPE)


def _linear(x, w, b=None):
    return F.linear(_r(x), _r(w), b)


# --------------------------------------------------------------------------- #
# building blocks
# --------------------------------------------------------------------------- #
def layer_norm(x, w, b, eps=1e-5):
    return F.layer_norm(x.float(), (x.shape[-1],), w, b, eps)


def quick_gelu(x):
    return x * torch.sigmoid(1.702 * x)


def gelu_new(x):
    # transformers.activations "gelu_new" (compacter_model.py:8,172)
    return 0.5 * x * (1.0 + torch.tanh(math.sqrt(2.0 / math.pi) * (x + 0.044715 * torch.pow(x, 3.0))))


def kron_sum(A, B):
    """sum_i kron(A[i], B[i]) -- the reference materialises the (b, a*k, c*p) tensor."""
    res = torch.einsum("bac,bkp->bakcp", A, B).reshape(A.size(0), A.size(1) * B.size(1), A.size(2) * B.size(2))
    return res.sum(0)


def kadapt_delta(x, p, a, t, which):
    """x: (N,B,E).  ``which`` in {'q','v'}; both use q_proj_adapter1_* (SURVEY 9.1)."""
    Wq = torch.bmm(p[a + "q_proj_adapter1_left"], p[a + "q_proj_adapter1_right"])
    r = "1" if which == "q" else "2"
    rule = torch.bmm(p[t + f"phm_rule{r}_left"], p[t + f"phm_rule{r}_right"])
    H = kron_sum(rule, Wq)
    return torch.matmul(_r(x), _r(H)) * KADAPT_SCALE + p[a + "b"]


def lora_delta(x, w1, w2):
    r = w1.shape[0]
    return torch.matmul(torch.matmul(_r(x), _r(w1.T)), w2.T) * (LORA_ALPHA / r)


def _heads(x, N, B, H, hd):
    return x.contiguous().view(N, B * H, hd).transpose(0, 1)


def attention_site_mha(x, p, a, t, heads, method):
    """Self-attention with the delta injected at q and v. x: (N,B,E) contiguous."""
    N, B, E = x.shape
    hd = E // heads
    qkv = _linear(x, p[a + "in_proj_weight"], p[a + "in_proj_bias"])
    q, k, v = qkv.chunk(3, dim=-1)
    q = _heads(q, N, B, heads, hd)
    k = _heads(k, N, B, heads, hd)
    v = _heads(v, N, B, heads, hd)
    q = q / math.sqrt(hd)
    if method == "kadaptation":
        dq = kadapt_delta(x, p, a, t, "q")
        dv = kadapt_delta(x, p, a, t, "v")
    else:
        dq = lora_delta(x, p[a + "q_proj_adapter1.weight"], p[a + "q_proj_adapter2.weight"])
        dv = lora_delta(x, p[a + "v_proj_adapter1.weight"], p[a + "v_proj_adapter2.weight"])
    # raw reinterpretation of the contiguous (N,B,E) buffer
    q = q.contiguous() + dq.reshape(B * heads, N, hd)
    v = v.contiguous() + dv.reshape(B * heads, N, hd)
    w = torch.softmax(torch.bmm(_r(q), _r(k).transpose(-2, -1)), dim=-1)
    o = torch.bmm(_r(w), _r(v)).transpose(0, 1).contiguous().view(N * B, E)
    o = _linear(o, p[a + "out_proj.weight"], p[a + "out_proj.bias"])
    return o.view(N, B, E)


def stock_mha(x, p, a, heads):
    N, B, E = x.shape
    hd = E // heads
    qkv = _linear(x, p[a + "in_proj_weight"], p[a + "in_proj_bias"])
    q, k, v = qkv.chunk(3, dim=-1)
    q = _heads(q, N, B, heads, hd) / math.sqrt(hd)
    k = _heads(k, N, B, heads, hd)
    v = _heads(v, N, B, heads, hd)
    w = torch.softmax(torch.bmm(_r(q), _r(k).transpose(-2, -1)), dim=-1)
    o = torch.bmm(_r(w), _r(v)).transpose(0, 1).contiguous().view(N * B, E)
    return _linear(o, p[a + "out_proj.weight"], p[a + "out_proj.bias"]).view(N, B, E)


def mlp(x, p, pre):
    h = _linear(x, p[pre + "mlp.c_fc.weight"], p[pre + "mlp.c_fc.bias"])
    return _linear(quick_gelu(h), p[pre + "mlp.c_proj.weight"], p[pre + "mlp.c_proj.bias"])


def bottleneck_adapter(h, res, p, a):
    z = F.layer_norm(h, (h.shape[-1],), p[a + "adapter_norm_before.weight"], p[a + "adapter_norm_before.bias"], 1e-5)
    z = F.relu(F.linear(z, p[a + "adapter_down.1.weight"], p[a + "adapter_down.1.bias"]))
    up = F.linear(z, p[a + "adapter_up.weight"], p[a + "adapter_up.bias"])
    return up + res


def phm_linear(x, rule, W_left, W_right, b):
    H = kron_sum(rule, torch.bmm(W_left, W_right))
    return torch.matmul(x, H) + b


def compacter_adapter(h, p, a, rule):
    z = F.layer_norm(h, (h.shape[-1],), p[a + "adapter_norm_before.weight"], p[a + "adapter_norm_before.bias"], 1e-5)
    z = phm_linear(z, rule, p[a + "adapter_down.1.W_left"], p[a + "adapter_down.1.W_right"], p[a + "adapter_down.1.b"])
    z = gelu_new(z)
    up = phm_linear(z, rule, p[a + "adapter_up.W_left"], p[a + "adapter_up.W_right"], p[a + "adapter_up.b"])
    return up + h


def block_forward(x, p, i, heads, method, tower="visual.transformer."):
    pre = f"{tower}resblocks.{i}."
    xn = layer_norm(x, p[pre + "ln_1.weight"], p[pre + "ln_1.bias"])
    if method in ("kadaptation", "lora"):
        x = x + attention_site_mha(xn, p, pre + "attn.", tower, heads, method)
        return x + mlp(layer_norm(x, p[pre + "ln_2.weight"], p[pre + "ln_2.bias"]), p, pre)
    x = x + stock_mha(xn, p, pre + "attn.", heads)
    xn2 = layer_norm(x, p[pre + "ln_2.weight"], p[pre + "ln_2.bias"])
    if method == "adapter":
        # the reference evaluates the MLP twice (adapter_model.py:333)
        return x + bottleneck_adapter(mlp(xn2, p, pre), mlp(xn2, p, pre), p, pre + "adapter.")
    if method == "compacter":
        return x + compacter_adapter(mlp(xn2, p, pre), p, pre + "compacter.", p[tower + "phm_rule"])
    if method == "none":
        return x + mlp(xn2, p, pre)
    raise ValueError(method)


def transformer_forward(x, p, layers, heads, method, tower="visual.transformer."):
    for i in range(layers):
        x = block_forward(x, p, i, heads, method, tower)
    return x


def visual_dims(p):
    E = p["visual.conv1.weight"].shape[0]
    P = p["visual.conv1.weight"].shape[-1]
    L = len([k for k in p if k.startswith("visual.") and k.endswith(".attn.in_proj_weight")])
    N = p["visual.positional_embedding"].shape[0]
    return dict(width=E, patch=P, layers=L, tokens=N, heads=E // 64,
                resolution=P * round((N - 1) ** 0.5), out_dim=p["visual.proj"].shape[1])


def visual_forward(images, p, method, return_tokens=False):
    d = visual_dims(p)
    x = F.conv2d(_r(images), _r(p["visual.conv1.weight"]), None, stride=d["patch"])
    x = x.reshape(x.shape[0], x.shape[1], -1).permute(0, 2, 1)
    cls = p["visual.class_embedding"] + torch.zeros(x.shape[0], 1, x.shape[-1], dtype=x.dtype)
    x = torch.cat([cls, x], dim=1) + p["visual.positional_embedding"]
    x = layer_norm(x, p["visual.ln_pre.weight"], p["visual.ln_pre.bias"])
    x = x.permute(1, 0, 2)
    x = transformer_forward(x, p, d["layers"], d["heads"], method)
    tokens = x
    x = x.permute(1, 0, 2)
    x = layer_norm(x[:, 0, :], p["visual.ln_post.weight"], p["visual.ln_post.bias"])
    x = _r(x) @ _r(p["visual.proj"])
    return (x, tokens) if return_tokens else x


# --------------------------------------------------------------------------- #
# classifier + one fine-tune step
# --------------------------------------------------------------------------- #
class BNState:
    """BatchNorm1d(D, affine=False) buffers (kadaptation_clip.py:128-131)."""

    def __init__(self, dim):
        self.running_mean = torch.zeros(dim)
        self.running_var = torch.ones(dim)
        self.training = True      # train mode until the first validate()


def classifier_forward(images, p, head_w, head_b, bn: BNState, method):
    feat = visual_forward(images, p, method)
    out = F.batch_norm(feat, bn.running_mean, bn.running_var, None, None, bn.training, 0.1, 1e-5)
    return F.linear(out, head_w, head_b)


def trainable_names(p, method):
    return [k for k in p if k.startswith("visual.") and is_trainable(method, k)]


class OracleTrainer:
    """forward -> CE -> backward -> SGD(momentum) exactly as train_one does
    (kadaptation_clip.py:347-354) on a dict of tensors."""

    def __init__(self, p, method, num_classes, lr=0.01, wd=0.0, momentum=0.9, head_seed=5):
        self.method = method
        self.p = {k: v.detach().clone().float() for k, v in p.items()}
        D = self.p["visual.proj"].shape[1]
        g = torch.Generator(device="cpu"); g.manual_seed(head_seed)
        bound = 1.0 / math.sqrt(D)
        self.head_w = ((torch.rand((num_classes, D), generator=g) * 2 - 1) * bound)
        self.head_b = ((torch.rand((num_classes,), generator=g) * 2 - 1) * bound)
        self.bn = BNState(D)
        self.names = trainable_names(self.p, method)
        for k in self.names:
            self.p[k].requires_grad_(True)
        self.head_w.requires_grad_(True)
        self.head_b.requires_grad_(True)
        self.params = [self.p[k] for k in self.names] + [self.head_w, self.head_b]
        self.opt = torch.optim.SGD(self.params, lr=lr, momentum=momentum, weight_decay=wd, nesterov=False)

    def n_trainable(self):
        return sum(t.numel() for t in self.params)

    def forward(self, images):
        return classifier_forward(images, self.p, self.head_w, self.head_b, self.bn, self.method)

    def loss_and_grads(self, images, labels):
        self.opt.zero_grad(set_to_none=True)
        logits = self.forward(images)
        loss = F.cross_entropy(logits, labels)
        loss.backward()
        return logits.detach(), loss.detach()

    def step(self, images, labels):
        logits, loss = self.loss_and_grads(images, labels)
        self.opt.step()
        return logits, loss
