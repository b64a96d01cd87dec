"""Rounding-point-faithful CPU emulation of the HIP engine's bf16 path  --  TEST INFRASTRUCTURE, NOT PRODUCT CODE.

``oracle/ref_cpu.py`` restates the reference in fp32 (and, with ``operand_rounding``, with "both GEMM
operands rounded to bf16").  Two different bf16 roundings of the same randomly initialised tower
differ from each other about as much as each differs from fp32, so that mode can only calibrate a
wide gate.  This module rounds at EXACTLY the points where the engine stores a bf16 value
(``pevit_amd/csrc/capi.hip`` blocks_forward / blocks_backward) and nowhere
else, forward AND backward, so that the HIP path and this emulation differ only by the f32
summation order inside the contractions.  ``tests/test_gpu_emulation.py`` holds the production
kernels to it at logits <= 5e-3, gradients <= 2e-2 relative L2.

Storage points reproduced (engine buffer -> here):
  forward   xn1, xn2 (bf16 LayerNorm outputs)            RoundSTE(layer_norm)
            frozen weights bf16, q rows pre-scaled 1/8   bf(W)
            q, k, v in head layout (bf16)                QKVAug / LinearE(round_out)
            t = xn P with P stored bf16, t kept f32      QKVAug
            delta = ascale t bf16(Q)^T + b, q/v RMW       RoundSTE(q + delta)
            probabilities bf16, row sum of the ROUNDED p AttnCore
            attention output bf16                        AttnCore
            h (bf16), gelu(h) (bf16, from the stored h)  LinearE(round_out), QuickGeluE
            residual stream, LN statistics f32           plain f32
            class-token LayerNorm bf16 -> proj bf16      RoundSTE, LinearE
  backward  bf16 copy of the gradient stream (dyb)       every LinearE rounds its upstream gradient
            the residual gradient stream ITSELF bf16     RoundGrad at x_mid and at the block output (round 5; attention-site
            (LayerNorm backward read-modify-writes dyb)  and fused post-MLP adapters: capi.hip gstream16_on; the dx of the lowest
                                                         block walked leaves in f32)
            dh = bf16(acc * gelu'(h))                    QuickGeluE.backward
            dxn2, dO, dxn1 bf16 (dX GEMM outputs)        LinearE(round_dx) / QKVAug.backward
            dS bf16, P bf16 for dV, delta = sum P dP (N <= 64) / from bf16 O    AttnCore.backward
            dq, dk, dv bf16                              AttnCore.backward
            u = dDelta Q (Q bf16) f32, bf16 copy         DeltaFromT.backward / QKVAug.backward
            dP = xn^T bf16(u), dQ = dDelta^T bf16(t)     QKVAug / DeltaFromT backward
Reference lines: model.py:563-584 (adapter_forward), :786-817 (attention), :959-975 (block),
lora_model.py:490-514, adapter_model.py:264-336, compacter_model.py:302-308,432-503.
"""
from __future__ import annotations

import math

import torch
import torch.nn.functional as F

from . import ref_cpu as R


# Named rounding points: ``POINTS_OFF`` switches single storage points of the emulation off (the value then passes in f32), which
# is how scripts/r4_rounding_ablation.py attributes the bf16 path's distance from the f32 reference to individual buffers.
POINTS = ("w", "img", "xn", "qkv", "P", "Q_fwd", "q_delta", "p", "attn_out", "h", "gelu", "cls", "dyb", "dh", "dx", "ds", "dqkv",
          "p_bwd", "u", "Q_bwd", "t_bwd", "bottleneck", "gstream")
POINTS_OFF = set()


def bf(x, point=None):
    if point is not None and point in POINTS_OFF:
        return x
    return x.to(torch.bfloat16).to(torch.float32)


class RoundSTE(torch.autograd.Function):
    """bf16 storage of a forward activation; the gradient passes unchanged."""

    @staticmethod
    def forward(ctx, x, point=None):
        return bf(x, point)

    @staticmethod
    def backward(ctx, g):
        return g, None


class RoundGrad(torch.autograd.Function):
    """identity; the gradient arriving at this point of the residual stream is stored in bf16 (the engine's dyb buffer)."""

    @staticmethod
    def forward(ctx, x):
        return x.view_as(x)

    @staticmethod
    def backward(ctx, g):
        return bf(g, "gstream")


class LinearE(torch.autograd.Function):
    """y = x W^T + b with a FROZEN bf16 weight, x already bf16-valued.
    backward: the upstream gradient is rounded to bf16 first (the engine's GEMMs read a bf16 copy of it),
    dx = bf16(g) W, stored bf16 when ``round_dx``."""

    @staticmethod
    def forward(ctx, x, w_b, b, round_out, round_dx, out_point="h"):
        ctx.save_for_backward(w_b)
        ctx.round_dx = round_dx
        y = x @ w_b.t()
        if b is not None:
            y = y + b
        return bf(y, out_point) if round_out else y

    @staticmethod
    def backward(ctx, g):
        (w_b,) = ctx.saved_tensors
        dx = bf(g, "dyb") @ w_b
        return (bf(dx, "dx") if ctx.round_dx else dx), None, None, None, None, None


class LinearTrain(torch.autograd.Function):
    """Bottleneck products of the post-MLP adapters: trainable f32 weight used through its bf16 panel.
    dW = bf16(g)^T x (x is bf16-valued), dx = bf16(g) bf16(W) in f32; d bias from the rounded or the f32 gradient."""

    @staticmethod
    def forward(ctx, x, w, b, bias_from_f32):
        w_b = bf(w, "bottleneck")
        ctx.save_for_backward(x, w_b)
        ctx.bias_from_f32 = bias_from_f32
        return x @ w_b.t() + b

    @staticmethod
    def backward(ctx, g):
        x, w_b = ctx.saved_tensors
        g_b = bf(g, "bottleneck")
        g2, x2 = g_b.reshape(-1, g_b.shape[-1]), x.reshape(-1, x.shape[-1])
        db = (g if ctx.bias_from_f32 else g_b).reshape(-1, g.shape[-1]).sum(0)
        return g_b @ w_b, g2.t() @ x2, db, None


class QuickGeluE(torch.autograd.Function):
    """g = bf16(h sigmoid(1.702 h)) on the STORED h; dh = bf16(dg * gelu'(h)), dg unrounded (MFMA accumulator)."""

    @staticmethod
    def forward(ctx, h):
        ctx.save_for_backward(h)
        return bf(h * torch.sigmoid(1.702 * h), "gelu")

    @staticmethod
    def backward(ctx, g):
        (h,) = ctx.saved_tensors
        s = torch.sigmoid(1.702 * h)
        return bf(g * (s * (1.0 + 1.702 * h * (1.0 - s))), "dh")


class ReluE(torch.autograd.Function):
    """Adapter bottleneck: act = bf16(relu(pre)); d pre = bf16(d act * (act > 0))."""

    @staticmethod
    def forward(ctx, pre):
        act = bf(torch.relu(pre), "bottleneck")
        ctx.save_for_backward(act)
        return act

    @staticmethod
    def backward(ctx, g):
        (act,) = ctx.saved_tensors
        return bf(g * (act > 0).float(), "bottleneck")


def _gelu_new_grad(x):
    c = math.sqrt(2.0 / math.pi)
    u = c * (x + 0.044715 * x * x * x)
    th = torch.tanh(u)
    return 0.5 * (1.0 + th) + 0.5 * x * (1.0 - th * th) * c * (1.0 + 3.0 * 0.044715 * x * x)


class GeluNewE(torch.autograd.Function):
    """Compacter bottleneck: apre = bf16(pre) saved, act = bf16(gelu_new(apre)); d pre = bf16(d act * gelu_new'(apre))."""

    @staticmethod
    def forward(ctx, pre):
        a = bf(pre, "bottleneck")
        ctx.save_for_backward(a)
        return bf(R.gelu_new(a), "bottleneck")

    @staticmethod
    def backward(ctx, g):
        (a,) = ctx.saved_tensors
        return bf(g * _gelu_new_grad(a), "bottleneck")


class AttnCore(torch.autograd.Function):
    """softmax(q k^T) v per head on bf16-valued q, k, v (q carries 1/sqrt(64)); pevit_amd/csrc/attention.hip."""

    @staticmethod
    def forward(ctx, q, k, v):
        s = torch.bmm(q, k.transpose(1, 2))
        m = s.max(dim=-1, keepdim=True).values
        p_b = bf(torch.exp(s - m), "p")
        l = p_b.sum(-1, keepdim=True)                 # row sum of the ROUNDED probabilities
        o = bf(torch.bmm(p_b, v) * (1.0 / l), "attn_out")
        ctx.save_for_backward(q, k, v, o, m + torch.log(l))
        return o

    @staticmethod
    def backward(ctx, go):
        q, k, v, o, lse = ctx.saved_tensors
        p = torch.exp(torch.bmm(q, k.transpose(1, 2)) - lse)      # recomputed, f32
        dp = torch.bmm(go, v.transpose(1, 2))
        # N <= 64 (attn_bwd_kernel<2, true, .>): delta = sum_keys P dP from the f32 values of pass A; the larger shapes keep
        # delta = sum_d dO O from the stored (bf16) output
        delta = (p * dp).sum(-1, keepdim=True) if q.shape[1] <= 64 else (go * o).sum(-1, keepdim=True)
        ds = bf(p * (dp - delta), "ds")
        dq = bf(torch.bmm(ds, k), "dqkv")
        dk = bf(torch.bmm(ds.transpose(1, 2), q), "dqkv")
        dv = bf(torch.bmm(bf(p, "p_bwd").transpose(1, 2), go), "dqkv")
        return dq, dk, dv


class QKVAug(torch.autograd.Function):
    """[q|k|v] = bf16(xn W^T + b), t = xn bf16(P) (f32): the QKV GEMM with the 64 adapter columns.
    backward receives d[q|k|v] (bf16-valued) and -- by convention with DeltaFromT below -- u = dDelta Q (NOT yet
    multiplied by ascale): dxn = bf16(dqkv W + bf16(u) bf16(ascale P)^T) in ONE accumulation, dP = ascale xn^T bf16(u)."""

    @staticmethod
    def forward(ctx, xn, w_b, b, P, ascale):
        ctx.save_for_backward(xn, w_b, P)
        ctx.ascale = ascale
        return bf(xn @ w_b.t() + b, "qkv"), xn @ bf(P, "P")

    @staticmethod
    def backward(ctx, gqkv, u):
        xn, w_b, P = ctx.saved_tensors
        u_b = bf(u, "u")
        dxn = bf(gqkv @ w_b + u_b @ bf(ctx.ascale * P, "P").t(), "dx")
        dP = ctx.ascale * (xn.t() @ u_b)
        return dxn, None, None, dP, None


class DeltaFromT(torch.autograd.Function):
    """delta = ascale t bf16(Q)^T + b (the engine multiplies the bf16 panel of Q by t split into bf16 hi + lo: exact in t).
    backward: returns u = dDelta bf16(Q) for t (see QKVAug), dQ = ascale dDelta^T bf16(t), d b = colsum(dDelta)."""

    @staticmethod
    def forward(ctx, t, Q, bias, ascale):
        ctx.save_for_backward(t, Q)
        ctx.ascale = ascale
        ctx.has_bias = bias is not None
        d = ascale * (t @ bf(Q, "Q_fwd").t())        # Q enters the delta product as its bf16 panel (round 4), t in f32 (hi + lo)
        return d + bias if bias is not None else d

    @staticmethod
    def backward(ctx, g):
        t, Q = ctx.saved_tensors
        u = g @ bf(Q, "Q_bwd")
        dQ = ctx.ascale * (g.t() @ bf(t, "t_bwd"))
        return u, dQ, (g.sum(0) if ctx.has_bias else None), None


# --------------------------------------------------------------------------- #
def _ln_b(x, w, b, point="xn"):
    return RoundSTE.apply(F.layer_norm(x, (x.shape[-1],), w, b, 1e-5), point)


def _heads(x, N, B, H, hd):
    return x.contiguous().view(N, B * H, hd).transpose(0, 1).contiguous()


def _kadapt_panels(p, a, t):
    """P[:, j] = s_j (x) l_j, Q[:, j] = t_j (x) r_j (E x 32 each, q then v); both deltas use q_proj_adapter1_* (SURVEY 9.1)."""
    l = p[a + "q_proj_adapter1_left"][:, :, 0]          # (32, F)
    r = p[a + "q_proj_adapter1_right"][:, 0, :]         # (32, F)
    out = []
    for rule in ("1", "2"):
        s = p[t + f"phm_rule{rule}_left"][:, :, 0]      # (32, 32): [j][a]
        tt = p[t + f"phm_rule{rule}_right"][:, 0, :]    # (32, 32): [j][a]
        P = (s[:, :, None] * l[:, None, :]).reshape(32, -1).t()
        Q = (tt[:, :, None] * r[:, None, :]).reshape(32, -1).t()
        out.append((P, Q))
    return out


def _pad32(m):
    return torch.cat([m, torch.zeros(m.shape[0], 32 - m.shape[1])], dim=1) if m.shape[1] < 32 else m


def attention_site(xn, p, a, t, heads, method, wcache):
    """xn: (N,B,E) bf16-valued.  Returns the attention branch output before the residual add (f32)."""
    N, B, E = xn.shape
    hd = E // heads
    x2 = xn.reshape(N * B, E)
    if method == "kadaptation":
        (Pq, Qq), (Pv, Qv) = _kadapt_panels(p, a, t)
        ascale, bias = R.KADAPT_SCALE, p[a + "b"]
    else:
        r = p[a + "q_proj_adapter1.weight"].shape[0]
        Pq, Qq = _pad32(p[a + "q_proj_adapter1.weight"].t()), _pad32(p[a + "q_proj_adapter2.weight"])
        Pv, Qv = _pad32(p[a + "v_proj_adapter1.weight"].t()), _pad32(p[a + "v_proj_adapter2.weight"])
        ascale, bias = R.LORA_ALPHA / r, None
    w_b, b_s = wcache(a + "in_proj")
    qkv, tt = QKVAug.apply(x2, w_b, b_s, torch.cat([Pq, Pv], dim=1), ascale)
    q, k, v = qkv.view(N, B, 3 * E).chunk(3, dim=-1)
    q, k, v = _heads(q, N, B, heads, hd), _heads(k, N, B, heads, hd), _heads(v, N, B, heads, hd)
    dq = DeltaFromT.apply(tt[:, :32], Qq, bias, ascale)
    dv = DeltaFromT.apply(tt[:, 32:], Qv, bias, ascale)
    # raw reinterpretation of the (N,B,E)-contiguous delta; read-modify-write of the bf16 q / v buffers
    q = RoundSTE.apply(q + dq.reshape(B * heads, N, hd), "q_delta")
    v = RoundSTE.apply(v + dv.reshape(B * heads, N, hd), "q_delta")
    o = AttnCore.apply(q, k, v).transpose(0, 1).contiguous().view(N * B, E)
    wo_b, bo = wcache(a + "out_proj")
    return LinearE.apply(o, wo_b, bo, False, True).view(N, B, E)


def stock_attention(xn, p, a, heads, wcache):
    N, B, E = xn.shape
    hd = E // heads
    w_b, b_s = wcache(a + "in_proj")
    qkv = LinearE.apply(xn.reshape(N * B, E), w_b, b_s, True, True, "qkv")
    q, k, v = qkv.view(N, B, 3 * E).chunk(3, dim=-1)
    o = AttnCore.apply(_heads(q, N, B, heads, hd), _heads(k, N, B, heads, hd), _heads(v, N, B, heads, hd))
    wo_b, bo = wcache(a + "out_proj")
    return LinearE.apply(o.transpose(0, 1).contiguous().view(N * B, E), wo_b, bo, False, True).view(N, B, E)


def mlp_h(xn2, p, pre, wcache):
    """c_fc -> QuickGELU -> c_proj (+ bias), f32 output (the residual add happens outside)."""
    wfc, bfc = wcache(pre + "mlp.c_fc")
    wpr, bpr = wcache(pre + "mlp.c_proj")
    h = LinearE.apply(xn2, wfc, bfc, True, True)
    return LinearE.apply(QuickGeluE.apply(h), wpr, bpr, False, False)


def _phm_weight(rule, W_left, W_right):
    """(in, out) matrix of a PHM layer, f32 (compacter_model.py:302-308)."""
    return R.kron_sum(rule, torch.bmm(W_left, W_right))


def block(x, p, i, heads, method, wcache, tower="visual.transformer."):
    pre = f"{tower}resblocks.{i}."
    xn = _ln_b(x, p[pre + "ln_1.weight"], p[pre + "ln_1.bias"])
    if method in ("kadaptation", "lora"):
        x = RoundGrad.apply(x + attention_site(xn, p, pre + "attn.", tower, heads, method, wcache))
        xn2 = _ln_b(x, p[pre + "ln_2.weight"], p[pre + "ln_2.bias"])
        return RoundGrad.apply(x + mlp_h(xn2, p, pre, wcache))
    x = x + stock_attention(xn, p, pre + "attn.", heads, wcache)
    if method != "none":
        x = RoundGrad.apply(x)                      # post-MLP adapters on their fused kernels: the stream is bf16 here too
    xn2 = _ln_b(x, p[pre + "ln_2.weight"], p[pre + "ln_2.bias"])
    h = mlp_h(xn2, p, pre, wcache)
    if method == "none":
        return x + h
    a = pre + ("adapter." if method == "adapter" else "compacter.")
    z = _ln_b(h, p[a + "adapter_norm_before.weight"], p[a + "adapter_norm_before.bias"])
    if method == "adapter":
        act = ReluE.apply(LinearTrain.apply(z, p[a + "adapter_down.1.weight"], p[a + "adapter_down.1.bias"], False))
        up = LinearTrain.apply(act, p[a + "adapter_up.weight"], p[a + "adapter_up.bias"], True)
    else:
        rule = p[tower + "phm_rule"]
        wd = _phm_weight(rule, p[a + "adapter_down.1.W_left"], p[a + "adapter_down.1.W_right"]).t()
        wu = _phm_weight(rule, p[a + "adapter_up.W_left"], p[a + "adapter_up.W_right"]).t()
        act = GeluNewE.apply(LinearTrain.apply(z, wd, p[a + "adapter_down.1.b"], False))
        up = LinearTrain.apply(act, wu, p[a + "adapter_up.b"], True)
    return RoundGrad.apply(x + h + up)


def make_wcache(p):
    """bf16 copies of the frozen weights as the engine stores them (q rows and q bias of in_proj carry 1/8)."""
    cache = {}

    def get(name):
        if name not in cache:
            if name.endswith("in_proj"):
                w, b = p[name + "_weight"].detach().clone(), p[name + "_bias"].detach().clone()
                E = w.shape[1]
                w[:E] *= 0.125
                b[:E] *= 0.125
            else:
                w, b = p[name + ".weight"].detach(), p[name + ".bias"].detach()
            cache[name] = (bf(w, "w"), b.float())
        return cache[name]

    return get


def visual_forward(images, p, method):
    d = R.visual_dims(p)
    wcache = make_wcache(p)
    x = F.conv2d(bf(images, "img"), bf(p["visual.conv1.weight"], "w"), None, stride=d["patch"])
    x = x.reshape(x.shape[0], x.shape[1], -1).permute(0, 2, 1)
    cls = p["visual.class_embedding"] + torch.zeros(x.shape[0], 1, x.shape[-1])
    x = torch.cat([cls, x], dim=1) + p["visual.positional_embedding"]
    x = F.layer_norm(x, (x.shape[-1],), p["visual.ln_pre.weight"], p["visual.ln_pre.bias"], 1e-5)
    x = x.permute(1, 0, 2).contiguous()
    for i in range(d["layers"]):
        x = block(x, p, i, d["heads"], method, wcache)
    xc = _ln_b(x[0], p["visual.ln_post.weight"], p["visual.ln_post.bias"], "cls")        # class token of every image
    return LinearE.apply(xc, bf(p["visual.proj"].detach().t(), "w"), None, False, False)


class EmulTrainer(R.OracleTrainer):
    """OracleTrainer with the tower evaluated through the rounding-point emulation (head, loss and SGD stay f32,
    as in the engine)."""

    def forward(self, images):
        feat = visual_forward(images, self.p, self.method)
        out = F.batch_norm(feat, self.bn.running_mean, self.bn.running_var, None, None, self.bn.training, 0.1, 1e-5)
        return F.linear(out, self.head_w, self.head_b)


def transformer_forward(x, p, layers, heads, method):
    """The (N,B,E) -> (N,B,E) seam of pevit_transformer_forward (all rows of every block; no class-token pruning)."""
    wcache = make_wcache(p)
    for i in range(layers):
        x = block(x, p, i, heads, method, wcache)
    return x
