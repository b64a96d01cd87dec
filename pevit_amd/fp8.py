"""Host-side statement of the fp8 weight format of the engine (include/pevit_hip.h: PEVIT_W_FP8_E4M3).

The engine packs the frozen block weights itself (``pevit_load_block`` -> csrc/fp8.hip); this module states the same
format in PyTorch so that callers can see exactly which weights an fp8 engine computes with
(``dequantized_state_dict``), e.g. to compare against a bf16 run of the same de-quantised weights
 or to export them.

    scale[r] = 2^ceil(log2(amax_r / 448))            one power of two per output channel (row of [out][in])
    code[r,c] = e4m3fn(W[r,c] / scale[r])            round-to-nearest-even, |.| <= 448 by construction
"""
from __future__ import annotations

import torch

E4M3_MAX = 448.0
# the four frozen products of a residual block (model.py:675,816,959-961)
BLOCK_WEIGHTS = ("attn.in_proj_weight", "attn.out_proj.weight", "mlp.c_fc.weight", "mlp.c_proj.weight")


def row_scales(w: torch.Tensor) -> torch.Tensor:
    amax = w.detach().float().abs().amax(dim=1)
    m, e = torch.frexp(amax / E4M3_MAX)                     # amax/448 = m * 2^e, m in [0.5, 1)
    exp = torch.where(m > 0.5, e, e - 1)
    s = torch.ldexp(torch.ones_like(amax), exp)
    return torch.where(amax > 0, s, torch.ones_like(s))


def quantize_rows(w: torch.Tensor):
    """(codes uint8 [rows, cols], scales f32 [rows]) of a [out][in] weight."""
    s = row_scales(w)
    q = (w.detach().float() / s[:, None]).to(torch.float8_e4m3fn)
    return q.view(torch.uint8), s


def dequantize_rows(codes: torch.Tensor, scales: torch.Tensor) -> torch.Tensor:
    return codes.view(torch.float8_e4m3fn).float() * scales[:, None]


def kperm(cols: int) -> torch.Tensor:
    """Storage position of input channel k inside its group of 128 (csrc/fp8.hip fp8_kperm):
    [k-tile parity][lane half][k-step][8]."""
    k = torch.arange(cols)
    kk = k & 127
    par, ks, half, j = kk >> 6, (kk >> 4) & 3, (kk >> 3) & 1, kk & 7
    return (k & ~127) + par * 64 + half * 32 + ks * 8 + j


def dequantized_state_dict(sd, prefix: str = "visual.transformer.resblocks."):
    """Copy of ``sd`` in which the frozen block weights are replaced by what an fp8 engine computes with."""
    out = dict(sd)
    for k, v in sd.items():
        if k.startswith(prefix) and k.endswith(BLOCK_WEIGHTS):
            codes, s = quantize_rows(v)
            out[k] = dequantize_rows(codes, s).to(v.dtype if v.dtype.is_floating_point else torch.float32)
    return out
