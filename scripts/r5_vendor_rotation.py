#!/usr/bin/env python3
"""The step's eight GEMM products, this build against the vendor library (torch.matmul -> hipBLASLt), LIKE FOR LIKE
(round-5 re-take of profiles/NOTES_gemm.md (r02_vendor_gemm_shapes); measurement only -- the product path never calls the vendor library).

Two cache regimes per product:
  * back to back   : the same buffers every call (operands and outputs warm in L2 / Infinity Cache) -- what the round-2
                     table measured;
  * in rotation    : ROT = 12 distinct sets of (A, B, outputs), like the 12 layers of the step, > 256 MB in all, so every
                     call finds its weights and activations where the step finds them at worst (HBM).
Three forms of this build's kernel per product: the production epilogue, the plain bf16-output epilogue (what the vendor
call computes) and the production epilogue with its stores skipped (gemm_ablate = 2: the k-loop alone).
usage: python scripts/r5_vendor_rotation.py [rot] > table.md"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from pevit_amd import _lib

lib = _lib.load()
P = lambda t: None if t is None else C.c_void_p(t.data_ptr())
S = lambda: C.c_void_p(torch.cuda.current_stream().cuda_stream)
EPI = dict(QKV=0, BIAS_RESID=1, BIAS_GELU=2, DGELU=3, F32=4, BF16=5)
ROT = int(sys.argv[1]) if len(sys.argv) > 1 else 12
E, H, NTOK = 768, 12, 50


def tune(key, val):
    assert lib.pevit_tune(None, key.encode(), int(val)) == 0


class Bufs:
    def __init__(self, epi, M, N, K):
        self.A = torch.randn(M, K, device="cuda").bfloat16()
        self.Nb = (N + 255) // 256 * 256
        self.B = (torch.randn(self.Nb, K, device="cuda") * 0.05).bfloat16()
        self.bias = torch.randn(max(N, 3 * E), device="cuda")
        self.resid = torch.randn(M, N, device="cuda") if epi == 1 else None
        self.outf = torch.empty(M, max(N, 64), device="cuda") if epi in (0, 1, 4) else None
        self.outb = torch.empty(3 * M * E if epi == 0 else M * N, dtype=torch.bfloat16, device="cuda")
        self.outb2 = torch.empty(M, N, dtype=torch.bfloat16, device="cuda") if epi == 2 else None
        self.aux = torch.randn(M, N, device="cuda").bfloat16() if epi == 3 else None
        self.vout = torch.empty(M, N, dtype=torch.bfloat16, device="cuda")


def ours(b, epi, M, N, K):
    rc = lib.pevit_op_gemm(S(), epi, P(b.A), K, P(b.B), K, b.Nb, M, N, K, P(b.bias), P(b.resid), N, P(b.outf), 64 if epi == 0 else N,
                           P(b.outb), N, P(b.outb2), N, P(b.aux), N, M * E, E, H, NTOK)
    assert rc == 0, lib.pevit_last_error()


def vendor(b, epi, M, N, K):
    torch.matmul(b.A, b.B[:N].T, out=b.vout)


def timed(fn, sets, reps):
    for s in sets:
        fn(s)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        for s in sets:
            fn(s)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / (reps * len(sets)) * 1e3


SH = [("qkv fwd (+t)", "QKV", 6400, 2368, 768), ("out_proj fwd", "BIAS_RESID", 6400, 768, 768), ("c_fc fwd (gelu)", "BIAS_GELU", 6400, 3072, 768),
      ("c_proj fwd", "BIAS_RESID", 6400, 768, 3072), ("c_proj bwd (dgelu)", "DGELU", 6400, 3072, 768), ("c_fc bwd", "F32", 6400, 768, 3072),
      ("out_proj bwd", "BF16", 6400, 768, 768), ("qkv bwd (+u)", "F32", 6400, 768, 2368)]
print(f"| product | M x N x K | regime | this build: production epilogue | k-loop only (stores skipped) | plain bf16 output | vendor, plain bf16 output |")
print("|---|---|---|---|---|---|---|")
tot = {}
for name, epin, M, N, K in SH:
    epi = EPI[epin]
    sets = [Bufs(epi, M, N, K) for _ in range(ROT)]
    for regime, ss, reps in (("back to back", sets[:1], 40), (f"rotation of {ROT}", sets, 4)):
        r = []
        for e, ab in ((epi, 0), (epi, 2), (EPI["BF16"], 0)):
            if e == EPI["BF16"] and epi == 0:
                r.append(float("nan")); continue      # the QKV product only exists with its head-layout epilogue
            tune("gemm_ablate", ab)
            r.append(timed(lambda s: ours(s, e, M, N, K), ss, reps))
        tune("gemm_ablate", 0)
        v = timed(lambda s: vendor(s, epi, M, N, K), ss, reps)
        t = tot.setdefault(regime, [0.0, 0.0, 0.0, 0.0])
        for i, x in enumerate(r + [v]):
            t[i] += 0.0 if x != x else x
        print(f"| {name} | {M} x {N} x {K} | {regime} | {r[0]:.1f} | {r[1]:.1f} | {r[2]:.1f} | {v:.1f} |", flush=True)
    del sets
    torch.cuda.empty_cache()
for regime, t in tot.items():
    print(f"| sum per layer | | {regime} | {t[0]:.1f} | {t[1]:.1f} | {t[2]:.1f} (without QKV) | {t[3]:.1f} |")
