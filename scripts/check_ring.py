#!/usr/bin/env python3
"""The 160x128x64 ring GEMM against the two-stage kernel (same operands, every epilogue the step uses)."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
from pevit_amd import _lib
lib = _lib.load()
P = lambda t: None if t is None else C.c_void_p(t.data_ptr())
S = lambda: C.c_void_p(torch.cuda.current_stream().cuda_stream)
torch.manual_seed(0)
worst = 0.0
for (M, N, K) in ((6400, 768, 3072), (6400, 768, 768), (1000, 384, 128), (161, 128, 64), (6272, 896, 2368)):
    for epi in (1, 4, 5, 2, 3):
        A = torch.randn(M, K, device="cuda").bfloat16()
        Nb = (N + 127) // 128 * 128
        B = (torch.randn(Nb, K, device="cuda") * 0.05).bfloat16()
        bias = torch.randn(N, device="cuda"); resid = torch.randn(M, N, device="cuda"); aux = torch.randn(M, N, device="cuda").bfloat16()
        outs = []
        for ring in (0, 2):
            lib.pevit_tune(b"gemm_ring", ring)
            outf = torch.zeros(M, N, device="cuda"); outb = torch.zeros(M, N, dtype=torch.bfloat16, device="cuda"); outb2 = torch.zeros_like(outb)
            rc = lib.pevit_op_gemm(S(), epi, P(A), K, P(B), K, Nb, M, N, K, P(bias), P(resid), N, P(outf), N, P(outb), N, P(outb2), N, P(aux), N, 0, 0, 0, 0)
            assert rc == 0, lib.pevit_last_error()
            torch.cuda.synchronize()
            outs.append((outf.clone(), outb.float().clone(), outb2.float().clone()))
        for a, b in zip(*outs):
            d = float((a - b).abs().max()); worst = max(worst, d)
            assert d <= 1e-2 * float(b.abs().max() + 1e-6), (M, N, K, epi, d)
        print(f"M={M} N={N} K={K} epi={epi}: ok", flush=True)
lib.pevit_tune(b"gemm_ring", 1)
print("worst abs difference", worst)
