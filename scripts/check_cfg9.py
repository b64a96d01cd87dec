#!/usr/bin/env python3
"""Config 9 (register-resident schedule on the 128x128 tile) against the default selection: results, then timings."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
from pevit_amd import _lib
lib = _lib.load()
P = lambda t: None if t is None else C.c_void_p(t.data_ptr())
S = lambda: C.c_void_p(torch.cuda.current_stream().cuda_stream)
torch.manual_seed(0)
for (M, N, K) in ((6400, 2304, 768), (6400, 768, 3072), (1000, 384, 128), (257, 256, 64), (6272, 896, 2368), (512, 512, 64)):
    for epi in (1, 4, 5, 2, 3):
        A = torch.randn(M, K, device="cuda").bfloat16()
        Nb = (N + 255) // 256 * 256
        B = (torch.randn(Nb, K, device="cuda") * 0.05).bfloat16()
        bias = torch.randn(N, device="cuda"); resid = torch.randn(M, N, device="cuda"); aux = torch.randn(M, N, device="cuda").bfloat16()
        outs = []
        for cfg in (0, 9):
            lib.pevit_tune(b"gemm_config", cfg)
            outf = torch.zeros(M, N, device="cuda"); outb = torch.zeros(M, N, dtype=torch.bfloat16, device="cuda"); outb2 = torch.zeros_like(outb)
            rc = lib.pevit_op_gemm(S(), epi, P(A), K, P(B), K, Nb, M, N, K, P(bias), P(resid), N, P(outf), N, P(outb), N, P(outb2), N, P(aux), N, 0, 0, 0, 0)
            assert rc == 0, lib.pevit_last_error()
            torch.cuda.synchronize()
            outs.append((outf.clone(), outb.float().clone(), outb2.float().clone()))
        for a, b in zip(*outs):
            assert torch.equal(a, b), (M, N, K, epi, float((a - b).abs().max()))
    print(f"M={M} N={N} K={K}: identical", flush=True)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import bench_gemm as bg
for cfg in (-1, 9, -1, 9):
    bg.lib.pevit_tune(b"gemm_config", cfg)
    print(f"==== gemm_config = {cfg}")
    bg.shapes(big=True)
bg.lib.pevit_tune(b"gemm_config", -1)
