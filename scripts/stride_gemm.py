#!/usr/bin/env python3
"""Does the row stride of the operands matter (L2 channel camping)?  Same GEMMs with lda/ldb padded."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
from pevit_amd import _lib
lib = _lib.load()
P = lambda t: None if t is None else C.c_void_p(t.data_ptr())
S = lambda: C.c_void_p(torch.cuda.current_stream().cuda_stream)

def run(M, N, K, pad_a, pad_b, iters=20):
    lda, ldb = K + pad_a, K + pad_b
    A = torch.randn(M, lda, device="cuda").bfloat16()
    Nb = (N + 127) // 128 * 128
    B = (torch.randn(Nb, ldb, device="cuda") * 0.05).bfloat16()
    out = torch.empty(M, N, dtype=torch.bfloat16, device="cuda")
    def call():
        assert lib.pevit_op_gemm(S(), 5, P(A), lda, P(B), ldb, Nb, M, N, K, None, None, 0, None, 0, P(out), N, None, 0, None, 0, 0, 0, 0, 0) == 0
    for _ in range(3): call()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): call()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / iters * 1e3
    return us, 2.0 * M * N * K / us / 1e6

for ring in (5, 0):
    lib.pevit_tune(b"gemm_ring", ring)
    for ab in (0, 10, 2):
        lib.pevit_tune(b"gemm_ablate", ab)
        print(f"==== ring {ring} ablate {ab}")
        for (M, N, K) in ((6400, 768, 3072), (6400, 3072, 768), (6400, 768, 768), (6400, 2304, 768), (4096, 4096, 4096)):
            row = []
            for pa, pb in ((0, 0), (64, 64), (128, 128), (64, 0), (0, 64), (32, 32)):
                us, tf = run(M, N, K, pa, pb)
                row.append(f"pad({pa},{pb}) {us:6.1f}us {tf:6.0f}TF")
            print(f"M={M} N={N} K={K}: " + " | ".join(row), flush=True)
lib.pevit_tune(b"gemm_ablate", 0)
