#!/usr/bin/env python3
"""Reference point: what the vendor library (torch.mm -> hipBLASLt/rocBLAS) reaches on the step's GEMM shapes
(plain bf16 NT products, no fused epilogue).  Measurement only; the product path never calls it."""
import torch
M = 6400
shapes = [("qkv", 2304, 768), ("out_proj", 768, 768), ("c_fc", 3072, 768), ("c_proj", 768, 3072), ("dqkv", 768, 2304), ("square", 4096, 4096)]
for name, N, K in shapes:
    m = 4096 if name == "square" else M
    a = torch.randn(m, K, device="cuda").bfloat16()
    w = (torch.randn(N, K, device="cuda") * 0.05).bfloat16()
    for _ in range(5):
        c = a @ w.t()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(30):
        c = a @ w.t()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 30 * 1e3
    print(f"{name:10s} M={m} N={N} K={K}: {us:7.1f} us  {2.0 * m * N * K / us / 1e6:7.1f} TF (bf16 out, no epilogue)")
