#!/usr/bin/env python3
"""Ring-kernel variants vs stream-K vs plain on the N = 768 products."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import bench_gemm as bg
import torch
SH = [("c_proj fwd", "BIAS_RESID", 6400, 768, 3072), ("c_fc bwd", "BF16", 6400, 768, 3072), ("qkv bwd", "BF16", 6400, 768, 2368),
      ("out_proj fwd", "BIAS_RESID", 6400, 768, 768), ("out_proj bwd", "BF16", 6400, 768, 768)]
for ring, label in ((0, "stream-K / heuristic (warm-up)"), (0, "stream-K / heuristic"), (7, "ring 128x160, 2 stages"), (1, "ring 160x128, 3 stages"), (8, "k-split 128x160, 8 waves"), (9, "k-split 160x128, 8 waves"), (0, "stream-K / heuristic (again)")):
    bg.tune("gemm_ring", ring)
    print("----", label)
    for name, epi, M, N, K in SH:
        bg.run(name, bg.EPI[epi], M, N, K, 768, 12, 50)
bg.tune("gemm_ring", 0)
# correctness of the ring variants against torch
A = (torch.randn(6400, 3072, device="cuda")).bfloat16(); B = (torch.randn(768, 3072, device="cuda") * 0.05).bfloat16()
ref = A.float() @ B.float().T
import ctypes as C
for ring in (1, 7, 8, 9):
    bg.tune("gemm_ring", ring)
    out = torch.zeros(6400, 768, device="cuda")
    rc = bg.lib.pevit_op_gemm(bg.S(), 4, bg.P(A), 3072, bg.P(B), 3072, 768, 6400, 768, 3072, None, None, 0, bg.P(out), 768, None, 0, None, 0, None, 0, 0, 0, 0, 0)
    torch.cuda.synchronize()
    print("ring", ring, "rc", rc, "max rel err", float((out - ref).abs().max() / ref.abs().max()))
bg.tune("gemm_ring", 0)
