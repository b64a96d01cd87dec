#!/usr/bin/env python3
"""The step's GEMM shapes through the vendor library (torch.matmul -> hipBLASLt, bf16 in, bf16 out, NO fused epilogue) next to
this build's kernels with their epilogues (measurement only: the product path never calls the vendor library)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch
import bench_gemm as bg

def vendor(M, N, K, iters=30):
    A = torch.randn(M, K, device="cuda").bfloat16(); B = (torch.randn(N, K, device="cuda") * 0.05).bfloat16()
    for _ in range(5): torch.matmul(A, B.T)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): torch.matmul(A, B.T)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3

import io, contextlib
SH = [("qkv fwd (+t)", "QKV", 6400, 2368, 768), ("out_proj fwd", "BIAS_RESID", 6400, 768, 768), ("c_fc fwd (gelu)", "BIAS_GELU", 6400, 3072, 768),
      ("c_proj fwd", "BIAS_RESID", 6400, 768, 3072), ("c_proj bwd (dgelu)", "DGELU", 6400, 3072, 768), ("c_fc bwd", "BF16", 6400, 768, 3072),
      ("out_proj bwd", "BF16", 6400, 768, 768), ("qkv bwd (+u)", "BF16", 6400, 768, 2368)]
print("| product | M x N x K | this build, fused epilogue (us) | same, epilogue stores skipped | vendor library, plain bf16 output (us) |")
print("|---|---|---|---|---|")
tot = [0, 0, 0]
for name, epi, M, N, K in SH:
    r = []
    for ab in (0, 2):
        bg.tune("gemm_ablate", ab)
        buf = io.StringIO()
        with contextlib.redirect_stdout(buf):
            r.append(bg.run(name, bg.EPI[epi], M, N, K, 768, 12, 50))
    bg.tune("gemm_ablate", 0)
    v = vendor(M, N, K)
    tot[0] += r[0]; tot[1] += r[1]; tot[2] += v
    print(f"| {name} | {M} x {N} x {K} | {r[0]:.1f} | {r[1]:.1f} | {v:.1f} |", flush=True)
print(f"| sum per layer | | {tot[0]:.1f} | {tot[1]:.1f} | {tot[2]:.1f} |")
