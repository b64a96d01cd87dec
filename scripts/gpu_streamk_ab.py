#!/usr/bin/env python3
"""Stream-K vs the plain 128x128 tiling on the few-tile products, by k-iterations per workgroup, with the hand-off ablated
(gemm_ablate 16)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import bench_gemm as bg
SH = [("c_proj fwd", "BIAS_RESID", 6400, 768, 3072), ("c_fc bwd", "BF16", 6400, 768, 3072)]
bg.tune("gemm_config", 0); bg.tune("gemm_streamk", 0)
print("---- plain 128x128")
for name, epi, M, N, K in SH:
    bg.run(name, bg.EPI[epi], M, N, K, 768, 12, 50)
bg.tune("gemm_config", -1); bg.tune("gemm_streamk", 2)
for share in (0, 48, 40, 36, 32, 30, 29):
    for ab in (0, 16):
        bg.tune("gemm_sk_share", share); bg.tune("gemm_ablate", ab)
        print("---- stream-K, share", share, "(equal split)" if share == 0 else "", "[no hand-off]" if ab else "")
        for name, epi, M, N, K in SH:
            bg.run(name, bg.EPI[epi], M, N, K, 768, 12, 50)
bg.tune("gemm_ablate", 0); bg.tune("gemm_config", -1); bg.tune("gemm_streamk", 1); bg.tune("gemm_sk_share", 0)
