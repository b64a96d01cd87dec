#!/usr/bin/env python3
"""Stream-K k-iterations-per-workgroup sweep (M=6400, N=768, K from argv, default 3072), with and without the hand-off."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import bench_gemm as bg
bg.tune("gemm_config", -1); bg.tune("gemm_streamk", 2)
K = int(sys.argv[1]) if len(sys.argv) > 1 else 3072
nk = K // 64
for share in list(range(max(8, 300 * nk // 512 - 2), nk + 1)):
    res = []
    for ab in (0, 16):
        bg.tune("gemm_sk_share", share); bg.tune("gemm_ablate", ab)
        import io, contextlib
        buf = io.StringIO()
        with contextlib.redirect_stdout(buf):
            us = bg.run("x", bg.EPI["BF16"], 6400, 768, K, 768, 12, 50)
        res.append(us)
    wgs = (300 * nk + share - 1) // share
    print(f"share {share:3d}  workgroups {wgs:4d}  grid {(wgs + 7) & ~7:4d}   {res[0]:6.1f} us   no hand-off {res[1]:6.1f} us", flush=True)
bg.tune("gemm_ablate", 0); bg.tune("gemm_streamk", 1); bg.tune("gemm_sk_share", 0)
