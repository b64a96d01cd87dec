#!/usr/bin/env python3
"""Where a GEMM's time goes: full / no stores / no stores + no operand stream (compute only) / no stores + no compute
(stream only), per tile configuration, on a few shapes of the step."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import bench_gemm as bg
SHAPES = [("c_fc fwd (gelu)", "BIAS_GELU", 6400, 3072, 768), ("c_proj fwd", "BIAS_RESID", 6400, 768, 3072),
          ("qkv fwd", "QKV", 6400, 2368, 768), ("square 4096", "BF16", 4096, 4096, 4096)]
MODES = [(0, "full"), (2, "no stores"), (6, "compute only"), (10, "stream only")]
for cfg in [int(c) for c in (sys.argv[1] if len(sys.argv) > 1 else "0,1,4,5").split(",")]:
    bg.tune("gemm_config", cfg)
    for ab, label in MODES:
        bg.tune("gemm_ablate", ab)
        print(f"---- gemm_config {cfg} ({bg.NAMES[cfg]})  [{label}]")
        for name, epi, M, N, K in SHAPES:
            bg.run(name, bg.EPI[epi], M, N, K, 768, 12, 50, iters=20)
bg.tune("gemm_ablate", 0); bg.tune("gemm_config", -1)
