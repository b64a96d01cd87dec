#!/usr/bin/env python3
"""Two 160x256 tiles per CU (persistent loop: the first tile's stores drain under the second tile's k-loop) against one 320x256 tile."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import bench_gemm as bg
SHAPES = [("c_fc fwd (gelu)", "BIAS_GELU", 6400, 3072, 768), ("c_proj bwd (dgelu)", "DGELU", 6400, 3072, 768), ("qkv fwd", "QKV", 6400, 2368, 768)]
for rnd in range(2):
    for cfg in (-1, 9, 4):
        bg.tune("gemm_config", cfg)
        print(f"---- gemm_config {cfg} round {rnd}")
        for name, epi, M, N, K in SHAPES:
            bg.run(name, bg.EPI[epi], M, N, K, 768, 12, 50, iters=20)
bg.tune("gemm_config", -1)
