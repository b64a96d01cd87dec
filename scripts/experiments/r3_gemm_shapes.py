#!/usr/bin/env python3
"""Per-shape timing of the 8-wave products of the ViT-B/32 step, old kernel vs staggered kernel (gemm_stagger / gemm_ksp),
interleaved in one process."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import bench_gemm as bg
SHAPES = [("c_fc fwd (gelu)", "BIAS_GELU", 6400, 3072, 768), ("c_proj bwd (dgelu)", "DGELU", 6400, 3072, 768),
          ("qkv fwd", "QKV", 6400, 2368, 768), ("square 4096", "BF16", 4096, 4096, 4096)]
MODES = [("old", 0, 1), ("stagger ksp1", 1, 1), ("stagger ksp2", 1, 2)]
ABL = [(0, "full"), (2, "no stores")]
for rnd in range(2):
    for label, stg, ksp in MODES:
        bg.tune("gemm_stagger", stg); bg.tune("gemm_ksp", ksp)
        for ab, abl in ABL:
            bg.tune("gemm_ablate", ab)
            print(f"---- {label} [{abl}] round {rnd}")
            for name, epi, M, N, K in SHAPES:
                bg.run(name, bg.EPI[epi], M, N, K, 768, 12, 50, iters=20)
bg.tune("gemm_ablate", 0); bg.tune("gemm_stagger", 1); bg.tune("gemm_ksp", 1)
