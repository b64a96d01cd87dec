#!/usr/bin/env python3
"""Upper bound for a stream-K decomposition of the N = 768, K = 3072 products: a problem whose 512 tiles of 128x128 carry the
per-workgroup share of k-iterations (300 x 48 / 512 = 28) that stream-K would hand to each of the 512 resident workgroups."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import bench_gemm as bg
bg.tune("gemm_config", 0)
for ab, label in ((0, "full"), (2, "no stores")):
    bg.tune("gemm_ablate", ab)
    print("----", label)
    bg.run("c_proj fwd (300 tiles x 48)", bg.EPI["BIAS_RESID"], 6400, 768, 3072, 768, 12, 50)
    bg.run("512 tiles x 28 k-iterations", bg.EPI["BIAS_RESID"], 8192, 1024, 1792, 768, 12, 50)
    bg.run("512 tiles x 24", bg.EPI["BIAS_RESID"], 8192, 1024, 1536, 768, 12, 50)
    bg.run("c_fc bwd bf16 (300 x 48)", bg.EPI["BF16"], 6400, 768, 3072, 768, 12, 50)
    bg.run("512 tiles x 28 bf16", bg.EPI["BF16"], 8192, 1024, 1792, 768, 12, 50)
    bg.run("256 tiles x 48 bf16 (one per CU)", bg.EPI["BF16"], 4096, 1024, 3072, 768, 12, 50)
bg.tune("gemm_ablate", 0); bg.tune("gemm_config", -1)
