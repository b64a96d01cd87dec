#!/usr/bin/env python3
"""Ring GEMM: effect of staggering the k start of tiles that share a panel (dbg bit 16)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import bench_gemm as bg
bg.lib.pevit_tune(b"gemm_config", -1)
for ring in (5, 3):
    bg.lib.pevit_tune(b"gemm_ring", ring)
    for ab in (10, 26, 2, 18, 0, 16):
        bg.lib.pevit_tune(b"gemm_ablate", ab)
        print(f"==== ring {ring} ablate {ab}")
        bg.shapes(big=True)
bg.lib.pevit_tune(b"gemm_ablate", 0)
