#!/usr/bin/env python3
"""Where does a GEMM launch spend its time?  k-loop only / epilogue only / both, per step shape."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import bench_gemm as bg
for cfg in (-1, 5):
    bg.lib.pevit_tune(b"gemm_config", cfg)
    for ab, what in ((0, "full"), (1, "no k-loop (prologue+epilogue)"), (2, "no stores (k-loop + LDS transpose)")):
        bg.lib.pevit_tune(b"gemm_ablate", ab)
        print(f"==== config {cfg} ablate {ab}: {what}")
        bg.shapes()
bg.lib.pevit_tune(b"gemm_ablate", 0)
