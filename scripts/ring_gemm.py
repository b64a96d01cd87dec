#!/usr/bin/env python3
"""Default tile selection vs the single-stage 4-workgroups-per-CU variant (config 8) on the step's shapes."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import bench_gemm as bg
bg.lib.pevit_tune(b"gemm_ring", 0)
for cfg in (-1, 8, -1, 8):
    bg.lib.pevit_tune(b"gemm_config", cfg)
    print(f"==== config {cfg}")
    bg.shapes(big=True)
