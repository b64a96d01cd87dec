#!/usr/bin/env python3
"""Two-stage kernel vs the 160x128x64 ring on the step's shapes (ring 1 = automatic selection)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import bench_gemm as bg
bg.lib.pevit_tune(b"gemm_config", -1)
for ring in (0, 1, 0, 1):
    bg.lib.pevit_tune(b"gemm_ring", ring)
    print(f"==== ring {ring}")
    bg.shapes()
