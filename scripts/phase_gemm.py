#!/usr/bin/env python3
"""Does de-phasing co-resident workgroups (so that one's epilogue overlaps the other's k-loop) help?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import bench_gemm as bg
bg.lib.pevit_tune(b"gemm_config", -1)
bg.lib.pevit_tune(b"gemm_ring", 0)
for mode, name in ((0, "in phase"), (32, "second half of the grid late"), (64, "odd local index late")):
    for units in ((0,) if mode == 0 else (4, 8, 16, 24)):
        bg.lib.pevit_tune(b"gemm_ablate", mode | (units << 8))
        print(f"==== {name}, delay {units} x 512 clk")
        bg.shapes()
bg.lib.pevit_tune(b"gemm_ablate", 0)
