#!/usr/bin/env python3
"""Large square / large-M problems: two-stage 128x128 kernel vs the 256x256 register-resident kernel."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import bench_gemm as bg
EPI = bg.EPI
for mode, ab, name in ((0, 0, "128x128 two-stage"), (2, 0, "256x256"), (2, 32, "256x256 + setprio"), (0, 0, "128x128 two-stage"), (2, 0, "256x256"), (2, 32, "256x256 + setprio")):
    bg.lib.pevit_tune(b"gemm_256", mode); bg.lib.pevit_tune(b"gemm_ablate", ab)
    print("====", name)
    for (M, N, K) in ((4096, 4096, 4096), (8192, 8192, 4096), (12800, 3072, 768), (12800, 2304, 768), (8224, 3072, 1024), (8224, 4096, 1024), (8224, 1024, 4096)):
        bg.run(f"{M}x{N}x{K}", EPI["BF16"], M, N, K, iters=10)
bg.lib.pevit_tune(b"gemm_256", 0); bg.lib.pevit_tune(b"gemm_ablate", 0)
