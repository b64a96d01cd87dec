#!/usr/bin/env python3
"""N = E products: alternate-k-tile k-split kernel (gemm_ksplit_stagger=1) vs the phased one (=2), interleaved in one process."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import bench_gemm as bg
E = 768
def shapes(M):
    return [("out_proj fwd", "BIAS_RESID", M, E, E), ("c_proj fwd", "BIAS_RESID", M, E, 4 * E), ("c_fc bwd", "F32", M, E, 4 * E),
            ("out_proj bwd", "BF16", M, E, E), ("qkv bwd (+u)", "F32", M, E, 3 * E + 64)]
for M in (6400, 3200):
    for rnd in range(2):
        for mode in (1, 2):
            bg.tune("gemm_ksplit_stagger", mode)
            print(f"---- M={M} ksplit_stagger={mode} round {rnd}")
            for name, epi, m, n, k in shapes(M):
                bg.run(name, bg.EPI[epi], m, n, k, E, 12, 50, iters=30)
bg.tune("gemm_ksplit_stagger", 2)
