#!/usr/bin/env python3
"""N = E products on the phased k-split kernel: LDS-DMA requests in the LOAD section (gemm_kphase_nl=8) vs between the MFMAs
(2: two in LOAD, the rest behind every second MFMA; 0: all between the MFMAs); gemm_ksplit_stagger=1 = the alternate-k-tile
kernel.  Interleaved in one process."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import bench_gemm as bg
E = 768
def shapes(M):
    return [("out_proj fwd", "BIAS_RESID", M, E, E), ("c_proj fwd", "BIAS_RESID", M, E, 4 * E), ("c_fc bwd", "F32", M, E, 4 * E),
            ("out_proj bwd", "BF16", M, E, E), ("qkv bwd (+u)", "F32", M, E, 3 * E + 64)]
MODES = [(1, 8)] + [(2, nl) for nl in (8, 2, 0)]
for M in [int(m) for m in (sys.argv[1:] or ["6400", "3200"])]:
    for rnd in range(2):
        for mode, nl in MODES:
            bg.tune("gemm_ksplit_stagger", mode); bg.tune("gemm_kphase_nl", nl)
            print(f"---- M={M} ksplit_stagger={mode} kphase_nl={nl} round {rnd}")
            for name, epi, m, n, k in shapes(M):
                bg.run(name, bg.EPI[epi], m, n, k, E, 12, 50, iters=30)
bg.tune("gemm_ksplit_stagger", 2); bg.tune("gemm_kphase_nl", 8)
