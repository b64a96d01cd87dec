#!/usr/bin/env python3
"""Per-shape timing of the GEMM kernel family on the shapes of one fine-tune step, for every tile
configuration (pevit_tune gemm_config), plus the k-loop-only / epilogue-only ablations.

    python scripts/bench_gemm.py [--arch b32|l14|b16] [--configs -1,0,1,3,4,5] [--ablate]
"""
import argparse
import ctypes as C
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from pevit_amd import _lib

lib = _lib.load()
P = lambda t: None if t is None else C.c_void_p(t.data_ptr())
S = lambda: C.c_void_p(torch.cuda.current_stream().cuda_stream)
EPI = dict(QKV=0, BIAS_RESID=1, BIAS_GELU=2, DGELU=3, F32=4, BF16=5)
NAMES = {-1: "heuristic", 0: "128x128 4w", 1: "64x128 4w", 2: "64x64 4w", 3: "256x128 8w", 4: "256x256 8w", 5: "320x256 8w", 6: "128x64 4w", 7: "128x128 4w pipelined", 8: "64x128 4w pipelined", 9: "160x256 8w (1x8 waves, two tiles per CU)"}


def tune(key, val):
    assert lib.pevit_tune(None, key.encode(), int(val)) == 0


def run(name, epi, M, N, K, E, H, ntok, iters=30):
    A = (torch.randn(M, K, device="cuda") * 1.0).bfloat16()
    Nb = (N + 255) // 256 * 256
    B = (torch.randn(Nb, K, device="cuda") * 0.05).bfloat16()
    bias = torch.randn(max(N, 3 * E), device="cuda")
    resid = torch.randn(M, N, device="cuda")
    outf = torch.empty(M, max(N, 64), device="cuda")
    outb = torch.empty(3 * M * E if epi == 0 else M * N, dtype=torch.bfloat16, device="cuda")
    outb2 = torch.empty(M, N, dtype=torch.bfloat16, device="cuda")
    aux = (torch.randn(M, N, device="cuda")).bfloat16()

    def call():
        rc = lib.pevit_op_gemm(S(), epi, P(A), K, P(B), K, Nb, M, N, K, P(bias), P(resid), N, P(outf), 64 if epi == 0 else N,
                               P(outb), N, P(outb2), N, P(aux), N, M * E, E, H, ntok)
        assert rc == 0, lib.pevit_last_error()
    for _ in range(5):
        call()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        call()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / iters * 1e3
    tf = 2.0 * M * N * K / us / 1e6
    print(f"{name:22s} M={M:5d} N={N:5d} K={K:5d}  {us:8.1f} us  {tf:7.1f} TF", flush=True)
    return us


ARCH = {"b32": (128 * 50, 768, 12, 50, 12), "b16": (64 * 197, 768, 12, 197, 12), "l14": (32 * 257, 1024, 16, 257, 24)}


def shapes(arch):
    M, E, H, ntok, L = ARCH[arch]
    tot = 0
    tot += run("qkv fwd (+t)", EPI["QKV"], M, 3 * E + 64, E, E, H, ntok)
    tot += run("out_proj fwd", EPI["BIAS_RESID"], M, E, E, E, H, ntok)
    tot += run("c_fc fwd (gelu)", EPI["BIAS_GELU"], M, 4 * E, E, E, H, ntok)
    tot += run("c_proj fwd", EPI["BIAS_RESID"], M, E, 4 * E, E, H, ntok)
    tot += run("c_proj bwd (dgelu)", EPI["DGELU"], M, 4 * E, E, E, H, ntok)
    tot += run("c_fc bwd", EPI["F32"], M, E, 4 * E, E, H, ntok)
    tot += run("out_proj bwd", EPI["BF16"], M, E, E, E, H, ntok)
    tot += run("qkv bwd (+u)", EPI["F32"], M, E, 3 * E + 64, E, H, ntok)
    print(f"sum per layer {tot:.1f} us -> x{L} = {tot * L / 1e3:.2f} ms")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--arch", default="b32")
    ap.add_argument("--configs", default="-1,0,1,3,4,5")
    ap.add_argument("--ablate", action="store_true")
    ap.add_argument("--square", action="store_true")
    args = ap.parse_args()
    for cfg in [int(c) for c in args.configs.split(",")]:
        tune("gemm_config", cfg)
        for ab in ((0, 2, 1) if args.ablate else (0,)):
            tune("gemm_ablate", ab)
            print(f"---- gemm_config {cfg} ({NAMES[cfg]})" + {0: "", 2: "  [k-loop only: no stores]", 1: "  [prologue+epilogue only]"}[ab])
            shapes(args.arch)
            if args.square and cfg in (0, 3, 4, 5) and ab == 0:
                run("square 4096", EPI["BF16"], 4096, 4096, 4096, 768, 12, 50, iters=10)
                run("8192x8192x4096", EPI["BF16"], 8192, 8192, 4096, 768, 12, 50, iters=5)
        tune("gemm_ablate", 0)
    tune("gemm_config", -1)


if __name__ == "__main__":
    main()
