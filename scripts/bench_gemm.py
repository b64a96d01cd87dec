#!/usr/bin/env python3
"""Per-shape timing of the GEMM kernel family on the shapes of one ViT-B/32 bs=128 step."""
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


def run(name, epi, M, N, K, iters=30):
    T, E = M, 768
    A = (torch.randn(M, K, device="cuda") * 1.0).bfloat16()
    Nb = (N + 127) // 128 * 128
    B = (torch.randn(Nb, K, device="cuda") * 0.05).bfloat16()
    bias = torch.randn(max(N, 3 * E), device="cuda")
    resid = torch.randn(M, N, device="cuda")
    outf = torch.empty(M, max(N, 64), device="cuda")
    outb = torch.empty(3 * M * E if epi == 0 else M * N, dtype=torch.bfloat16, device="cuda")
    outb2 = torch.empty(M, N, dtype=torch.bfloat16, device="cuda")
    aux = (torch.randn(M, N, device="cuda")).bfloat16()

    def call():
        rc = lib.pevit_op_gemm(S(), epi, P(A), K, P(B), K, Nb, M, N, K, P(bias), P(resid), N, P(outf), 64 if epi == 0 else N,
                               P(outb), N, P(outb2), N, P(aux), N, M * E, E, 12, 50)
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
    print(f"{name:28s} M={M:5d} N={N:5d} K={K:5d}  {us:8.1f} us  {tf:7.1f} TF", flush=True)
    return us


def main():
    names = {-1: "heuristic", 0: "128x128x64", 1: "128x128x32", 2: "128x64x64", 3: "64x64x64", 4: "64x128x64"}
    names.update({5: "128x128x64 hoisted frags", 6: "128x128x64 hoisted + setprio"})
    for persistent in (1,):
        lib.pevit_tune(b"gemm_persistent", persistent)
        for cfg in (0, 5, 6, -1):
            lib.pevit_tune(b"gemm_config", cfg)
            print(f"---- persistent {persistent} gemm_config {cfg} ({names[cfg]})")
            shapes(big=(cfg in (0, 5, 6)))


def shapes(big=False):
    M = 6400
    tot = 0
    tot += run("qkv fwd (+t)", EPI["QKV"], M, 2368, 768)
    tot += run("out_proj fwd", EPI["BIAS_RESID"], M, 768, 768)
    tot += run("c_fc fwd (gelu)", EPI["BIAS_GELU"], M, 3072, 768)
    tot += run("c_proj fwd", EPI["BIAS_RESID"], M, 768, 3072)
    tot += run("c_proj bwd (dgelu)", EPI["DGELU"], M, 3072, 768)
    tot += run("c_fc bwd", EPI["F32"], M, 768, 3072)
    tot += run("out_proj bwd", EPI["BF16"], M, 768, 768)
    tot += run("qkv bwd (+u)", EPI["F32"], M, 768, 2368)
    print(f"sum per layer {tot:.1f} us -> x12 = {tot * 12 / 1e3:.2f} ms")
    if big:
        run("square 4096", EPI["BF16"], 4096, 4096, 4096, iters=10)


if __name__ == "__main__":
    main()
