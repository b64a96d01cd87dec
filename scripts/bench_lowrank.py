#!/usr/bin/env python3
"""Timings of the attention-site adapter kernels at ViT-B/32, B=128 (delta_add, lowrank_u, lowrank_grad)."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
from pevit_amd import _lib
lib = _lib.load()
P = lambda t: None if t is None else C.c_void_p(t.data_ptr())
S = lambda: C.c_void_p(torch.cuda.current_stream().cuda_stream)
def timeit(fn, iters=30):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
B, H, N, E = 128, 12, 50, 768
T = B * N; ld = 3 * E + 64
q = torch.randn(T * E, device="cuda").bfloat16(); v = torch.randn(T * E, device="cuda").bfloat16()
t = torch.randn(T, 64, device="cuda"); q32 = torch.randn(E, 64, device="cuda"); bias = torch.randn(E, device="cuda")
print(f"delta_add    {timeit(lambda: lib.pevit_op_delta_add(S(), P(q), P(v), P(t), P(q32), P(bias), 160.0, B, N, E)):.1f} us")
dqkv = torch.randn(T, ld, device="cuda").bfloat16(); qT = torch.randn(64, E, device="cuda").bfloat16()
u32 = torch.zeros(T, 64, device="cuda")
print(f"lowrank_u    {timeit(lambda: lib.pevit_op_lowrank_u(S(), P(dqkv), ld, P(qT), P(u32), C.c_void_p(dqkv.data_ptr() + 3 * E * 2), B, H, N, E)):.1f} us")
xn = torch.randn(T, E, device="cuda").bfloat16()
chunks = lib.pevit_op_lowrank_chunks(T)
partial = torch.zeros(chunks * 4 * E * 32 + 1024, device="cuda"); dbias = torch.zeros(chunks * 2 * E + 1024, device="cuda")
print(f"lowrank_grad {timeit(lambda: lib.pevit_op_lowrank_grad(S(), P(xn), E, P(u32), P(dqkv), ld, P(t), P(partial), P(dbias), B, H, N, E)):.1f} us")
