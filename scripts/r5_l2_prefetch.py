#!/usr/bin/env python3
"""Feasibility probe (round 5): does touching the NEXT product's weight panel from a side stream, while the current product runs,
shorten the products?  (A k-tile costs rows x mean latency / 64 clocks and a quarter of the requests are first touches of a panel by
an XCD's private L2: profiles/NOTES_gemm.md.  The weights of the next product are known in advance; its activations are not.)
Rotation of ROT operand sets (> 256 MB in all), this build's production kernels through pevit_op_gemm; the touch kernel is
scripts/probes/l2_touch.hip (every XCD pulls the whole buffer; per_xcd workgroups per XCD).
usage: python scripts/r5_l2_prefetch.py [rot] > table.md"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from pevit_amd import _lib

lib = _lib.load()
probe = C.CDLL(os.path.join(ROOT, "scripts", "probes", "libl2touch.so"))
probe.probe_l2_touch.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p]
P = lambda t: None if t is None else C.c_void_p(t.data_ptr())
EPI = dict(QKV=0, BIAS_RESID=1, BIAS_GELU=2, DGELU=3, F32=4, BF16=5)
ROT = int(sys.argv[1]) if len(sys.argv) > 1 else 12
E, H, NTOK = 768, 12, 50
main = torch.cuda.current_stream()
side = torch.cuda.Stream()
sink = torch.zeros(4, dtype=torch.int32, device="cuda")


class Bufs:
    def __init__(self, epi, M, N, K):
        self.A = torch.randn(M, K, device="cuda").bfloat16()
        self.Nb = (N + 255) // 256 * 256
        self.B = (torch.randn(self.Nb, K, device="cuda") * 0.05).bfloat16()
        self.bias = torch.randn(max(N, 3 * E), device="cuda")
        self.resid = torch.randn(M, N, device="cuda") if epi == 1 else None
        self.outf = torch.empty(M, max(N, 64), device="cuda") if epi in (0, 1, 4) else None
        self.outb = torch.empty(3 * M * E if epi == 0 else M * N, dtype=torch.bfloat16, device="cuda")
        self.outb2 = torch.empty(M, N, dtype=torch.bfloat16, device="cuda") if epi == 2 else None
        self.aux = torch.randn(M, N, device="cuda").bfloat16() if epi == 3 else None


def ours(b, epi, M, N, K):
    rc = lib.pevit_op_gemm(C.c_void_p(main.cuda_stream), epi, P(b.A), K, P(b.B), K, b.Nb, M, N, K, P(b.bias), P(b.resid), N, P(b.outf),
                           64 if epi == 0 else N, P(b.outb), N, P(b.outb2), N, P(b.aux), N, M * E, E, H, NTOK)
    assert rc == 0, lib.pevit_last_error()


def touch(t, per_xcd):
    assert probe.probe_l2_touch(C.c_void_p(side.cuda_stream), P(t), t.numel() * t.element_size(), per_xcd, P(sink)) == 0


def run(sets, epi, M, N, K, what, per_xcd, reps):
    def loop():
        for i, s in enumerate(sets):
            if what:
                nxt = sets[(i + 1) % len(sets)]
                ev = torch.cuda.Event(); ev.record(main); side.wait_event(ev)      # the touch starts when this product starts
                if "0" in what: touch(sink, 1)                                     # control: the event traffic and an empty launch
                if "B" in what: touch(nxt.B[:N], per_xcd)
                if "A" in what: touch(nxt.A, per_xcd)
            ours(s, epi, M, N, K)
    loop(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(main)
    for _ in range(reps): loop()
    e1.record(main); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / (reps * len(sets)) * 1e3


SH = [("out_proj fwd", "BIAS_RESID", 6400, 768, 768), ("c_proj fwd", "BIAS_RESID", 6400, 768, 3072), ("c_fc bwd", "BF16", 6400, 768, 3072),
      ("c_fc fwd (gelu)", "BIAS_GELU", 6400, 3072, 768), ("qkv fwd (+t)", "QKV", 6400, 2368, 768)]
print("| product | M x N x K | no touch | control: events + a 16-byte touch | next B, 1 / XCD | next B, 2 / XCD | next B, 4 / XCD | next A and B, 4 / XCD (upper bound: A is not known ahead in the step) |")
print("|---|---|---|---|---|---|---|---|")
for name, epin, M, N, K in SH:
    epi = EPI[epin]
    sets = [Bufs(epi, M, N, K) for _ in range(ROT)]
    r = [run(sets, epi, M, N, K, "", 0, 6)]
    for what, px in (("0", 1), ("B", 1), ("B", 2), ("B", 4), ("AB", 4)):
        r.append(run(sets, epi, M, N, K, what, px, 6))
    print(f"| {name} | {M} x {N} x {K} | " + " | ".join(f"{x:.1f}" for x in r) + " |", flush=True)
    del sets
    torch.cuda.empty_cache()
