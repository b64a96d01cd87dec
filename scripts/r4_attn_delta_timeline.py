#!/usr/bin/env python3
"""Phase timeline of attn_fwd_delta_kernel (s_memtime per workgroup) + back-to-back timing against delta_add + attn_fwd."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pevit_amd import _lib
lib = _lib.load()
P = lambda t: None if t is None else C.c_void_p(t.data_ptr())
S = lambda: C.c_void_p(torch.cuda.current_stream().cuda_stream)
B, H, N, E = int(sys.argv[1]) if len(sys.argv) > 1 else 128, 12, 50, 768
T = B * N
g = torch.Generator(device="cuda").manual_seed(0)
q = (torch.randn(B * H, N, 64, device="cuda", generator=g) * 0.35).bfloat16(); k = torch.randn(B * H, N, 64, device="cuda", generator=g).bfloat16()
v = torch.randn(B * H, N, 64, device="cuda", generator=g).bfloat16()
t = torch.randn(T, 64, device="cuda", generator=g); q32 = (torch.randn(E, 64, device="cuda", generator=g) * 0.3).bfloat16(); bias = torch.randn(E, device="cuda", generator=g) * 0.2
out = torch.zeros(T, E, dtype=torch.bfloat16, device="cuda"); lse = torch.zeros(B * H, N, device="cuda")
big = torch.empty(256 << 20, dtype=torch.uint8, device="cuda")      # flush L2 / MALL between launches

def fused(): assert lib.pevit_op_attn_fwd_delta(S(), P(q), P(k), P(v), P(t), P(q32), P(bias), 1e-3, P(out), E, P(lse), B, H, N) == 0
def two():
    assert lib.pevit_op_delta_add(S(), P(q), P(v), P(t), P(q32), P(bias), 1e-3, B, N, E) == 0
    assert lib.pevit_op_attn_fwd(S(), P(q), P(k), P(v), P(out), E, P(lse), B, H, N) == 0
for name, fn in (("fused", fused), ("delta_add + attn_fwd", two)):
    for _ in range(3): fn()
    ts = []
    for _ in range(20):
        big.zero_()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1) * 1e3)
    ts.sort(); print(f"{name:24s} median {ts[len(ts)//2]:.1f} us  min {ts[0]:.1f}")
nwg = (B * H + 5) // 6
tl = torch.zeros(nwg * 8, dtype=torch.int64, device="cuda")
lib.pevit_debug_timeline(P(tl))
big.zero_(); fused(); torch.cuda.synchronize()
lib.pevit_debug_timeline(None)
x = tl.view(nwg, 8).cpu().double()
names = ["start", "loads landed", "LDS staged+barrier", "delta done (wave 0)", "barrier", "q'v' stores issued", "attention done", "stores drained"]
print("s_memtime ticks per workgroup relative to ITS OWN start (the counters of the XCDs have different bases); 100 MHz -> us = ticks / 100")
for i, n in enumerate(names):
    c = x[:, i] - x[:, 0]
    print(f"  {n:24s} mean {c.mean() / 100:.2f} us   min {c.min() / 100:.2f}   max {c.max() / 100:.2f}")
# workgroups of one XCD share a counter: spread of their start / end stamps
for xcd in range(8):
    w = x[xcd::8]
    print(f"  XCD {xcd}: first start -> last start {(w[:, 0].max() - w[:, 0].min()) / 100:.2f} us, first start -> last end {(w[:, 7].max() - w[:, 0].min()) / 100:.2f} us")
