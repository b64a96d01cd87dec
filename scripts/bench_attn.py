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
q = (torch.randn(B * H, N, 64, device="cuda") * 0.3).bfloat16(); k = torch.randn(B * H, N, 64, device="cuda").bfloat16(); v = torch.randn(B * H, N, 64, device="cuda").bfloat16()
out = torch.zeros(B * N, E, dtype=torch.bfloat16, device="cuda"); lse = torch.zeros(B * H, N, device="cuda")
do = torch.randn(B * N, E, device="cuda").bfloat16(); ld = 3 * E + 64
dqkv = torch.zeros(B * N, ld, dtype=torch.bfloat16, device="cuda")
fwd = lambda: lib.pevit_op_attn_fwd(S(), P(q), P(k), P(v), P(out), E, P(lse), B, H, N)
bwd = lambda: lib.pevit_op_attn_bwd(S(), P(q), P(k), P(v), P(out), E, P(do), E, P(lse), P(dqkv), ld, B, H, N)
print(f"attn fwd {timeit(fwd):.1f} us")
print(f"attn bwd {timeit(bwd):.1f} us")
