import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
from pevit_amd import _lib
lib = _lib.load()
P = lambda t: None if t is None else C.c_void_p(t.data_ptr())
S = lambda: C.c_void_p(torch.cuda.current_stream().cuda_stream)
def run(M, N, K, cfg, iters):
    lib.pevit_tune(None, b"gemm_config", cfg)
    A = torch.randn(M, K, device="cuda").bfloat16(); B = (torch.randn((N + 127) // 128 * 128, K, device="cuda") * 0.05).bfloat16()
    out = torch.empty(M, N, dtype=torch.bfloat16, device="cuda")
    for _ in range(iters):
        assert lib.pevit_op_gemm(S(), 5, P(A), K, P(B), K, B.shape[0], M, N, K, None, None, 0, None, 0, P(out), N, None, 0, None, 0, 0, 0, 0, 0) == 0
    torch.cuda.synchronize()
run(6400, 3072, 768, 0, 10)
run(6400, 768, 3072, 2, 10)
run(4096, 4096, 4096, 0, 4)
run(6400, 3072, 768, 3, 10)
