import torch
for M, N, K in ((6400, 768, 3072), (6400, 768, 2368), (6400, 768, 768), (6400, 3072, 768)):
    A = torch.randn(M, K, device="cuda").bfloat16(); B = (torch.randn(N, K, device="cuda") * 0.05).bfloat16()
    for _ in range(10): torch.matmul(A, B.T)
    torch.cuda.synchronize()
