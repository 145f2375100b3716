"""Which library kernels torch.matmul picks on the tile-encoder GEMM shapes (run under rocprofv3 --kernel-trace --stats)."""
import torch
M = 131070
for N, K in ((4096, 1024), (3072, 1024), (1024, 1024), (1024, 4096)):
    a = torch.randn(M, K, device="cuda").half()
    w = torch.randn(N, K, device="cuda").half()
    for _ in range(5):
        torch.matmul(a, w.t())
    torch.cuda.synchronize()
