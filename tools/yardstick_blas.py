"""Yardstick only (NOT on the product path): what the vendor BLAS reaches on the tile encoder's GEMM shapes."""
import sys
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from tools.bench_kernels import timeit
dev = torch.device("cuda:0")
M = 65535
g = torch.Generator().manual_seed(0)
for name, N, K in (("qkv", 3072, 1024), ("proj", 1024, 1024), ("fc1", 4096, 1024), ("fc2", 1024, 4096)):
    for dt in (torch.float16, torch.bfloat16):
        A = torch.randn(M, K, generator=g).to(dev, dt); W = (torch.randn(N, K, generator=g) / K ** 0.5).to(dev, dt)
        out = torch.empty(M, N, dtype=dt, device=dev)
        t = timeit(lambda: torch.mm(A, W.t(), out=out), iters=10, warm=5)
        print(f"vendor BLAS {name:5s} {str(dt):15s} M={M} N={N} K={K}: {t*1e6:8.1f} us {2*M*N*K/t/1e12:7.1f} TF/s", flush=True)
