"""GEMM timing sweep over M (working-set size) for one tile config."""
import sys
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from stamp_amd import _lib, ops
from tools.bench_kernels import timeit
dev = torch.device("cuda:0"); dt = torch.float16
cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 3
g = torch.Generator().manual_seed(0)
for (N, K, epi, name) in ((4096, 1024, _lib.EPI_BIAS, "N4096,K1024,bias"), (1024, 4096, _lib.EPI_BIAS, "N1024,K4096,bias"), (1024, 4096, _lib.EPI_RESIDUAL, "N1024,K4096,resid")):
    w = (torch.randn(N, K, generator=g) / K ** 0.5).to(dev, dt)
    bias = torch.randn(N, generator=g).to(dev)
    for M in (4096, 8192, 16384, 32768, 65536, 131072):
        A = torch.randn(M, K, generator=g).to(dev, dt)
        out = torch.zeros(M, N, device=dev) if epi == _lib.EPI_RESIDUAL else torch.empty(M, N, dtype=dt, device=dev)
        t = timeit(lambda: ops.gemm(A, w, epi, bias=bias, out=out, cfg=cfg), iters=8)
        print(f"cfg={cfg} {name:18s} M={M:7d} blocks={((M+255)//256)*(N//256):5d}: {t*1e6:8.1f} us {2*M*N*K/t/1e12:7.1f} TF/s", flush=True)
