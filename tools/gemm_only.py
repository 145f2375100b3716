"""Runs one GEMM shape a few times (for rocprofv3 --pmc passes)."""
import sys
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from stamp_amd import _lib, ops
name, cfg = sys.argv[1], int(sys.argv[2])
B = int(sys.argv[3]) if len(sys.argv) > 3 else 128
dev = torch.device("cuda:0"); dt = torch.float16
M, D, Hd = B * 257, 1024, 4096
N, K, epi = {"qkv": (3 * D, D, _lib.EPI_BIAS), "proj": (D, D, _lib.EPI_RESIDUAL), "fc1": (Hd, D, _lib.EPI_BIAS_GELU), "fc2": (D, Hd, _lib.EPI_RESIDUAL)}[name]
g = torch.Generator().manual_seed(0)
A = torch.randn(M, K, generator=g).to(dev, dt); w = (torch.randn(N, K, generator=g) / K ** 0.5).to(dev, dt)
bias = torch.randn(N, generator=g).to(dev)
out = torch.zeros(M, N, device=dev) if epi == _lib.EPI_RESIDUAL else torch.empty(M, N, dtype=dt, device=dev)
for _ in range(5):
    ops.gemm(A, w, epi, bias=bias, out=out, cfg=cfg)
torch.cuda.synchronize()
