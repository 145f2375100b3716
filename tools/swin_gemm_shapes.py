"""The stage-3 / stage-4 GEMM shapes of CTransPath (1024 tiles per chunk) under different kernel ids, round-robin, best of 3.
python tools/swin_gemm_shapes.py"""
import sys
import time
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from stamp_amd import _lib, ops  # noqa: E402


def t_of(fn, n=10):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e6


shapes = [("s3 qkv", 200704, 1152, 384, _lib.EPI_BIAS), ("s3 proj", 200704, 384, 384, _lib.EPI_RESIDUAL), ("s3 fc1", 200704, 1536, 384, _lib.EPI_BIAS_GELU),
          ("s3 fc2", 200704, 384, 1536, _lib.EPI_RESIDUAL), ("s4 qkv", 50176, 2304, 768, _lib.EPI_BIAS), ("s4 proj", 50176, 768, 768, _lib.EPI_RESIDUAL),
          ("s4 fc1", 50176, 3072, 768, _lib.EPI_BIAS_GELU), ("s4 fc2", 50176, 768, 3072, _lib.EPI_RESIDUAL), ("merge2", 50176, 768, 1536, _lib.EPI_BIAS_F32)]
for name, M, N, K, epi in shapes:
    a = torch.randn(M, K, device="cuda").half()
    w = (torch.randn(N, K, device="cuda") / K ** 0.5).half()
    b = torch.zeros(N, device="cuda")
    out = torch.zeros(M, N, device="cuda") if epi == _lib.EPI_RESIDUAL else None
    best = {}
    for _ in range(3):
        for cfg in (-1, 0, 9, 8, 10):
            best[cfg] = min(best.get(cfg, 1e9), t_of(lambda: ops.gemm(a, w, epi, bias=b, out=out, cfg=cfg)))
    print(f"{name:8s} M={M} N={N} K={K}: " + "  ".join(f"cfg {c}: {t:6.0f} us" for c, t in best.items()) + f"   ({2.0 * M * N * K / best[-1] / 1e6:.0f} TF/s default)")
