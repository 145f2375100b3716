"""time(K) = a + b K for the production GEMM at the tile-encoder shapes: a = fixed per-launch cost (prologue, epilogue, tail),
1/b = asymptotic MFMA rate.  python tools/gemm_k_sweep.py"""
import sys
import time
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from stamp_amd import _lib, ops  # noqa: E402

M = 131070
CFG = int(sys.argv[1]) if len(sys.argv) > 1 else -1
for N, epi, name in ((4096, _lib.EPI_BIAS_GELU, "fc1+gelu"), (3072, _lib.EPI_BIAS, "qkv"), (1024, _lib.EPI_RESIDUAL, "residual")):
    rows = []
    for K in (256, 512, 1024, 2048, 4096):
        a = torch.randn(M, K, device="cuda").half()
        w = (torch.randn(N, K, device="cuda") / K ** 0.5).half()
        b = torch.zeros(N, device="cuda")
        out = torch.zeros(M, N, device="cuda") if epi == _lib.EPI_RESIDUAL else None
        for _ in range(3):
            ops.gemm(a, w, epi, bias=b, out=out, cfg=CFG)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10):
            ops.gemm(a, w, epi, bias=b, out=out, cfg=CFG)
        torch.cuda.synchronize()
        us = (time.perf_counter() - t0) / 10 * 1e6
        rows.append((K, us))
        del a, w
    (k1, t1), (k2, t2) = rows[1], rows[-1]
    slope = (t2 - t1) / (k2 - k1)
    icpt = t1 - slope * k1
    print(name, " ".join(f"K={k}:{t:.0f}us({2.0 * M * N * k / t / 1e6:.0f}TF)" for k, t in rows),
          f"| fixed {icpt:.0f} us, asymptotic {2.0 * M * N / slope / 1e6:.0f} TF/s")
