"""A/B of the production GEMM's LDS-DMA schedules (kernel ids 12 / 13) on the four ViT-L shapes at one 1020-tile chunk, sustained:
every candidate is warmed for ~0.4 s, then 40 launches are timed between two events; two rounds.  python tools/gemm_sched_ab.py"""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from stamp_amd import _lib, ops  # noqa: E402

M = int(sys.argv[1]) if len(sys.argv) > 1 else 262140
CFGS = tuple(int(c) for c in sys.argv[2].split(",")) if len(sys.argv) > 2 else (12, 13)
shapes = (("qkv", 3072, 1024, _lib.EPI_BIAS), ("proj", 1024, 1024, _lib.EPI_RESIDUAL), ("fc1", 4096, 1024, _lib.EPI_BIAS_GELU),
          ("fc2", 1024, 4096, _lib.EPI_RESIDUAL))
for rnd in range(2):
    for name, N, K, epi in shapes:
        a = torch.randn(M, K, device="cuda").half()
        w = (torch.randn(N, K, device="cuda") / K ** 0.5).half()
        b = torch.zeros(N, device="cuda")
        sc = torch.full((N,), 0.1, device="cuda") if epi == _lib.EPI_RESIDUAL else None
        out = torch.zeros(M, N, device="cuda") if epi == _lib.EPI_RESIDUAL else None
        line = f"{name:5s} N={N} K={K}:"
        for cfg in CFGS:
            t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
            n_warm = max(4, int(0.4 / (2.0 * M * N * K / 1.0e15)))
            for _ in range(n_warm):
                ops.gemm(a, w, epi, bias=b, scale=sc, out=out, cfg=cfg)
            t0.record()
            for _ in range(40):
                ops.gemm(a, w, epi, bias=b, scale=sc, out=out, cfg=cfg)
            t1.record(); torch.cuda.synchronize()
            us = t0.elapsed_time(t1) / 40 * 1e3
            line += f"  cfg {cfg}: {us:7.1f} us {2.0 * M * N * K / us / 1e6:6.0f} TF/s"
        print(line, flush=True)
        del a, w, out
