"""GEMM shapes of the MIL `vit` training step (M = 64 bags x 1025 tokens) across kernel ids:  python tools/gemm_mil_shapes.py"""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from stamp_amd import _lib, ops  # noqa: E402

M = 65600
for N, K, epi, name in ((512, 1024, _lib.EPI_BIAS, "project"), (1536, 512, _lib.EPI_BIAS, "qkv"), (512, 512, _lib.EPI_RESIDUAL, "out"), (2048, 512, _lib.EPI_BIAS, "fc1"),
                        (512, 2048, _lib.EPI_BIAS_F32, "fc2"), (512, 1536, _lib.EPI_BIAS_F32, "dqkv W"), (1024, 512, _lib.EPI_BIAS_F32, "dbags")):
    a = torch.randn(M, K, generator=torch.Generator().manual_seed(0)).to("cuda", torch.bfloat16)
    w = (torch.randn(N, K, generator=torch.Generator().manual_seed(1)) * 0.05).to("cuda", torch.bfloat16)
    b = torch.zeros(N, device="cuda")
    f32 = epi in (_lib.EPI_RESIDUAL, _lib.EPI_BIAS_F32)
    out = torch.zeros(M, N, device="cuda", dtype=torch.float32 if f32 else torch.bfloat16)
    res = []
    for cfg in (0, 8, 10, 12, -2):
        fn = lambda: ops.gemm(a, w, epi, bias=b, out=out, cfg=cfg)  # noqa: E731
        for _ in range(3):
            fn()
        t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0.record()
        for _ in range(20):
            fn()
        t1.record(); torch.cuda.synchronize()
        us = t0.elapsed_time(t1) / 20 * 1e3
        res.append(f"cfg {cfg:2d}: {us:6.1f} us {2.0 * M * N * K / us / 1e6:5.0f} TF/s")
    print(f"{name:8s} N={N:4d} K={K:4d}: " + " | ".join(res), flush=True)
