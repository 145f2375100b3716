"""Reference point for the GEMM rate: torch.matmul (hipBLASLt: Custom_Cijk_..._MT256x256x64_MI16x16x1, 256 persistent 4-wave
workgroups, stream-K) vs amds_gemm on the tile-encoder shapes.  Each candidate is warmed for ~0.7 s (sustained clocks) and
then timed with events over 100 launches.  python tools/vendor_gemm_ref.py"""
import sys
import time
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from stamp_amd import _lib, ops  # noqa: E402


def timed(fn, warm_s=0.7, n=100):
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < warm_s:
        for _ in range(20):
            fn()
        torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


M = 131070
for name, N, K in (("fc1", 4096, 1024), ("qkv", 3072, 1024), ("proj", 1024, 1024), ("fc2", 1024, 4096), ("big-K", 4096, 4096)):
    a = torch.randn(M, K, device="cuda").half()
    w = (torch.randn(N, K, device="cuda") / K ** 0.5).half()
    b = torch.zeros(N, device="cuda")
    bh = b.half()
    res = torch.zeros(M, N, device="cuda")
    flop = 2.0 * M * N * K
    rows = [("torch.matmul f16 out", lambda: torch.matmul(a, w.t())),
            ("F.linear + bias", lambda: torch.nn.functional.linear(a, w, bh)),
            ("amds cfg 8 bias f16", lambda: ops.gemm(a, w, _lib.EPI_BIAS, bias=b, cfg=8)),
            ("amds cfg 8 bias+gelu", lambda: ops.gemm(a, w, _lib.EPI_BIAS_GELU, bias=b, cfg=8)),
            ("amds cfg 8 residual f32", lambda: ops.gemm(a, w, _lib.EPI_RESIDUAL, bias=b, out=res, cfg=8)),
            ("amds cfg 10 bias f16", lambda: ops.gemm(a, w, _lib.EPI_BIAS, bias=b, cfg=10)),
            ("amds cfg 10 bias+gelu", lambda: ops.gemm(a, w, _lib.EPI_BIAS_GELU, bias=b, cfg=10)),
            ("amds cfg 10 residual f32", lambda: ops.gemm(a, w, _lib.EPI_RESIDUAL, bias=b, out=res, cfg=10))]
    for label, fn in rows:
        us = timed(fn)
        print(f"{name:6s} {label:26s} {us:7.0f} us {flop / us / 1e6:6.0f} TF/s")
    del a, w, res
