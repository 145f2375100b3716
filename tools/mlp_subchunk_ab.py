"""Does the 256 MB Infinity Cache help if fc1 -> fc2 run over row sub-chunks whose hidden activations fit it?
fc1 (+GELU, 16-bit out [M, 4096]) then fc2 (+bias, fp32 residual read-modify-write [M, 1024]) of ViT-L over M = 262144 rows:
one pair of launches against S pairs over M / S rows.  python tools/mlp_subchunk_ab.py"""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from stamp_amd import _lib, ops  # noqa: E402

M, D, F = 262144, 1024, 4096
g = torch.Generator().manual_seed(0)
x16 = (torch.randn(M, D, generator=g) * 0.5).to("cuda", torch.float16)
w1 = (torch.randn(F, D, generator=g) * 0.02).to("cuda", torch.float16)
w2 = (torch.randn(D, F, generator=g) * 0.02).to("cuda", torch.float16)
b1 = torch.zeros(F, device="cuda")
b2 = torch.zeros(D, device="cuda")
h = torch.empty(M, F, device="cuda", dtype=torch.float16)
res = torch.zeros(M, D, device="cuda")


def run(S, reuse=False):
    rows = M // S
    for i in range(S):
        sl = slice(i * rows, (i + 1) * rows)
        hs = h[:rows] if reuse else h[sl]          # reuse: every sub-chunk's hidden rows live in the SAME buffer (dirty lines overwritten in cache)
        ops.gemm(x16[sl], w1, _lib.EPI_BIAS_GELU, bias=b1, out=hs, cfg=12)
        ops.gemm(hs, w2, _lib.EPI_RESIDUAL, bias=b2, out=res[sl], cfg=12)


def timeit(fn, n=5):
    for _ in range(2):
        fn()
    t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0.record()
    for _ in range(n):
        fn()
    t1.record(); torch.cuda.synchronize()
    return t0.elapsed_time(t1) / n * 1e3


for rep in range(2):
    for S in (1, 8, 12, 16, 24, 32):
        if M % (S * 256):
            continue
        us = timeit(lambda: run(S))
        us2 = timeit(lambda: run(S, True))
        print(f"S={S:2d} ({M // S:6d} rows, hidden {M // S * F * 2 / 2**20:6.0f} MB per sub-chunk): {us:8.1f} us per fc1+fc2 over all rows; one hidden buffer reused: {us2:8.1f} us", flush=True)
