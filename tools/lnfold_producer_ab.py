"""proj / fc2 shapes: plain RESIDUAL epilogue against the LayerNorm-fold producer form (16-bit row copy + partial row sums), sustained.
python tools/lnfold_producer_ab.py"""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from stamp_amd import _lib, ops  # noqa: E402

M = 262140
for name, N, K in (("proj", 1024, 1024), ("fc2", 1024, 4096)):
    a = torch.randn(M, K, device="cuda").half()
    w = (torch.randn(N, K, device="cuda") / K ** 0.5).half()
    b = torch.zeros(N, device="cuda")
    sc = torch.full((N,), 0.1, device="cuda")
    x = torch.zeros(M, N, device="cuda")
    xh = torch.empty(M, N, device="cuda", dtype=torch.float16)
    rp = torch.empty(M, N // 128, 2, device="cuda")
    res = {}
    for rnd in range(2):
        for tag, fn in (("plain", lambda: ops.gemm(a, w, _lib.EPI_RESIDUAL, bias=b, scale=sc, out=x, cfg=12)),
                        ("fold", lambda: ops.gemm_lnfold(a, w, _lib.EPI_RESIDUAL, out=x, bias=b, scale=sc, xh=xh, rowpart=rp))):
            for _ in range(max(4, int(0.4 / (2.0 * M * N * K / 1.0e15)))):
                fn()
            t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            t0.record()
            for _ in range(40):
                fn()
            t1.record(); torch.cuda.synchronize()
            res[tag] = t0.elapsed_time(t1) / 40 * 1e3
        print(f"{name} N={N} K={K}: plain {res['plain']:7.1f} us, producer form {res['fold']:7.1f} us (+{res['fold'] - res['plain']:.0f})", flush=True)
