"""Round 4 overlap probe (library built with `make PROBE=1`): what would an epilogue's HBM traffic cost if it were perfectly overlapped with the K loop?
Kernel ids 21 / 22 / 23 are the production gemm_4w16 kernel (id 12), real epilogue included, whose K loop additionally issues (0, 2) / (1, 1) / (4, 4)
16-byte (loads, stores) per lane and K tile against a scratch region -- exactly one more epilogue's worth of bytes for the qkv / fc1, fc2 and proj shapes --
spread behind the MFMAs and waited for with a counted vmcnt (they stay in flight across the operand wait).  T(probe) - T(12) is the price of those bytes
when nothing but the memory system has to carry them; `additive` is what the same bytes cost as un-overlapped epilogue traffic (bytes / 5.4 TB/s, the
law of profiles/r02_gemm_schedules.txt).  python tools/gemm_overlap_probe.py [M]"""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from stamp_amd import _lib, ops  # noqa: E402

M = int(sys.argv[1]) if len(sys.argv) > 1 else 262140
shapes = (("qkv", 3072, 1024, _lib.EPI_BIAS, 21, 2), ("proj", 1024, 1024, _lib.EPI_RESIDUAL, 23, 8), ("fc1", 4096, 1024, _lib.EPI_BIAS_GELU, 21, 2),
          ("fc2", 1024, 4096, _lib.EPI_RESIDUAL, 22, 8))


def timed(fn, flop):
    n_warm = max(4, int(0.4 / (flop / 1.0e15)))
    for _ in range(n_warm):
        fn()
    t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
    t0.record()
    for _ in range(40):
        fn()
    t1.record(); torch.cuda.synchronize()
    return t0.elapsed_time(t1) / 40 * 1e3


for rnd in range(3):
    for name, N, K, epi, pcfg, bpe in shapes:
        a = torch.randn(M, K, device="cuda").half()
        w = (torch.randn(N, K, device="cuda") / K ** 0.5).half()
        b = torch.randn(N, device="cuda") * 0.1
        res = epi == _lib.EPI_RESIDUAL
        sc = torch.full((N,), 0.1, device="cuda") if res else None
        flop = 2.0 * M * N * K
        outs = {}
        ts = {}
        for cfg in (12, pcfg):
            out = torch.zeros(M, N, device="cuda") if res else None
            if rnd == 0:      # same bits with and without the probe traffic (one launch on a zeroed residual / fresh output)
                o = ops.gemm(a, w, epi, bias=b, scale=sc, out=out, cfg=cfg)
                outs[cfg] = (out if res else o).clone()
            ts[cfg] = timed(lambda: ops.gemm(a, w, epi, bias=b, scale=sc, out=out, cfg=cfg), flop)
            del out
        extra_gb = M * N * bpe / 1e9
        same = "" if rnd else f"  bit-identical: {bool(torch.equal(outs[12], outs[pcfg]))}"
        print(f"{name:5s} N={N} K={K}: id 12 {ts[12]:7.1f} us | id {pcfg} (+{extra_gb:.2f} GB in the K loop) {ts[pcfg]:7.1f} us  delta {ts[pcfg] - ts[12]:+6.1f} us"
              f" | additive law {extra_gb / 5.4e3 * 1e6:5.0f} us{same}", flush=True)
        del a, w, outs
