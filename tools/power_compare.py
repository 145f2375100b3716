"""Socket power and shader clock (rocm-smi, sampled from a thread while the queue stays full) of the vendor GEMM and of the
library's kernels on one shape, 4 s each.  python tools/power_compare.py [N K]"""
import subprocess
import sys
import threading
import time
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from stamp_amd import _lib, ops  # noqa: E402

M = 131070
N, K = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (3072, 1024)
a = torch.randn(M, K, device="cuda").half()
w = (torch.randn(N, K, device="cuda") / K ** 0.5).half()
b = torch.zeros(N, device="cuda")
res = torch.zeros(M, N, device="cuda")
stop = False
samples = []


def sampler():
    while not stop:
        r = subprocess.run(["rocm-smi", "--showclocks", "--showpower", "--csv"], capture_output=True, text=True)
        row = r.stdout.strip().split("\n")[-1].split(",")
        try:
            samples.append((float(row[5].strip("()Mhz")), float(row[-1])))
        except ValueError:
            pass
        time.sleep(0.2)


cases = [("vendor matmul", lambda: torch.matmul(a, w.t())),
         ("amds cfg 10 bias", lambda: ops.gemm(a, w, _lib.EPI_BIAS, bias=b, cfg=10)),
         ("amds cfg 12 bias", lambda: ops.gemm(a, w, _lib.EPI_BIAS, bias=b, cfg=12)),
         ("amds cfg 13 bias", lambda: ops.gemm(a, w, _lib.EPI_BIAS, bias=b, cfg=13)),
         ("amds cfg 12 bias (2)", lambda: ops.gemm(a, w, _lib.EPI_BIAS, bias=b, cfg=12)),
         ("amds cfg 13 bias (2)", lambda: ops.gemm(a, w, _lib.EPI_BIAS, bias=b, cfg=13)),
         ("amds cfg 8 bias", lambda: ops.gemm(a, w, _lib.EPI_BIAS, bias=b, cfg=8)),
         ("amds cfg 10 residual", lambda: ops.gemm(a, w, _lib.EPI_RESIDUAL, bias=b, out=res, cfg=10)),
         ("amds cfg 12 residual", lambda: ops.gemm(a, w, _lib.EPI_RESIDUAL, bias=b, out=res, cfg=12)),
         ("amds cfg 10 gelu", lambda: ops.gemm(a, w, _lib.EPI_BIAS_GELU, bias=b, cfg=10)),
         ("amds cfg 12 gelu", lambda: ops.gemm(a, w, _lib.EPI_BIAS_GELU, bias=b, cfg=12))]
for name, fn in cases:
    stop = False
    samples.clear()
    for _ in range(10):
        fn()
    torch.cuda.synchronize()
    th = threading.Thread(target=sampler)
    th.start()
    t0 = time.perf_counter()
    n = 0
    ev = None
    while time.perf_counter() - t0 < 4.0:
        for _ in range(40):
            fn()
        n += 40
        if ev is not None:
            ev.synchronize()          # keep ~40 launches queued, never drain the queue
        ev = torch.cuda.Event()
        ev.record()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    stop = True
    th.join()
    s = samples[2:] or samples
    clk = sum(x[0] for x in s) / max(len(s), 1)
    pw = sum(x[1] for x in s) / max(len(s), 1)
    tf = 2.0 * M * N * K * n / dt / 1e12
    print(f"N={N} K={K} {name:22s} {dt / n * 1e6:7.0f} us {tf:6.0f} TF/s   sclk {clk:5.0f} MHz  power {pw:5.0f} W   {tf / pw * 1e3:5.0f} GFLOP/J")
