"""Shader clock / power while the production GEMM loops (rocm-smi sampled from a thread).  python tools/clock_under_load.py [cfg]"""
import subprocess
import sys
import threading
import time
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from stamp_amd import _lib, ops  # noqa: E402

cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 8
M, N, K = 131070, 4096, 1024
a = torch.randn(M, K, device="cuda").half()
w = (torch.randn(N, K, device="cuda") / 32).half()
b = torch.zeros(N, device="cuda")
stop = False
samples = []


def sampler():
    while not stop:
        r = subprocess.run(["rocm-smi", "--showclocks", "--showpower", "--csv"], capture_output=True, text=True)
        samples.append((time.perf_counter(), r.stdout.strip().replace("\n", " | ")))
        time.sleep(0.3)


def smi_once(tag):
    r = subprocess.run(["rocm-smi", "--showclocks", "--showpower", "--csv"], capture_output=True, text=True)
    print(tag, r.stdout.strip().replace("\n", " | ")[:600])


smi_once("idle:")
for epi, name in ((_lib.EPI_BIAS, "bias"), (_lib.EPI_BIAS_GELU, "gelu")):
    stop = False
    samples.clear()
    th = threading.Thread(target=sampler)
    th.start()
    t0 = time.perf_counter()
    n = 0
    while time.perf_counter() - t0 < 4.0:
        for _ in range(50):
            ops.gemm(a, w, epi, bias=b, cfg=cfg)
        torch.cuda.synchronize()
        n += 50
    dt = time.perf_counter() - t0
    stop = True
    th.join()
    print(f"{name}: {dt / n * 1e6:.0f} us/launch, {2.0 * M * N * K * n / dt / 1e12:.0f} TFLOP/s")
    for t, s in samples[-4:]:
        print("  ", s[:600])
