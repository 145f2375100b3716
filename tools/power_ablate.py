"""Sustained (power-capped) rate of the ablated 4-wave GEMM variants: which part of the K loop costs the joules.
python tools/power_ablate.py"""
import ctypes as C
import subprocess
import sys
import threading
import time
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from stamp_amd import _lib  # noqa: E402

lib = _lib.lib()
f = lib.amds_gemm_ablate
f.restype = C.c_int
f.argtypes = [C.c_int, C.c_void_p, C.c_long, C.c_void_p, C.c_long, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_long, C.c_void_p, C.c_void_p]
M, N, K = (131070, 3072, 1024) if len(sys.argv) > 2 and sys.argv[2] == "qkv" else (131070, 1024, 4096)
g = torch.Generator().manual_seed(0)
A = torch.randn(M, K, generator=g).to("cuda", torch.float16)
w = (torch.randn(N, K, generator=g) / K ** 0.5).to("cuda", torch.float16)
bias = torch.randn(N, generator=g).to("cuda")
out = torch.empty(M, N, dtype=torch.float16, device="cuda")
st = torch.cuda.current_stream().cuda_stream
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


names = {2000: "full", 2001: "no LDS-DMA", 2004: "no fragment reads", 2008: "no barriers", 2005: "no DMA, no reads", 2013: "MFMA only"}
for abl, nm in names.items():
    fn = lambda: f(abl, A.data_ptr(), K, w.data_ptr(), K, M, N, K, out.data_ptr(), N, bias.data_ptr(), st)  # noqa: E731
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
    while time.perf_counter() - t0 < 3.0:
        for _ in range(40):
            fn()
        n += 40
        if ev is not None:
            ev.synchronize()
        ev = torch.cuda.Event()
        ev.record()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    stop = True
    th.join()
    s = samples[2:] or samples
    clk = sum(x[0] for x in s) / max(len(s), 1)
    pw = sum(x[1] for x in s) / max(len(s), 1)
    print(f"4w64 {nm:20s} {dt / n * 1e6:7.0f} us ({2.0 * M * N * K * n / dt / 1e12:6.0f} 'TF/s')  sclk {clk:5.0f} MHz  power {pw:5.0f} W")
