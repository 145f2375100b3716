"""Socket power / shader clock while the tile encoder's attention and LayerNorm kernels loop (same protocol as power_compare.py)."""
import subprocess
import sys
import threading
import time
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from stamp_amd import ops  # noqa: E402

B, T, H = 510, 257, 16
qkv = torch.randn(B * T, 3 * H * 64, device="cuda").half()
x = torch.randn(B * T, 1024, device="cuda")
g = torch.ones(1024, device="cuda"); b = torch.zeros(1024, device="cuda")
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


for name, fn in (("attention_vit 510 x 16 x 257", lambda: ops.attention_vit(qkv, B, T, H)),
                 ("layernorm 131070 x 1024 f32->f16", lambda: ops.layernorm(x, g, b, 1e-6, torch.float16))):
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
    print(f"{name:36s} {dt / n * 1e6:7.1f} us   sclk {sum(v[0] for v in s) / len(s):5.0f} MHz  power {sum(v[1] for v in s) / len(s):5.0f} W")
