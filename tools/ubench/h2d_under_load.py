"""H2D bandwidth of a pinned 256 MB buffer on a side stream, idle and while the tile encoder runs on the main stream."""
import sys
import time
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))
from stamp_amd.vit import PRESETS, HipViT, random_vit_state_dict  # noqa: E402

cfg = PRESETS["vit_large_patch14_224"]
model = HipViT(cfg, random_vit_state_dict(cfg, 0), device="cuda")
tiles = torch.randint(0, 256, (1020, 224, 224, 3), dtype=torch.uint8, device="cuda")
h = torch.empty(256 << 20, dtype=torch.uint8).pin_memory()
d = torch.empty(256 << 20, dtype=torch.uint8, device="cuda")
side = torch.cuda.Stream()


def copies(n):
    with torch.cuda.stream(side):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(side)
        for _ in range(n):
            d.copy_(h, non_blocking=True)
        b.record(side)
    return a, b


model(tiles)
torch.cuda.synchronize()
a, b = copies(8)
torch.cuda.synchronize()
print(f"idle:       {8 * 0.268 / (a.elapsed_time(b) * 1e-3):.1f} GB/s")
for _ in range(4):
    model(tiles)
a, b = copies(8)
t0 = time.perf_counter()
torch.cuda.synchronize()
print(f"under load: {8 * 0.268 / (a.elapsed_time(b) * 1e-3):.1f} GB/s   (4 encoder calls queued before the copies; drained in {time.perf_counter() - t0:.2f} s)")
