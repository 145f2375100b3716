"""The reference's DataLoader batch (64 tiles) through HipViT, tiles resident on the device, for rocprofv3:  python tools/b64_only.py [n_batches]"""
import sys
import time
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from stamp_amd.vit import PRESETS, HipViT, random_vit_state_dict  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 8
cfg = PRESETS["vit_large_patch14_224"]
model = HipViT(cfg, random_vit_state_dict(cfg, seed=0, init="moderate"), device="cuda")
tiles = torch.randint(0, 256, (64, 224, 224, 3), dtype=torch.uint8, device="cuda")
for _ in range(3):
    model(tiles)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(n):
    model(tiles)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / n
print(f"B=64 resident: {64 / dt:.0f} tiles/s, {dt * 1e3:.2f} ms per batch")
