"""The multi-slide leg of bench.py alone (extract_slides over ten synthetic slides vs one extract_slide call per slide):  python tools/slides_only.py"""
import json
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import bench  # noqa: E402
from stamp_amd.vit import PRESETS, HipViT, random_vit_state_dict  # noqa: E402

dev = torch.device("cuda:0")
cfg = PRESETS["vit_large_patch14_224"]
model = HipViT(cfg, random_vit_state_dict(cfg, seed=0), device=dev, chunk=1020)
tiles = torch.randint(0, 256, (1020, 224, 224, 3), dtype=torch.uint8, device=dev)
for _ in range(2):
    model(tiles)
torch.cuda.synchronize()
import time
t0 = time.perf_counter()
for _ in range(5):
    model(tiles)
torch.cuda.synchronize()
resident = 5 * 1020 / (time.perf_counter() - t0)
out = bench.slides_leg(model, dev)
out["hbm_resident_tiles_per_s"] = round(resident, 1)
out["vs_hbm_resident"] = round(out["value"] / resident, 4)
print(json.dumps(out))
