"""Does a captured graph help the reference's batch of 64?  python tools/b64_graph_probe.py [n]  -- plain calls vs one torch.cuda.CUDAGraph replay per batch,
plus the host time of issuing one forward (launch-bound or not)."""
import sys
import time
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from stamp_amd.vit import PRESETS, HipViT, random_vit_state_dict  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 16
cfg = PRESETS["vit_large_patch14_224"]
model = HipViT(cfg, random_vit_state_dict(cfg, seed=0, init="moderate"), device="cuda")
tiles = torch.randint(0, 256, (64, 224, 224, 3), dtype=torch.uint8, device="cuda")
for _ in range(3):
    ref = model(tiles)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(n):
    model(tiles)
t_issue = (time.perf_counter() - t0) / n
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / n
print(f"plain : {64 / dt:.0f} tiles/s, {dt * 1e3:.2f} ms per batch, host issue time {t_issue * 1e3:.2f} ms")
try:
    g = torch.cuda.CUDAGraph()
    static_in = tiles.clone()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(2):
            model(static_in)
    torch.cuda.current_stream().wait_stream(s)
    with torch.cuda.graph(g):
        static_out = model(static_in)
    g.replay()
    torch.cuda.synchronize()
    print("graph result equals plain:", torch.equal(static_out, ref))
    t0 = time.perf_counter()
    for _ in range(n):
        static_in.copy_(tiles)
        g.replay()
        out = static_out.clone()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    print(f"graph : {64 / dt:.0f} tiles/s, {dt * 1e3:.2f} ms per batch")
except Exception as e:  # noqa: BLE001
    print("graph capture failed:", repr(e)[:400])
