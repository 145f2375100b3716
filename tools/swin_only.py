"""Run only the HIP CTransPath forward (for rocprofv3 / A-B timing):  python tools/swin_only.py [tiles] [chunk] [iters]"""
import sys
import time
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch  # noqa: E402

from stamp_amd.swin import SWIN_PRESETS, HipSwin, random_swin_state_dict  # noqa: E402

tiles_n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
chunk = int(sys.argv[2]) if len(sys.argv) > 2 else 256
iters = int(sys.argv[3]) if len(sys.argv) > 3 else 5
cfg = SWIN_PRESETS["ctranspath"]
model = HipSwin(cfg, random_swin_state_dict(cfg, 0), device="cuda", chunk=chunk)
tiles = torch.randint(0, 256, (tiles_n, 224, 224, 3), dtype=torch.uint8, generator=torch.Generator().manual_seed(1)).cuda()
for _ in range(2):
    model(tiles)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(iters):
    f = model(tiles)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / iters
print(f"ctranspath: {tiles_n} tiles chunk {chunk}: {dt*1e3:.2f} ms/forward, {tiles_n/dt:.0f} tiles/s, finite={bool(torch.isfinite(f.float()).all())}")
