"""Where along the depth does the HIP tile encoder leave its rounding emulation?  For a preset truncated to its first k blocks: the kernel's
final tokens / class row vs an fp32 torch evaluation, next to what tools/rounding_budget.py's emulation of the 16-bit rounding sites predicts.
GPU box:  python tools/parity_probe.py uni2_h 2 5 9"""
import sys
from dataclasses import replace
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
sys.path.insert(0, str(Path(__file__).resolve().parent))
from rounding_budget import SITES, forward  # noqa: E402
from stamp_amd.vit import PRESETS, HipViT, random_vit_state_dict  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "uni2_h"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 2
seed = int(sys.argv[3]) if len(sys.argv) > 3 else 5
tseed = int(sys.argv[4]) if len(sys.argv) > 4 else 9
depths = [int(a) for a in sys.argv[5:]] or [1, 2, 4, 8, PRESETS[name].depth]
dev = "cuda"
full = PRESETS[name]
sd_full = random_vit_state_dict(full, seed=seed)
tiles = torch.randint(0, 256, (n, 224, 224, 3), dtype=torch.uint8, generator=torch.Generator().manual_seed(tseed)).to(dev)
rel = lambda a, b: ((a.double() - b.double()).norm() / b.double().norm()).item()  # noqa: E731
for k in depths:
    cfg = replace(full, depth=k)
    sd = {key: v for key, v in sd_full.items() if not key.startswith("blocks.") or int(key.split(".")[1]) < k}
    sdd = {key: v.float().to(dev) for key, v in sd.items()}
    with torch.no_grad():
        ref = forward(tiles, sdd, cfg, set(), dev=dev)
        emu = forward(tiles, sdd, cfg, set(SITES), dev=dev)
        emu_x = forward(tiles, sdd, cfg, set(SITES), dev=dev, cls_exact=True)
    line = f"depth {k:2d}: emulation tokens {rel(emu, ref):.3e} cls {rel(emu[:, 0], ref[:, 0]):.3e} (cls rows exact: {rel(emu_x[:, 0], ref[:, 0]):.3e})"
    for fold in (True, False):
        f, t = HipViT(cfg, sd, device=dev, chunk=n, ln_fold=fold)(tiles, return_tokens=True)
        line += f" | kernel{'(fold)' if fold else '(plain LN)'} tokens {rel(t, ref):.3e} cls {rel(t[:, 0], ref[:, 0]):.3e} vs emu {rel(t, emu):.3e}"
    print(line, flush=True)
