"""The synthetic-slide leg of bench.py alone (for rocprofv3 --kernel-trace): python tools/slide_only.py [tiles] [model]"""
import sys
import tempfile
import time
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from bench import SyntheticSlide  # noqa: E402
from stamp_amd.extractor import Extractor, u8_tile_transform  # noqa: E402
from stamp_amd.preprocess import extract_slide  # noqa: E402
from stamp_amd.vit import PRESETS, HipViT, random_vit_state_dict  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
name = sys.argv[2] if len(sys.argv) > 2 else "vit_large_patch14_224"
canny = None if len(sys.argv) > 3 and sys.argv[3] == "nocanny" else 0.02
workers = int(sys.argv[4]) if len(sys.argv) > 4 else 32
spb = int(sys.argv[5]) if len(sys.argv) > 5 else 16
cfg = PRESETS[name]
model = HipViT(cfg, random_vit_state_dict(cfg, 0), device="cuda")
ex = Extractor(model=model, transform=u8_tile_transform, identifier="amdstamp-bench")
side = max(2, int(round((n / 4) ** 0.5)))
with tempfile.TemporaryDirectory() as td:
    extract_slide(SyntheticSlide(8 * 1024, 8 * 1024, seed=5), ex, Path(td) / "w.h5", slide_mpp=0.5, max_workers=32)
    torch.cuda.synchronize()
    big = SyntheticSlide(side * 1024, side * 1024, seed=5)
    t0 = time.perf_counter()
    st = extract_slide(big, ex, Path(td) / "s.h5", slide_mpp=0.5, max_workers=workers, canny_cutoff=canny, supertiles_per_batch=spb)
    el = time.perf_counter() - t0
import stamp_amd.preprocess as PP
if len(sys.argv) > 6:
    PP._TIMELINE = []
    with tempfile.TemporaryDirectory() as td:
        extract_slide(SyntheticSlide(side * 1024, side * 1024, seed=5), ex, Path(td) / "t.h5", slide_mpp=0.5, max_workers=workers, canny_cutoff=canny, supertiles_per_batch=spb)
    torch.cuda.synchronize()
    encs = [e for e in PP._TIMELINE if e[0] == "enc_enqueue"]
    base = encs[0][2]
    print("host time of encoder enqueue (s) / GPU start of that encoder call relative to the first (ms):")
    print([(round(t, 3), round(base.elapsed_time(ev), 1)) for _, t, ev in encs])
    h2 = [e for e in PP._TIMELINE if e[0] == "h2d"]
    print("H2D copies: (host enqueue s, GPU start ms rel. first encoder, duration ms):", [(round(t, 3), round(base.elapsed_time(p[0]), 1), round(p[0].elapsed_time(p[1]), 1)) for _, t, p in h2[:40]])
    print("batch arrivals (s):", [round(t, 3) for n, t, _ in PP._TIMELINE if n == "batch_ready"])
print({"workers": workers, "spb": spb, **st, "seconds": round(el, 2), "tiles_per_s": round(st["tiles_kept"] / el, 1)})
