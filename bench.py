#!/usr/bin/env python
"""bench.py -- tiles/s of the MI355X tile-encoder hot path (BASELINE.json configs[1]: ViT-L/14 tile extraction
on synthetic 224x224x3 u8 tiles, random-init weights of that architecture).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

A "step" is one pass of the hot path over one virtual slide of `--tiles` u8 tiles already resident in HBM:
im2col -> patch-embed GEMM -> 24 x {LN, QKV GEMM, attention, proj GEMM(+residual), LN, fc1 GEMM(+GELU),
fc2 GEMM(+residual)} -> final LN on CLS -> fp16 features in HBM.  Slides shard across ranks (weak scaling: every
rank encodes its own slide per step); for N>1 every step ends with the path's one collective, an RCCL all-gather
of the slide-level embeddings.  Rank 0 prints ONE JSON line.  Beside `value` (HBM-resident, the contract's metric) the
single-GPU line carries:
  end_to_end   SURVEY.md 8d's M1: a pinned host pool of --e2e-tiles DISTINCT tiles -> double-buffered H2D -> encode -> fp16
               features D2H, wall clock from the first H2D enqueue to the last feature row on the host (PCIe-inclusive)
  drop_in_b64  what an unmodified `extract_` gets: `model(tiles.to(device)).detach().half().cpu()` on batches of 64
               (reference src/stamp/preprocessing/__init__.py:315-327)
  secondary    MIL bags/s (vit head deploy / train / +ALiBi, TransMIL), CTransPath tiles/s
  cpu_baseline the oracle on the host cores (ViT-L/14, batch 64) + the MIL / pooling CPU baselines of BASELINE.md section 3
"""
from __future__ import annotations

import argparse
import ctypes as C
import json
import os
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

import numpy as np  # noqa: E402
import torch  # noqa: E402

MFMA_PEAK_TFLOPS = 2500.0   # dense bf16/fp16 MFMA peak, /opt/skills/guides/MI355X_MICROARCH.md


def parse() -> argparse.Namespace:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=8)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--tiles", type=int, default=1020, help="tiles per step per GPU (one virtual slide)")
    ap.add_argument("--chunk", type=int, default=1020, help="tiles per internal forward chunk")
    ap.add_argument("--model", default="vit_large_patch14_224", help="a ViT preset, or ctranspath (ConvStem + Swin-T)")
    ap.add_argument("--swin-chunk", type=int, default=1024, help="tiles per internal chunk of the CTransPath forward (5.2 MB of workspace per tile)")
    ap.add_argument("--act", default="f16", choices=["f16", "bf16"])
    ap.add_argument("--overlap", type=int, default=0, help="1 = two chunks in flight on two streams")
    ap.add_argument("--fp8", action="store_true", help="headline run with HipViT(fp8=True): the blocks' Linears on the fp8 MFMA (opt-in, ~6 %% feature error)")
    ap.add_argument("--check", default="fallback", choices=("fallback", "raise", "off"), help="HipViT's guard in front of the features (default: the product's default; 'off' for A/B)")
    ap.add_argument("--exact", action="store_true", help="headline run with HipViT(exact=True): the class-token rows also on an exact-fp32 stream")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true", help="headline workload only (for a ViT-only rocprofv3 kernel trace)")
    ap.add_argument("--cpu-seconds", type=float, default=20.0)
    ap.add_argument("--e2e-tiles", type=int, default=100_000, help="distinct tiles of the end-to-end (M1) leg; 0 = skip")
    ap.add_argument("--slide-tiles", type=int, default=20_736, help="tiles of the synthetic-slide leg (decode threads -> GPU resize / Canny -> encoder -> .h5); 0 = skip")
    ap.add_argument("--e2e-warmup", type=int, default=2, help="batches of the end-to-end leg run before its clock starts")
    return ap.parse_args()


# ---------------------------------------------------------------------------------------------------------------------------
# CPU baselines (the oracle = the reference's algorithm on torch CPU fp32 kernels), rank 0 at N = 1 only, bounded samples
# ---------------------------------------------------------------------------------------------------------------------------
def _pick_threads(fn, candidates) -> int:
    """Untimed calibration: the host has 2 sockets x 64 cores x SMT; torch's default (all hardware threads) is far from the
    fastest setting for GEMM-bound fp32 work.  Try a few, keep the best."""
    best, best_t = candidates[0], float("inf")
    for n in candidates:
        torch.set_num_threads(n)
        fn()
        t0 = time.perf_counter()
        fn()
        dt = time.perf_counter() - t0
        if dt < best_t:
            best, best_t = n, dt
    torch.set_num_threads(best)
    return best


def cpu_baseline(cfg, sd, seconds: float, swin: bool = False) -> dict:
    """C1 of BASELINE.md section 3: u8 tiles -> transform -> ViT forward (fp32, batch 64 like the reference's DataLoader,
    src/stamp/preprocessing/__init__.py:317, attention through F.scaled_dot_product_attention as timm does) -> fp16."""
    if swin:
        from oracle.swin_ctranspath import swin_encode_f16 as run
    else:
        from oracle.vit_tile_encoder import extract_features

        def run(t, s, c):
            return extract_features(t, s, c, sdpa=True)
    hw = os.cpu_count() or 8
    g = torch.Generator().manual_seed(1234)
    batch = 64
    tiles = torch.randint(0, 256, (batch, cfg.img, cfg.img, 3), dtype=torch.uint8, generator=g)
    cands = sorted({n for n in (16, 32, 64, 128) if n <= hw}) or [hw]
    threads = _pick_threads(lambda: run(tiles[:32], sd, cfg), cands)
    run(tiles[:8], sd, cfg)           # warm
    n, t0 = 0, time.perf_counter()
    while True:
        run(tiles, sd, cfg)
        n += batch
        el = time.perf_counter() - t0
        if el >= seconds or n >= 1024:
            break
    return {"value": round(n / el, 3), "unit": "tiles/s", "cores": threads, "kind": "port",
            "sample": (f"{n} synthetic 224x224 tiles (batches of {batch}, the reference's DataLoader batch), "
                       f"{'CTransPath (Swin-T)' if swin else 'ViT'} fp32 oracle on torch CPU kernels with SDPA attention, {el:.1f}s; "
                       f"thread count chosen by an untimed calibration over {cands} of {hw} hardware threads")}


def cpu_baseline_mil(seconds_each: float = 4.0) -> dict:
    """C2-C4 of BASELINE.md section 3 on the host: MIL `vit` train step (fwd + bwd + AdamW through torch autograd on the
    oracle network, dropout sites live as in the reference's train mode), MIL forward (deploy: batch 1, full bag), gated-attention
    pooling.  Bags of 1024 x 1024-d.  Bounded: a few steps each."""
    from oracle.gated_attention import KEYS, gated_attention_pool
    from oracle.mil_vit import mil_vit_forward
    from stamp_amd.mil import VisionTransformer

    out = {}
    torch.manual_seed(1)
    model = VisionTransformer(dim_output=2, dim_input=1024, dim_model=512, n_layers=2, n_heads=8, dim_feedforward=512, dropout=0.25, use_alibi=False)
    params = {k: v.clone().requires_grad_(True) for k, v in model.state_dict().items()}
    opt = torch.optim.AdamW(list(params.values()), lr=1e-4)
    Bb, Tn, D, H = 8, 1024, 512, 8
    bags = torch.randn(Bb, Tn, 1024).half().float()
    coords = torch.zeros(Bb, Tn, 2)
    targets = torch.nn.functional.one_hot(torch.arange(Bb) % 2, 2).float()

    def drop_masks():
        def m(shape, p):
            return (torch.rand(shape) >= p).float() / (1 - p)
        d = {"proj": m((Bb, Tn, D), 0.25)}
        for l in range(2):
            d[f"attn{l}"], d[f"ff1_{l}"], d[f"ff2_{l}"] = m((Bb, H, Tn + 1, Tn + 1), 0.25), m((Bb, Tn + 1, D), 0.5), m((Bb, Tn + 1, D), 0.5)
        return d

    def train_step():
        opt.zero_grad()
        loss = torch.nn.functional.cross_entropy(mil_vit_forward(bags, coords, None, params, n_heads=H, use_alibi=False, drop=drop_masks()), targets)
        loss.backward()
        opt.step()

    hw = os.cpu_count() or 8
    threads = _pick_threads(train_step, sorted({n for n in (16, 32, 64, hw) if n <= hw}))
    n, t0 = 0, time.perf_counter()
    while time.perf_counter() - t0 < seconds_each and n < 40:
        train_step()
        n += 1
    el = time.perf_counter() - t0
    out["mil_vit_train"] = {"value": round(n * Bb / el, 2), "unit": "bags/s", "cores": threads,
                            "sample": f"{n} steps of batch {Bb}, bags of 1024 x 1024-d, fwd + bwd + AdamW, train-mode dropout, {el:.1f}s"}
    sdp = {k: v.detach() for k, v in params.items()}
    with torch.no_grad():
        mil_vit_forward(bags[:1], coords[:1], None, sdp, n_heads=H, use_alibi=False)
        n, t0 = 0, time.perf_counter()
        while time.perf_counter() - t0 < seconds_each and n < 200:
            mil_vit_forward(bags[n % Bb:n % Bb + 1], coords[:1], None, sdp, n_heads=H, use_alibi=False)
            n += 1
        el = time.perf_counter() - t0
    out["mil_vit_deploy"] = {"value": round(n / el, 2), "unit": "bags/s", "cores": threads, "sample": f"{n} forwards of batch 1, bag 1024 x 1024-d, {el:.1f}s"}
    g = torch.Generator().manual_seed(5)
    N, F_, L, Dd = 1024, 768, 512, 256
    sdg = {KEYS["fc_w"]: torch.randn(L, F_, generator=g) / F_ ** 0.5, KEYS["fc_b"]: torch.zeros(L), KEYS["a_w"]: torch.randn(Dd, L, generator=g) / L ** 0.5,
           KEYS["a_b"]: torch.zeros(Dd), KEYS["b_w"]: torch.randn(Dd, L, generator=g) / L ** 0.5, KEYS["b_b"]: torch.zeros(Dd),
           KEYS["c_w"]: torch.randn(1, Dd, generator=g) / Dd ** 0.5, KEYS["c_b"]: torch.zeros(1)}
    x = torch.randn(N, F_, generator=g)
    with torch.no_grad():
        gated_attention_pool(x, sdg)
        n, t0 = 0, time.perf_counter()
        while time.perf_counter() - t0 < min(seconds_each, 2.0) and n < 2000:
            gated_attention_pool(x, sdg)
            n += 1
        el = time.perf_counter() - t0
    out["gated_attention_pool"] = {"value": round(n / el, 1), "unit": "bags/s", "cores": threads, "sample": f"{n} bags of 1024 x 768, {el:.1f}s"}

    # C2, the other heads: `vit` with ALiBi (post-softmax distance bias, running-mean scalers updated in train mode) and TransMIL
    from oracle.transmil import transmil_forward
    from stamp_amd.mil import TransMIL
    torch.manual_seed(2)
    am = VisionTransformer(dim_output=2, dim_input=1024, dim_model=512, n_layers=2, n_heads=8, dim_feedforward=512, dropout=0.25, use_alibi=True)
    ap = {k: (v.clone().requires_grad_(True) if v.is_floating_point() and "running_mean" not in k else v.clone()) for k, v in am.state_dict().items()}
    aopt = torch.optim.AdamW([v for v in ap.values() if v.requires_grad], lr=1e-4)
    gx, gy = torch.meshgrid(torch.arange(32.0), torch.arange(32.0), indexing="ij")
    acoords = (torch.stack([gx.reshape(-1), gy.reshape(-1)], -1) * 256.0).expand(Bb, Tn, 2).contiguous()       # a 256 um grid, as BASELINE.md says

    def alibi_step():
        aopt.zero_grad()
        d = {k: v for k, v in drop_masks().items() if not k.startswith("attn")}        # ALiBi heads have no attention dropout
        loss = torch.nn.functional.cross_entropy(mil_vit_forward(bags, acoords, None, ap, n_heads=H, use_alibi=True, drop=d), targets)
        loss.backward()
        aopt.step()

    alibi_step()
    n, t0 = 0, time.perf_counter()
    while time.perf_counter() - t0 < seconds_each and n < 20:
        alibi_step()
        n += 1
    el = time.perf_counter() - t0
    out["mil_vit_train_alibi"] = {"value": round(n * Bb / el, 2), "unit": "bags/s", "cores": threads,
                                  "sample": f"{n} steps of batch {Bb}, bags of 1024 x 1024-d with coordinates, fwd + bwd + AdamW, {el:.1f}s"}
    tm = TransMIL(dim_output=2, dim_input=1024, dim_hidden=512)
    tp = {k: v.clone().requires_grad_(True) for k, v in tm.state_dict().items()}
    topt = torch.optim.AdamW(list(tp.values()), lr=1e-4)

    def transmil_step():
        topt.zero_grad()
        d = {name: (torch.rand(Bb, Tn + 1, 512) >= 0.1).float() / 0.9 for name in ("layer1", "layer2")}
        loss = torch.nn.functional.cross_entropy(transmil_forward(bags, tp, drop=d), targets)
        loss.backward()
        topt.step()

    transmil_step()
    n, t0 = 0, time.perf_counter()
    while time.perf_counter() - t0 < seconds_each and n < 20:
        transmil_step()
        n += 1
    el = time.perf_counter() - t0
    out["transmil_train"] = {"value": round(n * Bb / el, 2), "unit": "bags/s", "cores": threads,
                             "sample": f"{n} steps of batch {Bb}, bags of 1024 x 1024-d, fwd + bwd + AdamW, Dropout(0.1) live, {el:.1f}s"}

    # configs[4]'s buildable part on the host: Cox-survival `vit` head (dim_output 1, Efron), bags of 1024 x 768-d, batch 8
    from oracle.misc import cox_neg_partial_log_likelihood
    torch.manual_seed(4)
    sm = VisionTransformer(dim_output=1, dim_input=768, dim_model=512, n_layers=2, n_heads=8, dim_feedforward=512, dropout=0.25, use_alibi=False)
    sp = {k: v.clone().requires_grad_(True) for k, v in sm.state_dict().items()}
    sopt = torch.optim.AdamW(list(sp.values()), lr=1e-4)
    sbags = torch.randn(Bb, Tn, 768).half().float()
    stime, sevent = torch.rand(Bb) * 1970 + 30, torch.tensor([1, 1, 0, 1, 1, 0, 1, 1], dtype=torch.bool)

    def surv_step():
        sopt.zero_grad()
        pred = mil_vit_forward(sbags, coords, None, sp, n_heads=H, use_alibi=False, drop=drop_masks())
        cox_neg_partial_log_likelihood(pred.squeeze(-1), stime, sevent).backward()
        sopt.step()

    surv_step()
    n, t0 = 0, time.perf_counter()
    while time.perf_counter() - t0 < seconds_each and n < 20:
        surv_step()
        n += 1
    el = time.perf_counter() - t0
    out["survival_train"] = {"value": round(n * Bb / el, 2), "unit": "bags/s", "cores": threads,
                             "sample": f"{n} steps of batch {Bb}, bags of 1024 x 768-d, Cox-survival vit head (Efron), fwd + bwd + AdamW, train-mode dropout, {el:.1f}s"}

    # C5: BASELINE.json configs[0], tests/random_data.py-shaped: 64 patients x 256 tiles x 2048-d, binary, `vit` head, 2 epochs of
    # (51 training bags in one batch of the reference's batch size 64, then 13 full-bag validation forwards), wall seconds
    torch.manual_seed(3)
    c1 = VisionTransformer(dim_output=2, dim_input=2048, dim_model=512, n_layers=2, n_heads=8, dim_feedforward=512, dropout=0.25, use_alibi=False)
    cp = {k: v.clone().requires_grad_(True) for k, v in c1.state_dict().items()}
    copt = torch.optim.AdamW(list(cp.values()), lr=1e-4)
    cb = torch.rand(64, 256, 2048).half().float()
    ct = torch.nn.functional.one_hot(torch.arange(64) % 2, 2).float()
    cz = torch.zeros(64, 256, 2)
    t0 = time.perf_counter()
    for _ in range(2):
        copt.zero_grad()
        dm = {"proj": (torch.rand(51, 256, 512) >= 0.25).float() / 0.75}
        for l in range(2):
            dm[f"attn{l}"] = (torch.rand(51, 8, 257, 257) >= 0.25).float() / 0.75
            dm[f"ff1_{l}"], dm[f"ff2_{l}"] = (torch.rand(51, 257, 512) >= 0.5).float() * 2, (torch.rand(51, 257, 512) >= 0.5).float() * 2
        torch.nn.functional.cross_entropy(mil_vit_forward(cb[:51], cz[:51], None, cp, n_heads=8, use_alibi=False, drop=dm), ct[:51]).backward()
        copt.step()
        with torch.no_grad():
            for i in range(51, 64):
                mil_vit_forward(cb[i:i + 1], cz[i:i + 1], None, cp, n_heads=8, use_alibi=False)
    out["config1_two_epochs"] = {"value": round(time.perf_counter() - t0, 3), "unit": "s", "cores": threads,
                                 "sample": "64 bags x 256 tiles x 2048-d, 2 epochs: one training step over 51 bags + 13 validation forwards each"}
    return out


# ---------------------------------------------------------------------------------------------------------------------------
def end_to_end_leg(model, cfg, dev, n_tiles: int, batch: int, warm_batches: int) -> dict:
    """SURVEY.md 8d M1.  The pool is generated ON the GPU with a seeded generator (distinct tiles, 150 KB each) and parked in
    pinned host memory; then pinned host -> H2D -> encode -> fp16 D2H through stamp_amd.extractor.TilePipeline."""
    from stamp_amd.extractor import TilePipeline

    n_tiles = (n_tiles // batch) * batch
    pool = torch.empty(n_tiles, cfg.img, cfg.img, 3, dtype=torch.uint8).pin_memory()
    gen = torch.Generator(device=dev).manual_seed(1234)
    for i in range(0, n_tiles, 4 * batch):
        j = min(n_tiles, i + 4 * batch)
        pool[i:j].copy_(torch.randint(0, 256, (j - i, cfg.img, cfg.img, 3), dtype=torch.uint8, device=dev, generator=gen))
    dim = getattr(cfg, "out_dim", None) or cfg.dim
    feats = torch.empty(n_tiles, dim, dtype=torch.float16).pin_memory()
    pipe = TilePipeline(model, batch_size=batch, device=dev, tile_shape=(cfg.img, cfg.img, 3), feat_dim=dim)
    for i in range(warm_batches):                     # steady state: the first batches are discarded
        pipe.submit(pool[i * batch:(i + 1) * batch], feats[i * batch:(i + 1) * batch])
    pipe.finish()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(0, n_tiles, batch):
        pipe.submit(pool[i:i + batch], feats[i:i + batch])
    pipe.finish()
    el = time.perf_counter() - t0
    ok = bool(torch.isfinite(feats[::997].float()).all()) and not bool((feats[-1] == 0).all())
    # spot check: a pipelined feature row equals the one the plain call gives for the same tile
    same = bool(torch.equal(model(pool[-batch:].to(dev))[-1].cpu(), feats[-1]))
    return {"metric": "tiles/s end to end (M1: pinned host u8 -> H2D -> encode -> fp16 features on the host)", "value": round(n_tiles / el, 1),
            "unit": "tiles/s", "tiles": n_tiles, "distinct": True, "batch": batch, "seconds": round(el, 2), "finite": ok, "matches_plain_call": same,
            "h2d_gbytes": round(n_tiles * cfg.img * cfg.img * 3 / 1e9, 2), "d2h_mbytes": round(n_tiles * dim * 2 / 1e6, 1)}


class SyntheticSlide:
    """openslide's surface (`dimensions`, `read_region` -> RGBA PIL image, transparent past the edge, `get_thumbnail`) over a procedurally
    repeated base image, so that a 70 k x 70 k pixel slide costs 48 MB: pixel (x, y) = base[y % bh, x % bw].  `read_region` does real work per
    call (a gather of S x S pixels + the RGBA assembly), comparable to openslide serving an uncompressed region; JPEG decode is not modelled."""

    def __init__(self, width: int, height: int, seed: int = 0, base: int = 4096):
        import numpy as np
        rng = np.random.default_rng(seed)
        # H&E-like: two stain colours on white, blob-structured (low-frequency field, thresholded) + pixel noise, so the Canny filter sees edges
        f = rng.standard_normal((base // 16, base // 16)).astype(np.float32)
        f = np.kron(f, np.ones((16, 16), np.float32))
        f = (f + np.roll(f, 7, 0) + np.roll(f, 5, 1)) / 3 + 0.35 * rng.standard_normal((base, base)).astype(np.float32)
        a = np.clip(f, 0, 1)[..., None]
        b = np.clip(-f, 0, 1)[..., None]
        rgb = 255.0 - a * (255.0 - np.array([120.0, 60.0, 150.0])) - b * (255.0 - np.array([230.0, 130.0, 170.0]))
        rgb = np.clip(rgb + rng.normal(0, 6, rgb.shape), 0, 255).astype(np.uint8)
        self._base = np.ascontiguousarray(np.dstack([rgb, np.full(rgb.shape[:2], 255, np.uint8)]))       # RGBA
        self.dimensions = (int(width), int(height))

    def read_region(self, location, level, size):
        import numpy as np
        from PIL import Image
        assert level == 0
        x, y = location
        w, h = size
        W, H = self.dimensions
        bh, bw = self._base.shape[:2]
        vw, vh = max(0, min(w, W - x)), max(0, min(h, H - y))
        x0, y0 = x % bw, y % bh
        if vw == w and vh == h and x0 + w <= bw and y0 + h <= bh:          # inside the slide and inside one period: a strided view, one copy
            return Image.fromarray(np.ascontiguousarray(self._base[y0:y0 + h, x0:x0 + w]), "RGBA")
        out = np.zeros((h, w, 4), np.uint8)
        if vw and vh:
            out[:vh, :vw] = self._base[np.ix_((np.arange(y, y + vh) % bh), (np.arange(x, x + vw) % bw))]
        return Image.fromarray(out, "RGBA")

    def get_thumbnail(self, size):
        from PIL import Image
        return Image.new("RGB", tuple(int(v) for v in size), (140, 90, 150))          # tissue everywhere: every supertile is foreground


def slide_leg(model, dev, n_tiles: int) -> dict:
    """SURVEY.md N3 / VERDICT item 6: one synthetic slide OBJECT through `stamp_amd.preprocess.extract_slide` -- reader threads -> pinned ring ->
    GPU resize (PIL-exact) -> GPU Canny -> keep-mask compaction on the device -> encoder on accumulated chunks -> fp16 features -> .h5 -- wall
    clock from the call to the file on disk, next to the batch-by-batch form of round 2 on a slice of the same slide."""
    import tempfile

    from stamp_amd import h5io
    from stamp_amd.extractor import Extractor, u8_tile_transform
    from stamp_amd.preprocess import extract_slide, extract_slide_serial
    side = max(2, int(round((n_tiles / 4) ** 0.5)))          # mpp 0.5: 1024-pixel supertiles of 2 x 2 tiles
    slide = SyntheticSlide(side * 1024, side * 1024, seed=5)
    ex = Extractor(model=model, transform=u8_tile_transform, identifier="amdstamp-bench")
    workers = min(32, os.cpu_count() or 8)
    out = {}
    with tempfile.TemporaryDirectory() as td:
        small = SyntheticSlide(8 * 1024, 8 * 1024, seed=5)
        extract_slide(small, ex, Path(td) / "warm.h5", slide_mpp=0.5, max_workers=workers, device=dev)          # warm: allocations, PIL, HDF5
        t0 = time.perf_counter()
        st = extract_slide(slide, ex, Path(td) / "slide.h5", slide_mpp=0.5, max_workers=workers, device=dev)
        el = time.perf_counter() - t0
        feats, ci, _ = h5io.read_tile_features(Path(td) / "slide.h5")
        out = {"metric": "tiles/s from a slide object to the feature file (decode threads -> GPU resize / Canny / compaction -> encoder -> .h5)",
               "value": round(st["tiles_kept"] / el, 1), "unit": "tiles/s", "tiles_seen": st["tiles_seen"], "tiles_kept": st["tiles_kept"],
               "seconds": round(el, 2), "encoder_calls": st["encoder_calls"], "host_syncs": st["host_syncs"], "host_wait_for_reader_s": round(st["wait_reader_s"], 2),
               "host_wait_for_gpu_s": round(st["wait_gpu_s"], 2), "reader_threads": workers,
               "finite": bool(np.isfinite(feats.astype(np.float32)).all()), "rows_written": int(feats.shape[0])}
        # what the reader side alone sustains (the same threads and buffers, no GPU work): the pipeline cannot be faster than this or than
        # the encoder; how close it gets to min(reader, encoder) is the overlap
        from concurrent import futures

        from stamp_amd import tiling
        from stamp_amd.preprocess import _region_array
        origins = tiling.foreground_coords(slide.dimensions, slide.get_thumbnail((2 * side, 2 * side)), 1024, 240)[:2048]
        scratch = np.empty((workers, 1024, 1024, 4), np.uint8)

        def rd(io):
            scratch[io[0] % workers][...] = _region_array(slide, io[1][0], io[1][1], 1024)
        tr = time.perf_counter()
        with futures.ThreadPoolExecutor(workers) as pool:
            list(pool.map(rd, enumerate(origins)))
        out["reader_only"] = round(4 * len(origins) / (time.perf_counter() - tr), 1)
        # A/B: the batch-by-batch form on a 16 x 16-supertile corner (1 024 tiles), and the pipelined form on the same corner
        corner = SyntheticSlide(16 * 1024, 16 * 1024, seed=5)
        t1 = time.perf_counter()
        s1 = extract_slide_serial(corner, ex, Path(td) / "serial.h5", slide_mpp=0.5, max_workers=workers, device=dev)
        e1 = time.perf_counter() - t1
        t2 = time.perf_counter()
        s2 = extract_slide(corner, ex, Path(td) / "pipe.h5", slide_mpp=0.5, max_workers=workers, device=dev)
        e2 = time.perf_counter() - t2
        fa, _, _ = h5io.read_tile_features(Path(td) / "serial.h5")
        fb, _, _ = h5io.read_tile_features(Path(td) / "pipe.h5")
        out["serial_1024_tiles"] = round(s1["tiles_kept"] / e1, 1)
        out["pipelined_1024_tiles"] = round(s2["tiles_kept"] / e2, 1)
        out["identical_to_serial"] = bool(np.array_equal(fa.view(np.uint16), fb.view(np.uint16)))
    return out


def slides_leg(model, dev, sides=(24, 70, 32, 48, 36, 60, 28, 44, 40, 52)) -> dict:
    """the rank's slide LIST as one pipeline (`stamp_amd.preprocess.extract_slides`): ten synthetic slide objects of 2.3 k - 19.6 k tiles
    (supertile grids of `sides`^2, mpp 0.5: 4 tiles per supertile; 72 k tiles in all) -> ten feature files, wall clock from the call to the last file on
    disk; next to it the same slides one `extract_slide` call after the other (each paying its start-up and drain), and a bit-for-bit comparison of
    the two sets of files."""
    import tempfile

    from stamp_amd import h5io
    from stamp_amd.extractor import Extractor, u8_tile_transform
    from stamp_amd.preprocess import SlideJob, extract_slide, extract_slides
    ex = Extractor(model=model, transform=u8_tile_transform, identifier="amdstamp-bench")
    workers = min(32, os.cpu_count() or 8)
    slides = [SyntheticSlide(s * 1024, s * 1024, seed=20 + i) for i, s in enumerate(sides)]
    with tempfile.TemporaryDirectory() as td:
        td = Path(td)
        extract_slides([SlideJob(SyntheticSlide(8 * 1024, 8 * 1024, seed=5), td / "warm.h5", 0.5)], ex, max_workers=workers, device=dev)
        t0 = time.perf_counter()
        res = extract_slides([SlideJob(sl, td / f"p{i}.h5", 0.5, f"slide{i}") for i, sl in enumerate(slides)], ex, max_workers=workers, device=dev)
        el = time.perf_counter() - t0
        kept = sum(r.get("tiles_kept", 0) for r in res)
        t1 = time.perf_counter()
        kept1 = 0
        for i, sl in enumerate(slides):
            kept1 += extract_slide(sl, ex, td / f"s{i}.h5", slide_mpp=0.5, max_workers=workers, device=dev)["tiles_kept"]
        el1 = time.perf_counter() - t1
        same = True
        for i in range(len(slides)):
            fa, ca, _ = h5io.read_tile_features(td / f"p{i}.h5")
            fb, cb, _ = h5io.read_tile_features(td / f"s{i}.h5")
            same = same and bool(np.array_equal(fa.view(np.uint16), fb.view(np.uint16)) and np.array_equal(ca.coords_um, cb.coords_um))
    return {"metric": "tiles/s over a LIST of slide objects to their feature files, one pipeline across slides (extract_slides)", "value": round(kept / el, 1),
            "unit": "tiles/s", "slides": len(slides), "tiles_per_slide": [4 * s * s for s in sides], "tiles_kept": kept, "seconds": round(el, 2),
            "statuses": sorted({r["status"] for r in res}), "host_wait_for_reader_s": res[0].get("wait_reader_s"),
            "one_extract_slide_call_per_slide": round(kept1 / el1, 1), "files_identical_to_per_slide_calls": same, "reader_threads": workers,
            **({"trace": res[0]["trace"], "per_slide": [(r.get("plan_s"), r.get("planned_at_s"), r.get("finish_s"), r.get("finished_at_s")) for r in res]} if "trace" in res[0] else {})}


def drop_in_b64_leg(model, cfg, dev, n_batches: int = 48) -> dict:
    """The reference's own loop, literally (preprocessing/__init__.py:315-327): batches of 64 from host memory,
    ``model(tiles.to(device)).detach().half().cpu()``, one synchronous round trip per batch."""
    g = torch.Generator().manual_seed(77)
    host = torch.randint(0, 256, (64 * 8, cfg.img, cfg.img, 3), dtype=torch.uint8, generator=g)
    batches = [host[i * 64:(i + 1) * 64] for i in range(8)]
    with torch.inference_mode():
        for b in batches[:3]:
            model(b.to(dev)).detach().half().cpu()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(n_batches):
            f = model(batches[i % 8].to(dev)).detach().half().cpu()
        el = time.perf_counter() - t0
        # the same batches with the tiles already on the device (kernel path at M = 64 x tokens)
        dbat = [b.to(dev) for b in batches]
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for i in range(n_batches):
            model(dbat[i % 8])
        torch.cuda.synchronize()
        el2 = time.perf_counter() - t1
    return {"metric": "tiles/s of an unmodified extract_ loop: model(batch_of_64.to(device)).half().cpu() per batch", "value": round(64 * n_batches / el, 1),
            "unit": "tiles/s", "batches": n_batches, "hbm_resident_b64": round(64 * n_batches / el2, 1), "finite": bool(torch.isfinite(f.float()).all())}


class ClockPowerSampler:
    """Shader clock and socket power of this rank's GPU while a timed block runs, read from the amdgpu hwmon files (freq1_input in Hz, power1_input in
    microwatts; a few microseconds per read) by a thread every 20 ms.  Lets a reader tell a slow box (clock, power cap) from a regression: round 5's
    driver line read 3 % under the builder's leases with nothing on the line to say why.  Best effort: any failure leaves the fields None."""

    def __init__(self, local_rank: int = 0, period: float = 0.02):
        import glob
        import os
        self.files, self.picked_by = None, None
        cards = sorted(glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*/freq1_input"))
        if cards:
            # a box can expose more cards in sysfs than the one this process may use (a shared 8-GPU node): match this rank's device by PCI address;
            # without one, the visible-device index
            f = None
            try:
                pr = torch.cuda.get_device_properties(local_rank)
                want = f"{pr.pci_domain_id:04x}:{pr.pci_bus_id:02x}:{pr.pci_device_id:02x}."
                for c in cards:
                    if os.path.basename(os.path.realpath(c.split("/hwmon/")[0])).startswith(want):
                        f, self.picked_by = c, "pci " + want + "0"
                        break
            except Exception:
                f = None
            if f is None:
                f, self.picked_by = cards[min(local_rank, len(cards) - 1)], "index"
            self.files = (f, f.replace("freq1_input", "power1_input"))
        self.period, self.samples, self._stop, self._th = period, [], False, None

    def _run(self):
        while not self._stop:
            try:
                mhz = int(open(self.files[0]).read()) / 1e6
                try:
                    w = int(open(self.files[1]).read()) / 1e6
                except Exception:
                    w = None
                self.samples.append((mhz, w))
            except Exception:
                pass
            time.sleep(self.period)

    def __enter__(self):
        if self.files:
            import threading
            self.samples, self._stop = [], False
            self._th = threading.Thread(target=self._run, daemon=True)
            self._th.start()
        return self

    def __exit__(self, *exc):
        self._stop = True
        if self._th:
            self._th.join()

    def summary(self) -> dict:
        clk = [c for c, _ in self.samples]
        pw = [w for _, w in self.samples if w is not None]
        return {"sclk_mhz": round(sum(clk) / len(clk), 1) if clk else None, "sclk_mhz_min": round(min(clk), 1) if clk else None,
                "socket_w": round(sum(pw) / len(pw), 1) if pw else None, "socket_w_max": round(max(pw), 1) if pw else None, "samples": len(clk), "card": self.picked_by}


def gated_pool_leg(dev) -> dict:
    """GPU leg of the gated-attention pooling: bags/s, us per bag, launches per bag, fraction of the fp32-MFMA and HBM roofs (tools/gap_only.py is the longer A/B)."""
    from stamp_amd import ops

    N, F_, L, Dd = 1024, 768, 512, 256
    g = torch.Generator().manual_seed(5)
    w = {"fc_w": torch.randn(L, F_, generator=g) / F_ ** 0.5, "fc_b": torch.zeros(L), "a_w": torch.randn(Dd, L, generator=g) / L ** 0.5, "a_b": torch.zeros(Dd),
         "b_w": torch.randn(Dd, L, generator=g) / L ** 0.5, "b_b": torch.zeros(Dd), "c_w": torch.randn(1, Dd, generator=g) / Dd ** 0.5, "c_b": torch.zeros(1)}
    w = {k: v.to(dev).contiguous() for k, v in w.items()}
    B = 256
    xb = torch.randn(B, N, F_, generator=g).to(dev)
    offs = (torch.arange(B + 1, dtype=torch.int64) * N).to(dev)
    flop, byts = 2.0 * N * (F_ * L + 2 * L * Dd), 4.0 * N * F_
    F32_PEAK, HBM = 157.3e12, 6.29e12          # /opt/skills/guides/MI355X_MICROARCH.md: exact-fp32 MFMA peak; achievable HBM (float4 copy)

    def ev(fn, reps, warm=3):
        for _ in range(warm):
            fn()
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(reps):
            fn()
        b.record()
        torch.cuda.synchronize()
        return a.elapsed_time(b) * 1e-3 / reps

    def row(dt, bags, launches):
        return {"bags_per_s": round(bags / dt, 1), "us_per_bag": round(dt / bags * 1e6, 2), "launches_per_bag": launches,
                "tflops": round(bags * flop / dt / 1e12, 2), "frac_f32_mfma_peak": round(bags * flop / dt / F32_PEAK, 4), "frac_hbm_6p29": round(bags * byts / dt / HBM, 4)}

    dt_b = ev(lambda: ops.gated_attn_pool_batched(xb, [N] * B, w, offsets=offs), 20)
    dt_1 = ev(lambda: ops.gated_attn_pool(xb[0], w), 200)
    dt_6 = ev(lambda: ops.gated_attn_pool(xb[0], w, fused=False), 200)
    ob = ops.gated_attn_pool_batched(xb, [N] * B, w, offsets=offs)
    o6 = torch.stack([ops.gated_attn_pool(xb[i], w, fused=False) for i in (0, B - 1)])
    rel = ((ob[[0, B - 1]] - o6).norm() / o6.norm()).item()
    return {"metric": "gated-attention pooling bags/s (CHIEF Attn_Net_Gated + softmax pooling, bags of 1024 x 768 fp32, exact-fp32 MFMA), 256 bags per launch",
            "value": round(B / dt_b, 1), "unit": "bags/s", "gflop_per_bag": round(flop / 1e9, 4), "mb_per_bag": round(byts / 1e6, 3),
            "batched_256": row(dt_b, B, "1/256 kernel + 1/256 memset"), "one_bag_per_call": row(dt_1, 1, "1 kernel + 1 memset"),
            "six_launch_form_one_bag_per_call": row(dt_6, 1, "6 kernels + 4 copies"), "rel_l2_fused_vs_six_launch": float(f"{rel:.2e}"),
            "finite": bool(torch.isfinite(ob).all())}


def secondary_metrics(ctx, a, tiles, is_swin) -> dict:
    """BASELINE.json's secondary metric, MIL bags/s (bags of 1024 x 1024-d, batch 64), and the in-tree tile encoder."""
    from stamp_amd.mil import TransMIL as HipTransMIL
    from stamp_amd.mil import VisionTransformer as HipMil
    from stamp_amd.mil_train import HipMilVitTrainer
    from stamp_amd.swin import SWIN_PRESETS, HipSwin, random_swin_state_dict

    sec = {}
    torch.manual_seed(1)
    kw = dict(dim_output=2, dim_input=1024, dim_model=512, n_layers=2, n_heads=8, dim_feedforward=512)
    mil = HipMil(dropout=0.25, use_alibi=False, **kw).eval()            # template default dropout (config.yaml:343)
    bags = torch.randn(64, 1024, 1024, generator=torch.Generator().manual_seed(1)).half().to(ctx.device)

    def timeit(fn, n, warm=2):
        for _ in range(warm):
            fn()
        torch.cuda.synchronize()
        t = time.perf_counter()
        for _ in range(n):
            r = fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t) / n, r

    with torch.no_grad():
        dt, lg = timeit(lambda: mil(bags, coords=None, mask=None), 30, warm=3)          # (1.1 ms per call: five calls were a 5 ms window behind one synchronize)
    sec.update({"metric": "MIL bags/s (vit head forward, bags of 1024 x 1024-d fp16, batch 64, no mask)", "value": round(64 / dt, 1), "unit": "bags/s",
                "gflop_per_bag_fwd": 11.83, "finite": bool(torch.isfinite(lg).all())})
    tg = torch.nn.functional.one_hot(torch.arange(64) % 2, 2).float()
    cw = torch.tensor([0.5, 0.5])
    # The trainer follows torch's float32_matmul_precision like the TransMIL products do; the reference sets "high" before training (src/stamp/modeling/train.py:519):
    # fp16 operands (TF32's 10 explicit mantissa bits) + a static loss scale -- the value of every leg; "medium" = bf16 operands (the only mode before round 6) beside it.
    for key, alibi, drop in (("train", False, None), ("train_no_dropout", False, False), ("train_alibi", True, None)):
        model = mil if not alibi else HipMil(dropout=0.25, use_alibi=True, **kw).eval()
        crd = (torch.rand(64, 1024, 2, generator=torch.Generator().manual_seed(2)) * 4e4).to(ctx.device) if alibi else None
        vals = {}
        for prec in ("high", "medium"):
            trn = HipMilVitTrainer(model, device=ctx.device, total_steps=100, sched_interval="step", dropout=drop, precision=prec)
            dt, (ltr, _) = timeit(lambda: trn.step(bags, tg, cw, coords=crd), 16, warm=3)       # 16 steps (0.12 s): four were dominated by the first step's clock ramp
            vals[prec] = (round(64 / dt, 1), bool(torch.isfinite(ltr)))
            del trn
        sec[key] = {"metric": f"MIL bags/s (vit head{' with ALiBi' if alibi else ''}, fwd + bwd + AdamW, bags of 1024 x 1024-d, batch 64, float32_matmul_precision 'high' as the "
                              "reference's train_model_ sets it: fp16 operands (10 explicit mantissa bits, the TF32 class), loss scale 2^10, fp32 accumulation / residual stream / "
                              "gradients; " + ("all dropout sites off)" if drop is False else "train-mode dropout as the reference: 0.25 / 0.25 / 0.5 / 0.5)"),
                    "value": vals["high"][0], "unit": "bags/s", "loss_finite": vals["high"][1], "medium_bf16_operands": vals["medium"][0],
                    "medium_loss_finite": vals["medium"][1]}
    # BASELINE.json configs[4], the buildable part: Cox-survival head at bag scale -- `vit` head with dim_output = 1 on bags of 1024 x 768-d
    # (CONCH1.5 / TITAN feature width), Efron partial likelihood as LitTileSurvival.training_step (models/__init__.py:751-776), targets as
    # tests/random_data.py:173-175; fwd + bwd + AdamW, train-mode dropout
    from stamp_amd import losses as L
    gs = torch.Generator().manual_seed(7)
    sbags = torch.randn(64, 1024, 768, generator=gs).half().to(ctx.device)
    stg = torch.stack([torch.rand(64, generator=gs) * 1970 + 30, (torch.rand(64, generator=gs) < 0.7).float()], 1)
    sv = HipMil(dropout=0.25, use_alibi=False, dim_output=1, dim_input=768, dim_model=512, n_layers=2, n_heads=8, dim_feedforward=512).eval()
    trn = HipMilVitTrainer(sv, device=ctx.device, total_steps=100, sched_interval="step", precision="high")
    dt, (lsv, _) = timeit(lambda: trn.step(sbags, stg, loss_fn=L.cox_survival_loss), 16, warm=3)
    sec["survival"] = {"metric": "MIL bags/s (Cox-survival `vit` head, dim_output 1, Efron partial likelihood, fwd + bwd + AdamW, bags of 1024 x 768-d, batch 64, "
                                 "float32_matmul_precision 'high': fp16 operands, loss scale 2^10; train-mode dropout)", "value": round(64 / dt, 1), "unit": "bags/s", "loss_finite": bool(torch.isfinite(lsv))}
    del trn, sbags
    tm = HipTransMIL(dim_output=2, dim_input=1024, dim_hidden=512).eval().to(ctx.device)
    bags_f = bags.float()
    # The library follows torch's own flag: the reference asks for torch.set_float32_matmul_precision("medium") before deployment (modeling/deploy.py:398) and
    # "high" before training (modeling/train.py:519); both legs are timed at the reference's setting ("high" here = operands as hi + lo bf16, three bf16 MFMAs
    # per product) and, beside it, at torch's default "highest" (exact fp32 MFMA products).
    from stamp_amd import ops as hip_ops
    with torch.no_grad():
        dt_hi, lg_hi = timeit(lambda: tm(bags_f), 6, warm=2)
        with hip_ops.float32_matmul_precision("high"):
            dt, lg2 = timeit(lambda: tm(bags_f), 6, warm=2)
    sec["transmil"] = {"metric": "TransMIL bags/s (forward, bags of 1024 x 1024-d, batch 64, float32_matmul_precision 'high' as the reference's deploy / train set it: bf16 x 3 products)",
                       "value": round(64 / dt, 1), "finite": bool(torch.isfinite(lg2).all()), "highest_exact_fp32": round(64 / dt_hi, 1),
                       "max_abs_logit_diff_high_vs_highest": float((lg2 - lg_hi).abs().max())}
    # TransMIL training (BASELINE.json configs[2]): fwd + hand-derived bwd + torch AdamW on the module's parameters, train mode (Dropout(0.1))
    tm.train()
    opt = torch.optim.AdamW(tm.parameters(), lr=1e-4)
    tgd = tg.to(ctx.device)

    def tm_step():
        opt.zero_grad(set_to_none=True)
        loss = torch.nn.functional.cross_entropy(tm(bags_f), tgd)
        loss.backward()
        opt.step()
        return loss.detach()       # NOT the attached loss: see below
    # Three blocks of 4 steps after two warm steps; every block is reported and `value` is their MEDIAN.  The slow blocks of earlier runs (345 / 446 / 472
    # bags/s against 1.1 k) were this function's doing, not the library's: it returned the ATTACHED loss, whose graph pins the step's 15 GB
    # saved-activation arena; the caller's reference from the previous block then forced a third arena, i.e. one 15 GB hipMalloc (260-450 ms,
    # tools/transmil_step_times.py) inside a timed 230 ms block.  A training loop that keeps only detached values needs one arena.
    blocks = []
    with hip_ops.float32_matmul_precision("high"):             # train.py:519
        for i in range(3):
            dt, ltm = timeit(tm_step, 4, warm=2 if i == 0 else 0)
            blocks.append(round(64 / dt, 1))
    dt_hi, _ = timeit(tm_step, 4, warm=1)
    sec["transmil_train"] = {"metric": "TransMIL bags/s (fwd + bwd + AdamW, bags of 1024 x 1024-d, batch 64, fp32 tensors, float32_matmul_precision 'high' as the reference's "
                                       "train_model_ sets it: bf16 x 3 products; Dropout(0.1) live; median of three 4-step blocks)",
                             "value": sorted(blocks)[1], "blocks_bags_per_s": blocks, "unit": "bags/s", "loss_finite": bool(torch.isfinite(ltm)),
                             "highest_exact_fp32": round(64 / dt_hi, 1)}
    del bags_f, opt
    # BASELINE.json configs[0] (tests/random_data.py shape): 64 patients x 256 tiles x 2048-d, binary `vit` head, two epochs of one
    # training step over the 51 training bags + 13 full-bag validation forwards through stamp_amd.mil_train.fit; wall seconds incl. the
    # trainer's construction (weight packing), against cpu_baseline.mil.config1_two_epochs
    from stamp_amd.mil_train import fit as mil_fit
    torch.manual_seed(3)
    t0c = time.perf_counter()
    c1 = HipMil(dim_output=2, dim_input=2048, dim_model=512, n_layers=2, n_heads=8, dim_feedforward=512, dropout=0.25, use_alibi=False)
    cb = torch.rand(64, 256, 2048, device=ctx.device).half()
    ct = torch.nn.functional.one_hot(torch.arange(64) % 2, 2).float()
    tr1 = HipMilVitTrainer(c1, device=ctx.device, total_steps=2, sched_interval="step", precision="high")
    hist = mil_fit(tr1, lambda: [(cb[:51], None, None, ct[:51])], lambda: [(cb[i:i + 1], None, None, ct[i:i + 1]) for i in range(51, 64)], max_epochs=2, patience=16)
    torch.cuda.synchronize()
    sec["config1_two_epochs"] = {"metric": "wall seconds, BASELINE.json configs[0]: 64 bags x 256 tiles x 2048-d, `vit` head, 2 epochs (train step over 51 bags + 13 validation forwards each)",
                                 "value": round(time.perf_counter() - t0c, 3), "unit": "s", "higher_is_better": False,
                                 "validation_loss_finite": bool(all(v == v for v in hist["validation_loss"]))}
    del cb, tr1
    # Gated-attention pooling (CHIEF, chief.py:74-89) on the GPU: bags of 1024 x 768 fp32 -- the workload of cpu_baseline.mil.gated_attention_pool.
    # One bag per call (what the reference's per-slide loop issues), 256 bags per launch (amds_gated_attn_pool_batched), and the six-launch form the fused
    # kernel replaced; exact fp32 throughout (v_mfma_f32_16x16x4_f32: 157.3 TF peak).  Algorithmic work per bag: 1.342 GFLOP, 3.146 MB (x read once).
    try:
        sec["gated_attention_pool"] = gated_pool_leg(ctx.device)
    except Exception as e:
        sec["gated_attention_pool"] = {"error": repr(e)[:300]}
    if not is_swin:     # the reference's in-tree tile encoder, same tile shape
        scfg = SWIN_PRESETS["ctranspath"]
        sw = HipSwin(scfg, random_swin_state_dict(scfg, 0), device=ctx.device, chunk=a.swin_chunk)
        st_tiles = tiles[:1024] if tiles.shape[0] >= 1024 else tiles
        dt, sf = timeit(lambda: sw(st_tiles), 3, warm=1)
        sec["ctranspath"] = {"metric": "tiles/sec encoded (224x224, CTransPath = ConvStem + Swin-T)", "value": round(st_tiles.shape[0] / dt, 1),
                             "unit": "tiles/s", "gflop_per_tile": round(scfg.matmul_flops_per_tile() / 1e9, 3), "finite": bool(torch.isfinite(sf.float()).all())}
    return sec


def dry_run(a, ctx, D) -> None:
    """The multi-rank control flow of main() with a stand-in encoder (a fixed random projection on the CPU): same step(), same barrier +
    max-over-ranks timing, same line keys.  Asserts that every rank's slide id landed in the gathered table."""
    from stamp_amd.vit import PRESETS
    cfg = PRESETS[a.model]
    g = torch.Generator().manual_seed(1234 + ctx.rank)
    tiles = torch.randint(0, 256, (min(a.tiles, 8), 16, 16, 3), dtype=torch.uint8, generator=g)
    proj = torch.randn(16 * 16 * 3, 32, generator=torch.Generator().manual_seed(0))
    slide_ids = torch.tensor([ctx.rank])

    def step():
        feats = (tiles.reshape(tiles.shape[0], -1).float() @ proj).half()
        if ctx.world > 1:
            emb = feats.float().mean(dim=0, keepdim=True)
            return D.gather_slide_embeddings(ctx, emb, slide_ids, ctx.world)
        return feats

    for _ in range(a.warmup):
        step()
    D.barrier(ctx)
    t0 = time.perf_counter()
    for _ in range(a.steps):
        out = step()
    D.barrier(ctx)
    elapsed = D.max_over_ranks(ctx, time.perf_counter() - t0)
    if ctx.world > 1:
        mine = (tiles.reshape(tiles.shape[0], -1).float() @ proj).half().float().mean(dim=0)
        assert out.shape == (ctx.world, 32) and torch.equal(out[ctx.rank], mine), "this rank's slide embedding is not in the gathered table"
        assert bool((out.abs().sum(dim=1) > 0).all()), "a rank's row of the gathered table is empty"
    line = {"metric": "tiles/sec encoded (224x224, ViT-L/14)", "value": None, "unit": "tiles/s", "n_gpus": ctx.world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": round(elapsed / max(a.steps, 1) * 1e3, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": a.act,
            "data": "synthetic", "dry_run": True, "config": {"workload": "DRY RUN on CPU ranks (gloo) with a stand-in encoder: control flow only, no measurement",
                                                            "model": a.model, "gflop_per_tile": round(cfg.matmul_flops_per_tile() / 1e9, 3),
                                                            "parallelism": f"slide-sharded x{ctx.world}, all-gather of slide embeddings" if ctx.world > 1 else "single rank"},
            "roofline": None, "cpu_baseline": None}
    if ctx.is_main:
        print(json.dumps(line), flush=True)
    if ctx.world > 1:
        torch.distributed.destroy_process_group()


def main() -> None:
    a = parse()
    from stamp_amd import _lib, distributed as D
    from stamp_amd.swin import SWIN_PRESETS, HipSwin, random_swin_state_dict
    from stamp_amd.vit import PRESETS, HipViT, random_vit_state_dict

    # AMDS_BENCH_DRYRUN=1: exercise THIS file's control flow (argument handling, the N > 1 branch with its all-gather, barriers,
    # max-over-ranks timing, the JSON line) on CPU ranks over gloo with a stand-in for the encoder -- so that the first 8-GPU driver run cannot
    # die on a Python error (tests/test_cpu_host.py).  The line it prints says "dry_run": true and carries no measurement.
    dry = os.environ.get("AMDS_BENCH_DRYRUN") == "1"
    ctx = D.init_from_env(prefer_gpu=not dry)
    if ctx.device.type != "cuda" and not dry:
        raise SystemExit("bench.py needs a GPU (the product path has no CPU fallback)")
    if a.gpus != ctx.world:
        raise SystemExit(f"--gpus {a.gpus} but WORLD_SIZE={ctx.world}: launch with torch.distributed.run")
    if dry:
        return dry_run(a, ctx, D)
    act = torch.float16 if a.act == "f16" else torch.bfloat16
    is_swin = a.model in SWIN_PRESETS
    if is_swin:       # the reference's in-tree tile encoder (ctranspath.py): ConvStem + Swin-T
        cfg = SWIN_PRESETS[a.model]
        sd = random_swin_state_dict(cfg, seed=0)
        model = HipSwin(cfg, sd, device=ctx.device, act_dtype=act, chunk=a.swin_chunk)
    else:
        cfg = PRESETS[a.model]
        sd = random_vit_state_dict(cfg, seed=0, init="moderate")
        model = HipViT(cfg, sd, device=ctx.device, act_dtype=act, chunk=a.chunk, exact=a.exact, fp8=a.fp8, check=a.check)
        model.overlap = bool(a.overlap)
    g = torch.Generator().manual_seed(1234 + ctx.rank)
    tiles = torch.randint(0, 256, (a.tiles, cfg.img, cfg.img, 3), dtype=torch.uint8, generator=g).to(ctx.device)
    slide_ids = torch.tensor([ctx.rank], device=ctx.device)

    def step() -> torch.Tensor:
        feats = model(tiles)                                        # fp16 [tiles, D] in HBM
        if ctx.world > 1:                                           # collate slide-level embeddings (RCCL)
            emb = feats.float().mean(dim=0, keepdim=True)
            return D.gather_slide_embeddings(ctx, emb, slide_ids, ctx.world)
        return feats

    lib = _lib.lib()
    actx = _lib.ctx(ctx.device.index or 0)
    for _ in range(a.warmup):
        step()
    lib.amds_profile_reset(actx)
    lib.amds_profile_enable(actx, 1)
    sampler = ClockPowerSampler(int(os.environ.get("LOCAL_RANK", "0")))
    D.barrier(ctx)
    torch.cuda.synchronize()
    with sampler:
        t0 = time.perf_counter()
        for _ in range(a.steps):
            out = step()
        torch.cuda.synchronize()
        D.barrier(ctx)
        elapsed = time.perf_counter() - t0
    lib.amds_profile_enable(actx, 0)
    elapsed = D.max_over_ranks(ctx, elapsed)
    assert torch.isfinite(out.float()).all()
    clock0 = sampler.summary()

    def headline_block() -> dict:
        """The headline region again (same K steps, same synchronisation), for the per-block list: single-GPU line only."""
        torch.cuda.synchronize()
        with sampler:
            t = time.perf_counter()
            for _ in range(a.steps):
                step()
            torch.cuda.synchronize()
            el = time.perf_counter() - t
        return {"tiles_per_s": round(a.tiles * a.steps / el, 1), **sampler.summary()}

    # roofline of the dominant kernel (the MFMA GEMM), from HIP events recorded around every launch of it
    ms, n, work = C.c_double(), C.c_long(), C.c_double()
    kinds = {}
    for kind, name in ((0, "gemm"), (1, "attention"), (2, "layernorm"), (3, "im2col_or_stem")):
        _lib.check(lib.amds_profile_read(actx, kind, C.byref(ms), C.byref(n), C.byref(work)), "profile_read")
        kinds[name] = (ms.value, n.value, work.value)
    gms, gn, gflop = kinds["gemm"]
    achieved = gflop / (gms * 1e-3) / 1e12 if gms > 0 else 0.0
    total_tiles = a.tiles * a.steps * ctx.world
    value = total_tiles / elapsed
    # HBM traffic of the dominant kernel: PMC counters cannot be read from inside the process; the committed summary of the
    # two rocprofv3 --pmc passes over this same workload (profiles/*_pmc_gemm_traffic.json, tools/pmc_summary.py) is reported
    traffic, pmc_name = None, None
    for pmc_file in (ROOT / "profiles" / "r06_pmc_gemm_traffic.json", ROOT / "profiles" / "r05_pmc_gemm_traffic.json", ROOT / "profiles" / "r04_pmc_gemm_traffic.json", ROOT / "profiles" / "r03_pmc_gemm_traffic.json", ROOT / "profiles" / "r02_pmc_gemm_traffic.json", ROOT / "profiles" / "r01_pmc_gemm_traffic.json"):
        if not is_swin and a.model == "vit_large_patch14_224" and a.chunk == 1020 and pmc_file.is_file():
            try:
                traffic, pmc_name = json.loads(pmc_file.read_text())["traffic_bytes_per_launch"], pmc_file.name
                break
            except Exception:
                traffic = None
    tail_on = (not is_swin) and getattr(model, "cls_tail", False) and os.environ.get("AMDS_VIT_CLS_TAIL", "1") != "0"
    flops_exec = cfg.matmul_flops_per_tile() - (cfg.matmul_flops_skipped_by_cls_tail() if tail_on else 0.0)
    alg_bytes = None
    planes_on = False
    if not is_swin:
        # algorithmic HBM bytes per GEMM launch, averaged over the four launches of a block (operands once, fp32 residual rows read +
        # written, outputs once; with the LayerNorm folded in, proj / fc2 also write the 16-bit copy of the rows)
        Mrows, Dm, Hd = min(a.chunk, a.tiles) * cfg.tokens, cfg.dim, cfg.hidden_pad
        n1 = Hd * (2 if cfg.mlp == "swiglu" else 1)
        planes_on = getattr(model, "ln_fold", False) and not a.exact and act == torch.float16 and os.environ.get("AMDS_VIT_PLANES", "1") != "0"
        # residual update: fp32 rows read + written (8 B per element) plus, folded, the 16-bit copy (2 B); as two fp16 planes: 4 B read + 4 B written
        fold = 2 * Mrows * Dm if (getattr(model, "ln_fold", False) and not planes_on) else 0
        per = [2 * Mrows * Dm + 2 * 3 * Dm * Dm + 2 * Mrows * 3 * Dm, 2 * Mrows * Dm + 2 * Dm * Dm + 8 * Mrows * Dm + fold,
               2 * Mrows * Dm + 2 * n1 * Dm + 2 * Mrows * Hd, 2 * Mrows * Hd + 2 * Dm * Hd + 8 * Mrows * Dm + fold]
        alg_bytes = sum(per) / 4
    line = {
        "metric": "tiles/sec encoded (224x224, ViT-L/14)" if a.model == "vit_large_patch14_224" else f"tiles/sec encoded (224x224, {a.model})", "value": round(value, 2), "unit": "tiles/s",
        "n_gpus": ctx.world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(elapsed / a.steps * 1e3, 3),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": a.act, "data": "synthetic",
        "config": {"workload": ("CTransPath (ConvStem + Swin-T 96/(2,2,6,2)/window 7, the reference's in-tree tile encoder) on "
                                "synthetic 224x224x3 u8 tiles resident in HBM, seeded weights, fp16 768-d features out") if is_swin else
                               ("BASELINE.json configs[1]: ViT-L/14 (dim 1024, depth 24, 16 heads, 257 tokens, GELU MLP, "
                                "LayerScale) tile extraction on synthetic 224x224x3 u8 tiles resident in HBM, "
                                "random-init weights, fp16 CLS features out" + (", exact class-token rows" if a.exact else "") + (", OPT-IN fp8 (e4m3) GEMM operands" if a.fp8 else "")),
                   "model": a.model, "tiles_per_step_per_gpu": a.tiles, "chunk": a.swin_chunk if is_swin else a.chunk,
                   "operands": a.act, "accumulate": "f32",
                   "parallelism": f"slide-sharded x{ctx.world}, all-gather of slide embeddings" if ctx.world > 1 else "single GPU",
                   "gflop_per_tile": round(cfg.matmul_flops_per_tile() / 1e9, 3),
                   # the last block's class-row tail (include/amdstamp.h amds_vit_weights.cls_tail): rows of the last block that nothing reads are not
                   # computed, so the whole-path fraction is priced on the products actually EXECUTED, not on the network's nominal count
                   # the guard in front of the feature file runs INSIDE the timed region: amds_check_finite + the range counters of the folded LayerNorms,
                   # one 4-byte read-back and stream synchronisation per call (stamp_amd.vit.HipViT.check; default "fallback")
                   "guard": None if is_swin else f"check={model.check!r}: non-finite / |mean| > 8 sigma check per call, in the timed region; safe level {model.safe_level}",
                   "cls_tail": bool(tail_on), "residual_stream": "f16 hi|lo planes" if (not is_swin and planes_on) else "f32",
                   "gflop_per_tile_executed": round(flops_exec / 1e9, 3),
                   "whole_path_mfma_frac": round(value / ctx.world * flops_exec / 1e12 / MFMA_PEAK_TFLOPS, 4),
                   # shader clock / socket power of rank 0's GPU during the timed region (hwmon freq1_input / power1_input every 20 ms): `value` above is
                   # block 0; the single-GPU line re-times the same K steps after the end-to-end leg and at the very end (blocks_tiles_per_s, median_block)
                   "sclk_mhz": clock0["sclk_mhz"], "sclk_mhz_min": clock0["sclk_mhz_min"], "socket_w": clock0["socket_w"], "socket_w_max": clock0["socket_w_max"],
                   "clock_power_samples": clock0["samples"], "blocks_tiles_per_s": [round(value / ctx.world, 1)], "blocks": [{"tiles_per_s": round(value / ctx.world, 1), **clock0}]},
        "roofline": {"kernel": "MFMA GEMMs (gemm_tn_kernel 128x96 / 128x128 tiles where N is not a multiple of 256, gemm_4w16_kernel in stage 4 and the stage-3 MLP)" if is_swin else
                               ("gemm_4w16_kernel (256x256x64 tiles, 4 waves with 128x128 wave tiles, v_mfma_f32_16x16x32, buffer-form LDS-DMA, fused LDS-staged epilogues"
                                + ("; LayerNorm folded in: proj / fc2 also emit row sums, qkv / fc1 apply the row statistics" if getattr(model, "ln_fold", False) else "")
                                + ("; residual stream as two fp16 planes (hi = the next GEMM's A operand, lo = x - hi) updated in place by proj / fc2" if planes_on else
                                   ("; proj / fc2 update fp32 rows and write their 16-bit copy" if getattr(model, "ln_fold", False) else "")) + ")"), "bound": "mfma",
                     "achieved": round(achieved, 1), "peak": MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                     "frac": round(achieved / MFMA_PEAK_TFLOPS, 4), "traffic": traffic,
                     "traffic_note": (f"HBM-side bytes per GEMM launch (FETCH_SIZE x2 + WRITE_SIZE, separate rocprofv3 --pmc passes, profiles/{pmc_name}); "
                                      f"algorithmic {alg_bytes:.3g} at this chunk ({min(a.chunk, a.tiles)} tiles)") if traffic else None,
                     "algorithmic_bytes_per_launch": round(alg_bytes) if alg_bytes else None,
                     "launches": gn, "avg_launch_us": round(gms / max(gn, 1) * 1e3, 2),
                     "avg_gflop_per_launch": round(gflop / max(gn, 1) / 1e9, 2),
                     "time_share": {k: round(v[0] / (elapsed * 1e3), 4) for k, v in kinds.items()}},
    }
    # Everything below is single-GPU-line only: these blocks are wrapped in try/except, and a rank that raised while the others
    # wait in a collective would hang the N-GPU scaling run, whose purpose is the headline value.
    single = ctx.world == 1 and not a.no_secondary
    if single and a.e2e_tiles > 0:
        try:
            line["end_to_end"] = end_to_end_leg(model, cfg, ctx.device, a.e2e_tiles, a.swin_chunk if is_swin else a.chunk, a.e2e_warmup)
        except Exception as e:
            line["end_to_end"] = {"error": repr(e)[:300]}

    def add_block() -> None:
        try:
            b = headline_block()
            line["config"]["blocks"].append(b)
            line["config"]["blocks_tiles_per_s"].append(b["tiles_per_s"])
        except Exception as e:
            line["config"]["blocks"].append({"error": repr(e)[:200]})
    if single:
        add_block()          # block 1: after the end-to-end leg
    if single and not is_swin and not a.exact:
        # the opt-in exact class-token mode on the same workload (HipViT(exact=True), csrc/vit_exact.hip): what it costs, next to what it buys
        # (tests/test_gpu_vit.py: fp16 CLS feature error 5-8e-4 -> 3-4e-4)
        try:
            mx = HipViT(cfg, sd, device=ctx.device, act_dtype=act, chunk=a.chunk, exact=True)
            mx(tiles)
            torch.cuda.synchronize()
            t0x = time.perf_counter()
            for _ in range(3):
                fx = mx(tiles)
            torch.cuda.synchronize()
            elx = time.perf_counter() - t0x
            rel = ((fx.float() - out.float()).norm() / out.float().norm()).item()
            line["exact_mode"] = {"metric": "tiles/s with the class-token rows also carried on an exact-fp32 stream (opt-in)", "value": round(3 * a.tiles / elx, 1),
                                  "unit": "tiles/s", "vs_default": round(3 * a.tiles / elx / value, 4), "rel_l2_vs_default_features": float(f"{rel:.3e}"),
                                  "finite": bool(torch.isfinite(fx.float()).all())}
            del mx, fx
        except Exception as e:
            line["exact_mode"] = {"error": repr(e)[:300]}
    if single and tail_on:
        # the same workload with the WHOLE last block computed (cls_tail=False): what the class-row tail removes, and that the features agree
        try:
            mf = HipViT(cfg, sd, device=ctx.device, act_dtype=act, chunk=a.chunk, exact=a.exact, fp8=a.fp8, cls_tail=False)
            mf(tiles)
            torch.cuda.synchronize()
            t0f = time.perf_counter()
            for _ in range(3):
                ff = mf(tiles)
            torch.cuda.synchronize()
            elf = time.perf_counter() - t0f
            rel = ((ff.float() - out.float()).norm() / ff.float().norm()).item()
            line["full_last_block"] = {"metric": "tiles/s with every row of the last block computed (cls_tail=False)", "value": round(3 * a.tiles / elf, 1), "unit": "tiles/s",
                                       "gflop_per_tile_executed": round(cfg.matmul_flops_per_tile() / 1e9, 3),
                                       "whole_path_mfma_frac": round(3 * a.tiles / elf * cfg.matmul_flops_per_tile() / 1e12 / MFMA_PEAK_TFLOPS, 4),
                                       "default_vs_this": round(value / (3 * a.tiles / elf), 4), "rel_l2_default_vs_this_features": float(f"{rel:.3e}")}
            del mf, ff
        except Exception as e:
            line["full_last_block"] = {"error": repr(e)[:300]}
    if single and not is_swin and getattr(model, "ln_fold", False) and not a.exact and act == torch.float16 and os.environ.get("AMDS_VIT_PLANES", "1") != "0":
        # the same model with the residual stream as fp32 rows + a 16-bit copy (AMDS_VIT_PLANES=0, read per call) instead of the two fp16 planes
        try:
            os.environ["AMDS_VIT_PLANES"] = "0"
            model(tiles)
            torch.cuda.synchronize()
            t0r = time.perf_counter()
            for _ in range(3):
                fr = model(tiles)
            torch.cuda.synchronize()
            elr = time.perf_counter() - t0r
            rel = ((fr.float() - out.float()).norm() / fr.float().norm()).item()
            line["residual_fp32_rows"] = {"metric": "tiles/s with the residual stream as fp32 rows + a 16-bit copy (AMDS_VIT_PLANES=0) instead of two fp16 planes",
                                          "value": round(3 * a.tiles / elr, 1), "unit": "tiles/s", "default_vs_this": round(value / (3 * a.tiles / elr), 4),
                                          "rel_l2_default_vs_this_features": float(f"{rel:.3e}")}
            del fr
        except Exception as e:
            line["residual_fp32_rows"] = {"error": repr(e)[:300]}
        finally:
            os.environ.pop("AMDS_VIT_PLANES", None)
    if single and not is_swin and a.slide_tiles > 0:
        try:
            line["slide_synthetic"] = slide_leg(model, ctx.device, a.slide_tiles)
            line["slide_synthetic"]["vs_hbm_resident"] = round(line["slide_synthetic"]["value"] / value, 4)
            line["slide_synthetic"]["vs_min_of_reader_and_encoder"] = round(line["slide_synthetic"]["value"] / min(value, line["slide_synthetic"]["reader_only"]), 4)
        except Exception as e:
            line["slide_synthetic"] = {"error": repr(e)[:300]}
        try:
            line["slides_synthetic"] = slides_leg(model, ctx.device)
            line["slides_synthetic"]["vs_hbm_resident"] = round(line["slides_synthetic"]["value"] / value, 4)
        except Exception as e:
            line["slides_synthetic"] = {"error": repr(e)[:300]}
    if single and not is_swin and not a.exact and not a.fp8 and cfg.dim % 256 == 0 and cfg.hidden % 256 == 0:
        # the opt-in fp8 variant (BASELINE.json configs[4]: "fp8 MFMA weights") on the same workload: HipViT(fp8=True), csrc/gemm_fp8.hip; its own
        # roofline fraction is against the 5 PFLOP/s dense fp8 peak; what it costs in accuracy is in tests/test_gpu_fp8.py / DESIGN.md section 5
        try:
            m8 = HipViT(cfg, sd, device=ctx.device, act_dtype=act, chunk=a.chunk, fp8=True)
            m8(tiles)
            lib.amds_profile_reset(actx)
            lib.amds_profile_enable(actx, 1)
            torch.cuda.synchronize()
            t08 = time.perf_counter()
            for _ in range(3):
                f8 = m8(tiles)
            torch.cuda.synchronize()
            el8 = time.perf_counter() - t08
            lib.amds_profile_enable(actx, 0)
            _lib.check(lib.amds_profile_read(actx, 5, C.byref(ms), C.byref(n), C.byref(work)), "profile_read")
            tf8 = work.value / (ms.value * 1e-3) / 1e12 if ms.value > 0 else 0.0
            rel8 = ((f8.float() - out.float()).norm() / out.float().norm()).item()
            line["fp8_mode"] = {"metric": "tiles/s with the blocks' four Linears on the fp8 MFMA (e4m3 operands, per-row / per-channel scales; OPT-IN, never the default)",
                                "value": round(3 * a.tiles / el8, 1), "unit": "tiles/s", "vs_default": round(3 * a.tiles / el8 / value, 4),
                                "rel_l2_vs_default_features": float(f"{rel8:.3e}"),
                                "roofline": {"kernel": "gemm_fp8_kernel (v_mfma_f32_16x16x128_f8f6f4)", "bound": "mfma", "achieved": round(tf8, 1), "peak": 5000.0,
                                             "unit": "TFLOP/s", "frac": round(tf8 / 5000.0, 4), "launches": n.value},
                                "finite": bool(torch.isfinite(f8.float()).all())}
            del m8, f8
        except Exception as e:
            line["fp8_mode"] = {"error": repr(e)[:300]}
    if single:
        try:
            line["drop_in_b64"] = drop_in_b64_leg(model, cfg, ctx.device)
        except Exception as e:
            line["drop_in_b64"] = {"error": repr(e)[:300]}
        try:
            line["secondary"] = secondary_metrics(ctx, a, tiles, is_swin)
        except Exception as e:      # the headline metric must still be printed
            line["secondary"] = {"error": repr(e)[:300]}
        add_block()          # block 2: the last GPU work of the run
        bl = sorted(line["config"]["blocks_tiles_per_s"])
        line["config"]["median_block_tiles_per_s"] = bl[len(bl) // 2] if bl else None
    if ctx.is_main and ctx.world == 1 and not a.no_cpu_baseline:
        line["cpu_baseline"] = cpu_baseline(cfg, sd, a.cpu_seconds, swin=is_swin)
        if not a.no_secondary:
            try:
                line["cpu_baseline"]["mil"] = cpu_baseline_mil()
            except Exception as e:
                line["cpu_baseline"]["mil"] = {"error": repr(e)[:300]}
    elif ctx.is_main:
        line["cpu_baseline"] = None
    if ctx.is_main:
        print(json.dumps(line), flush=True)
    if ctx.world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
