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
of the slide-level embeddings.  Rank 0 prints ONE JSON line.
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
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true", help="headline workload only (for a ViT-only rocprofv3 kernel trace)")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    return ap.parse_args()


def cpu_baseline(cfg, sd, seconds: float, swin: bool = False) -> dict:
    """The oracle (= the reference's algorithm on torch CPU fp32 kernels, batch 64 like the reference's
    DataLoader, src/stamp/preprocessing/__init__.py:317) timed on this box's host cores on a bounded sample."""
    if swin:
        from oracle.swin_ctranspath import swin_encode_f16 as extract_features
    else:
        from oracle.vit_tile_encoder import extract_features

    threads = torch.get_num_threads()
    g = torch.Generator().manual_seed(1234)
    batch = 16
    tiles = torch.randint(0, 256, (batch, cfg.img, cfg.img, 3), dtype=torch.uint8, generator=g)
    extract_features(tiles[:2], sd, cfg)           # warm
    n, t0 = 0, time.perf_counter()
    while True:
        extract_features(tiles, sd, cfg)
        n += batch
        el = time.perf_counter() - t0
        if el >= seconds or n >= 512:
            break
    return {"value": round(n / el, 3), "unit": "tiles/s", "cores": threads, "kind": "port",
            "sample": f"{n} synthetic 224x224 tiles, {'CTransPath (Swin-T)' if swin else 'ViT'} fp32 oracle (torch CPU), batches of {batch}, {el:.1f}s"}


def main() -> None:
    a = parse()
    from stamp_amd import _lib, distributed as D, ops
    from stamp_amd.vit import PRESETS, HipViT, random_vit_state_dict

    ctx = D.init_from_env()
    if ctx.device.type != "cuda":
        raise SystemExit("bench.py needs a GPU (the product path has no CPU fallback)")
    if a.gpus != ctx.world:
        raise SystemExit(f"--gpus {a.gpus} but WORLD_SIZE={ctx.world}: launch with torch.distributed.run")
    from stamp_amd.swin import SWIN_PRESETS, HipSwin, random_swin_state_dict
    act = torch.float16 if a.act == "f16" else torch.bfloat16
    is_swin = a.model in SWIN_PRESETS
    if is_swin:       # the reference's in-tree tile encoder (ctranspath.py): ConvStem + Swin-T
        cfg = SWIN_PRESETS[a.model]
        sd = random_swin_state_dict(cfg, seed=0)
        model = HipSwin(cfg, sd, device=ctx.device, act_dtype=act, chunk=a.swin_chunk)
    else:
        cfg = PRESETS[a.model]
        sd = random_vit_state_dict(cfg, seed=0, init="moderate")
        model = HipViT(cfg, sd, device=ctx.device, act_dtype=act, chunk=a.chunk)
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
    for _ in range(a.warmup):
        step()
    lib.amds_profile_reset()
    lib.amds_profile_enable(1)
    D.barrier(ctx)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        out = step()
    torch.cuda.synchronize()
    D.barrier(ctx)
    elapsed = time.perf_counter() - t0
    lib.amds_profile_enable(0)
    elapsed = D.max_over_ranks(ctx, elapsed)
    assert torch.isfinite(out.float()).all()

    # roofline of the dominant kernel (the MFMA GEMM), from HIP events recorded around every launch of it
    ms, n, work = C.c_double(), C.c_long(), C.c_double()
    kinds = {}
    for kind, name in ((0, "gemm"), (1, "attention"), (2, "layernorm"), (3, "im2col_or_stem")):
        _lib.check(lib.amds_profile_read(kind, C.byref(ms), C.byref(n), C.byref(work)), "profile_read")
        kinds[name] = (ms.value, n.value, work.value)
    gms, gn, gflop = kinds["gemm"]
    achieved = gflop / (gms * 1e-3) / 1e12 if gms > 0 else 0.0
    total_tiles = a.tiles * a.steps * ctx.world
    value = total_tiles / elapsed
    # HBM traffic of the dominant kernel: PMC counters cannot be read from inside the process; the committed summary of the
    # two rocprofv3 --pmc passes over this same workload (profiles/r01_pmc_gemm_traffic.json, tools/pmc_summary.py) is reported
    traffic = None
    pmc_file = ROOT / "profiles" / "r01_pmc_gemm_traffic.json"
    if not is_swin and a.model == "vit_large_patch14_224" and a.chunk == 1020 and pmc_file.is_file():
        try:
            traffic = json.loads(pmc_file.read_text())["traffic_bytes_per_launch"]
        except Exception:
            traffic = None
    line = {
        "metric": "tiles/sec encoded (224x224, ViT-L/14)" if a.model == "vit_large_patch14_224" else f"tiles/sec encoded (224x224, {a.model})", "value": round(value, 2), "unit": "tiles/s",
        "n_gpus": ctx.world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(elapsed / a.steps * 1e3, 3),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": a.act, "data": "synthetic",
        "config": {"workload": ("CTransPath (ConvStem + Swin-T 96/(2,2,6,2)/window 7, the reference's in-tree tile encoder) on "
                                "synthetic 224x224x3 u8 tiles resident in HBM, seeded weights, fp16 768-d features out") if is_swin else
                               ("BASELINE.json configs[1]: ViT-L/14 (dim 1024, depth 24, 16 heads, 257 tokens, GELU MLP, "
                                "LayerScale) tile extraction on synthetic 224x224x3 u8 tiles resident in HBM, "
                                "random-init weights, fp16 CLS features out"),
                   "model": a.model, "tiles_per_step_per_gpu": a.tiles, "chunk": a.swin_chunk if is_swin else a.chunk,
                   "operands": a.act, "accumulate": "f32", "residual_stream": "f32",
                   "parallelism": f"slide-sharded x{ctx.world}, all-gather of slide embeddings" if ctx.world > 1 else "single GPU",
                   "gflop_per_tile": round(cfg.matmul_flops_per_tile() / 1e9, 3),
                   "whole_path_mfma_frac": round(value / ctx.world * cfg.matmul_flops_per_tile() / 1e12 / MFMA_PEAK_TFLOPS, 4)},
        "roofline": {"kernel": "MFMA GEMMs (gemm_tn_kernel 128x96 / 128x128 tiles where N is not a multiple of 256, gemm_4w16_kernel in stage 4 and the stage-3 MLP)" if is_swin else
                               "gemm_4w16_kernel (256x256x64 tiles, 4 waves with 128x128 wave tiles, v_mfma_f32_16x16x32, buffer-form LDS-DMA, fused LDS-staged epilogues)", "bound": "mfma",
                     "achieved": round(achieved, 1), "peak": MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                     "frac": round(achieved / MFMA_PEAK_TFLOPS, 4), "traffic": traffic,
                     "traffic_note": "HBM-side bytes per GEMM launch (FETCH_SIZE x2 + WRITE_SIZE, separate rocprofv3 --pmc passes, profiles/r01_pmc_gemm_traffic.json); algorithmic 2.97e9 at this chunk (1020 tiles, M = 262140) -> 1.43x (A panels re-fetched across N tiles, served by L2/MALL)" if traffic else None,
                     "launches": gn, "avg_launch_us": round(gms / max(gn, 1) * 1e3, 2),
                     "avg_gflop_per_launch": round(gflop / max(gn, 1) / 1e9, 2),
                     "time_share": {k: round(v[0] / (elapsed * 1e3), 4) for k, v in kinds.items()}},
    }
    # secondary metric of BASELINE.json: MIL bags/s (vit head, deploy-time forward, bags of 1024 x 1024-d, batch 64).
    # Single-GPU line only: these blocks contain rank collectives (max_over_ranks) inside a try/except, and a rank that raised
    # while the others wait in a collective would hang the N-GPU scaling run, whose purpose is the headline value.
    try:
        if ctx.world > 1 or a.no_secondary:
            raise RuntimeError("secondary metrics are reported on the single-GPU line" if ctx.world > 1 else "--no-secondary")
        from stamp_amd.mil import VisionTransformer as HipMil
        torch.manual_seed(1)
        mil = HipMil(dim_output=2, dim_input=1024, dim_model=512, n_layers=2, n_heads=8, dim_feedforward=512,
                     dropout=0.0, use_alibi=False).eval()
        bags = torch.randn(64, 1024, 1024, generator=torch.Generator().manual_seed(1)).half().to(ctx.device)
        with torch.no_grad():
            for _ in range(2):
                mil(bags, coords=None, mask=None)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for _ in range(5):
                lg = mil(bags, coords=None, mask=None)
            torch.cuda.synchronize()
        dt_mil = D.max_over_ranks(ctx, (time.perf_counter() - t1) / 5)
        line["secondary"] = {"metric": "MIL bags/s (vit head forward, bags of 1024 x 1024-d fp16, batch 64, no mask)",
                             "value": round(64 * ctx.world / dt_mil, 1), "unit": "bags/s", "gflop_per_bag_fwd": 11.83,
                             "finite": bool(torch.isfinite(lg).all())}
        # training step (fwd + bwd + AdamW) of the same head, batch 64 (reference default, modeling/config.py:153)
        from stamp_amd.mil_train import HipMilVitTrainer
        trn = HipMilVitTrainer(mil, device=ctx.device, total_steps=100)
        tg = torch.nn.functional.one_hot(torch.arange(64) % 2, 2).float()
        cw = torch.tensor([0.5, 0.5])
        for _ in range(2):
            trn.step(bags, tg, cw)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(4):
            ltr, _ = trn.step(bags, tg, cw)
        torch.cuda.synchronize()
        dt_tr = D.max_over_ranks(ctx, (time.perf_counter() - t1) / 4)
        line["secondary"]["train"] = {"metric": "MIL bags/s (vit head, fwd + bwd + AdamW, bags of 1024 x 1024-d, batch 64, bf16 operands)",
                                      "value": round(64 * ctx.world / dt_tr, 1), "unit": "bags/s", "loss_finite": bool(torch.isfinite(ltr))}
        del trn
        # the same head with use_alibi=True (MultiHeadALiBi: post-softmax distance bias, train-mode running-mean scalers)
        mil_a = HipMil(dim_output=2, dim_input=1024, dim_model=512, n_layers=2, n_heads=8, dim_feedforward=512, dropout=0.0, use_alibi=True).eval()
        crd = (torch.rand(64, 1024, 2, generator=torch.Generator().manual_seed(2)) * 4e4).to(ctx.device)
        trn = HipMilVitTrainer(mil_a, device=ctx.device, total_steps=100)
        for _ in range(2):
            trn.step(bags, tg, cw, coords=crd)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(4):
            lta, _ = trn.step(bags, tg, cw, coords=crd)
        torch.cuda.synchronize()
        dt_ta = D.max_over_ranks(ctx, (time.perf_counter() - t1) / 4)
        line["secondary"]["train_alibi"] = {"metric": "MIL bags/s (vit head with ALiBi, fwd + bwd + AdamW, bags of 1024 x 1024-d + coords, batch 64, bf16 operands)",
                                            "value": round(64 * ctx.world / dt_ta, 1), "unit": "bags/s", "loss_finite": bool(torch.isfinite(lta))}
        del trn, mil_a
        from stamp_amd.mil import TransMIL as HipTransMIL
        tm = HipTransMIL(dim_output=2, dim_input=1024, dim_hidden=512).eval().to(ctx.device)
        bags_f = bags.float()
        with torch.no_grad():
            tm(bags_f)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for _ in range(3):
                lg2 = tm(bags_f)
            torch.cuda.synchronize()
        dt_tm = D.max_over_ranks(ctx, (time.perf_counter() - t1) / 3)
        line["secondary"]["transmil"] = {"metric": "TransMIL bags/s (forward, bags of 1024 x 1024-d, batch 64, exact-fp32 MFMA)",
                                         "value": round(64 * ctx.world / dt_tm, 1), "finite": bool(torch.isfinite(lg2).all())}
        del bags_f
        if not is_swin:     # the reference's in-tree tile encoder, same tile shape
            scfg = SWIN_PRESETS["ctranspath"]
            sw = HipSwin(scfg, random_swin_state_dict(scfg, 0), device=ctx.device, chunk=a.swin_chunk)
            st_tiles = tiles[:1024] if tiles.shape[0] >= 1024 else tiles
            sw(st_tiles)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for _ in range(3):
                sf = sw(st_tiles)
            torch.cuda.synchronize()
            dt_sw = D.max_over_ranks(ctx, (time.perf_counter() - t1) / 3)
            line["secondary"]["ctranspath"] = {"metric": "tiles/sec encoded (224x224, CTransPath = ConvStem + Swin-T)",
                                               "value": round(st_tiles.shape[0] * ctx.world / dt_sw, 1), "unit": "tiles/s",
                                               "gflop_per_tile": round(scfg.matmul_flops_per_tile() / 1e9, 3),
                                               "finite": bool(torch.isfinite(sf.float()).all())}
            del sw
    except Exception as e:      # the headline metric must still be printed
        line.setdefault("secondary", {})["error"] = repr(e)[:200]
    if ctx.is_main and ctx.world == 1 and not a.no_cpu_baseline:
        line["cpu_baseline"] = cpu_baseline(cfg, sd, a.cpu_seconds, swin=is_swin)
    elif ctx.is_main:
        line["cpu_baseline"] = None
    if ctx.is_main:
        print(json.dumps(line), flush=True)
    if ctx.world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
