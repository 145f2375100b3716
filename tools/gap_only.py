#!/usr/bin/env python
"""Gated-attention pooling (CHIEF, chief.py:74-89) on the GPU alone: the fused single launch (csrc/gap_fused.hip) against the six-launch form
(csrc/gap.hip), one bag at a time and many bags per launch.  Prints one JSON object; `--trace` runs a short loop for rocprofv3 --kernel-trace.

    python tools/gap_only.py [--rows 1024] [--feat 768] [--reps 200]
"""
from __future__ import annotations

import argparse
import json
import sys
import time
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))

import torch  # noqa: E402

F32_MFMA_PEAK_TF = 157.3      # /opt/skills/guides/MI355X_MICROARCH.md: v_mfma_f32_16x16x4_f32, exact fp32
HBM_ACHIEVABLE_TBS = 6.29     # same guide: float4 copy


def weights(F, L, D, dev, seed=0):
    g = torch.Generator().manual_seed(seed)
    w = {"fc_w": torch.randn(L, F, generator=g) / F ** 0.5, "fc_b": torch.randn(L, generator=g) * 0.1,
         "a_w": torch.randn(D, L, generator=g) / L ** 0.5, "a_b": torch.randn(D, generator=g) * 0.1,
         "b_w": torch.randn(D, L, generator=g) / L ** 0.5, "b_b": torch.randn(D, generator=g) * 0.1,
         "c_w": torch.randn(1, D, generator=g) / D ** 0.5, "c_b": torch.randn(1, generator=g) * 0.1}
    return {k: v.to(dev).contiguous() for k, v in w.items()}


def timeit(fn, reps, warm=5):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) / reps
    return a.elapsed_time(b) * 1e-3 / reps, wall


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=1024)
    ap.add_argument("--feat", type=int, default=768)
    ap.add_argument("--hidden", type=int, default=512)
    ap.add_argument("--gate", type=int, default=256)
    ap.add_argument("--reps", type=int, default=200)
    ap.add_argument("--trace", action="store_true", help="short loop only (for rocprofv3 --kernel-trace --stats)")
    a = ap.parse_args()
    from stamp_amd import ops

    dev = torch.device("cuda:0")
    N, F, L, D = a.rows, a.feat, a.hidden, a.gate
    w = weights(F, L, D, dev)
    flop = 2.0 * N * (F * L + 2 * L * D)
    byts = 4.0 * N * F
    g = torch.Generator().manual_seed(1)
    x = torch.randn(N, F, generator=g).to(dev)
    out = {"shape": {"rows": N, "F": F, "L": L, "D": D}, "gflop_per_bag": round(flop / 1e9, 4), "mb_per_bag": round(byts / 1e6, 3)}
    if a.trace:
        xb = torch.randn(256, N, F, generator=g).to(dev)
        for _ in range(5):
            ops.gated_attn_pool(x, w, fused=False)
            ops.gated_attn_pool(x, w)
            ops.gated_attn_pool_batched(xb, [N] * 256, w)
        torch.cuda.synchronize()
        print(json.dumps({"trace": "5 x (six-launch bag, fused bag, fused batch of 256)"}))
        return
    # parity of the two forms on the same bag
    o_u, a_u = ops.gated_attn_pool(x, w, return_attn=True, fused=False)
    o_f, a_f = ops.gated_attn_pool(x, w, return_attn=True)
    out["fused_vs_six_launch"] = {"rel_l2_out": float(((o_f - o_u).norm() / o_u.norm()).item()), "max_abs_attn_raw": float((a_f - a_u).abs().max().item())}
    # one bag per call
    for name, fused, mode, launches in (("six_launch_one_bag", False, "auto", "6 kernels + 4 copies"), ("fused_one_bag", True, "auto", "1 kernel + 1 memset"),
                                        ("fused_one_bag_slab_form", True, "slab", "1 kernel + 1 memset"), ("fused_one_bag_split_form", True, "split", "1 kernel + 1 memset")):
        if mode == "split" and N > 4096:
            continue
        dt, wall = timeit(lambda: ops.gated_attn_pool(x, w, fused=fused, mode=mode), a.reps)
        out[name] = {"us_per_bag_gpu": round(dt * 1e6, 2), "us_per_bag_wall": round(wall * 1e6, 2), "bags_per_s": round(1 / dt, 1), "launches_per_bag": launches,
                     "tflops": round(flop / dt / 1e12, 2), "frac_f32_mfma_peak": round(flop / dt / 1e12 / F32_MFMA_PEAK_TF, 4),
                     "frac_hbm": round(byts / dt / 1e12 / HBM_ACHIEVABLE_TBS, 4)}
    # many bags per launch
    for B in (16, 64, 256, 1024):
        xb = torch.randn(B, N, F, generator=g).to(dev)
        lens = [N] * B
        offs = (torch.arange(B + 1, dtype=torch.int64) * N).to(dev)
        ob = ops.gated_attn_pool_batched(xb, lens, w, offsets=offs)
        o1 = ops.gated_attn_pool(xb[B - 1], w, mode="slab")      # the batch runs the slab form; `auto` would take the split form for one small bag (another sum order)
        same = bool(torch.equal(ob[B - 1], o1))
        dt, wall = timeit(lambda: ops.gated_attn_pool_batched(xb, lens, w, offsets=offs), max(3, a.reps // max(B // 8, 1)), warm=2)
        out[f"fused_batch_{B}"] = {"us_per_bag_gpu": round(dt / B * 1e6, 3), "bags_per_s": round(B / dt, 1), "launches_per_bag": f"1/{B} kernel + 1/{B} memset",
                                   "tflops": round(B * flop / dt / 1e12, 2), "frac_f32_mfma_peak": round(B * flop / dt / 1e12 / F32_MFMA_PEAK_TF, 4),
                                   "frac_hbm": round(B * byts / dt / 1e12 / HBM_ACHIEVABLE_TBS, 4), "last_bag_bit_equal_to_single_slab_form_call": same}
        if B == 64:   # the same 64 bags through the six-launch form, one call each
            dt6, _ = timeit(lambda: [ops.gated_attn_pool(xb[i], w, fused=False) for i in range(B)], max(2, a.reps // 32), warm=1)
            out["six_launch_64_calls"] = {"us_per_bag_gpu": round(dt6 / B * 1e6, 2), "bags_per_s": round(B / dt6, 1)}
        del xb
    # one slide-sized bag
    for Nb in (256, 2048, 4096, 5000, 8192, 12288, 16384, 20000, 50000):
        xl = torch.randn(Nb, F, generator=g).to(dev)
        fl = 2.0 * Nb * (F * L + 2 * L * D)
        for name, fused, mode in ((f"six_launch_bag_{Nb}", False, "auto"), (f"fused_bag_{Nb}", True, "auto"), (f"fused_bag_{Nb}_slab_form", True, "slab"),
                                  (f"fused_bag_{Nb}_split_form", True, "split")):
            if mode != "auto" and not 1024 < Nb <= 12288:
                continue
            dt, _ = timeit(lambda: ops.gated_attn_pool(xl, w, fused=fused, mode=mode), max(5, a.reps // 10))
            out[name] = {"us": round(dt * 1e6, 1), "tflops": round(fl / dt / 1e12, 2), "frac_f32_mfma_peak": round(fl / dt / 1e12 / F32_MFMA_PEAK_TF, 4)}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
