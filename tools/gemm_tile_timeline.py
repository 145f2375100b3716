"""Where a 256x256 tile's time goes in gemm_4w64 (debug build bit 16 of amds_gemm_ablate: thread 0 of every workgroup stamps
s_memrealtime / s_memtime at: start, first K tile landed, K loop done, values staged in LDS, stores issued) and how long a CU
waits between two workgroups.  python tools/gemm_tile_timeline.py"""
import ctypes as C
import sys
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from stamp_amd import _lib  # noqa: E402

lib = _lib.lib()
f = lib.amds_gemm_ablate
f.restype = C.c_int
f.argtypes = [C.c_int, C.c_void_p, C.c_long, C.c_void_p, C.c_long, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_long, C.c_void_p, C.c_void_p]
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
for M, N, K in ((131070, 3072, 1024), (131070, 1024, 4096)):
    A = torch.randn(M, K, generator=g).to(dev, torch.float16)
    w = (torch.randn(N, K, generator=g) / K ** 0.5).to(dev, torch.float16)
    out = torch.empty(M, N, dtype=torch.float16, device=dev)
    nwg = ((M + 255) // 256) * (N // 256)
    log = torch.zeros(nwg * 8, dtype=torch.int64, device=dev)
    st = torch.cuda.current_stream().cuda_stream
    bias = torch.randn(N, generator=g).to(dev)
    lib.amds_gemm_debug_log.argtypes = [C.c_void_p]
    lib.amds_gemm_debug_log.restype = None
    lib.amds_gemm_debug_log(log.data_ptr())
    for _ in range(3):
        log.zero_()
        f(2016, A.data_ptr(), K, w.data_ptr(), K, M, N, K, out.data_ptr(), N, bias.data_ptr(), st)
    torch.cuda.synchronize()
    L = log.cpu().numpy().reshape(nwg, 8).astype(np.float64)
    rt, t1, t2, t3, t4, t5 = (L[:, i] for i in range(6))
    cu = (L[:, 7].astype(np.int64) & 15) * 256 + ((L[:, 6].astype(np.int64) >> 8) & 255)
    seg = np.stack([t2 - t1, t3 - t2, t4 - t3, t5 - t4], 1)
    # shader clocks per 10-ns realtime tick, from the first and last workgroup of each CU
    print(f"M={M} N={N} K={K}: {nwg} workgroups on {len(np.unique(cu))} CUs")
    names = ["first K tile (issue -> landed)", "K loop", "values -> LDS (+ barrier)", "read back + stores issued"]
    tot = (t5 - t1)
    for nm, col in zip(names, seg.T):
        print(f"   {nm:32s} mean {col.mean():8.0f} cycles  p10 {np.percentile(col, 10):8.0f}  p90 {np.percentile(col, 90):8.0f}   {100 * col.sum() / tot.sum():5.1f} % of in-kernel time")
    gaps, spans, clk = [], [], []
    for c in np.unique(cu):
        idx = np.where(cu == c)[0]
        idx = idx[np.argsort(rt[idx])]
        if len(idx) < 3:
            continue
        dur_rt = np.diff(rt[idx])                     # start-to-start in 10-ns ticks
        dur_cy = (t5 - t1)[idx][:-1]                  # in-kernel shader cycles of the earlier workgroup
        cy_per_tick = (t1[idx][-1] - t1[idx][0]) / max(rt[idx][-1] - rt[idx][0], 1)
        clk.append(cy_per_tick)
        gaps.extend(dur_rt * cy_per_tick - dur_cy)    # cycles between 'stores issued' of one and 'start' of the next
        spans.extend(dur_rt * cy_per_tick)
    gaps, spans = np.array(gaps), np.array(spans)
    print(f"   shader clock {np.mean(clk) * 100:.0f} MHz; start-to-start on a CU {spans.mean():.0f} cycles; hand-over gap (stores issued -> next workgroup's first stamp) "
          f"mean {gaps.mean():.0f}  p10 {np.percentile(gaps, 10):.0f}  p90 {np.percentile(gaps, 90):.0f} cycles = {100 * gaps.sum() / spans.sum():.1f} % of the CU's time")
