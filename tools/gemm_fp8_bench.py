"""The opt-in fp8 GEMM at the tile encoder's four shapes (one 1020-tile chunk of ViT-L/14), next to the production fp16 kernel on the same box:
python tools/gemm_fp8_bench.py [M]"""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from stamp_amd import _lib, ops  # noqa: E402

M = int(sys.argv[1]) if len(sys.argv) > 1 else 262140
dev = "cuda"
g = torch.Generator(device=dev).manual_seed(0)


def timeit(fn, n=20, warm=3):
    for _ in range(warm):
        fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3      # us


print(f"M = {M}; fp8 peak 5000 TFLOP/s dense, fp16 2500")
print(f"{'shape':26s} {'fp8 us':>9s} {'TF/s':>8s} {'of 5 PF':>8s} {'fp16 us':>9s} {'TF/s':>8s} {'speed-up':>9s} {'quantise A us':>14s}")
tot8 = tot16 = totq = 0.0
for name, N, K, epi in (("qkv   N=3072 K=1024", 3072, 1024, "bias"), ("proj  N=1024 K=1024", 1024, 1024, "res"), ("fc1   N=4096 K=1024", 4096, 1024, "gelu"),
                        ("fc2   N=1024 K=4096", 1024, 4096, "res")):
    a = torch.randn(M, K, device=dev, generator=g).half()
    w = (torch.randn(N, K, device=dev, generator=g) / K ** 0.5).half()
    bias = torch.randn(N, device=dev, generator=g)
    a8, sa = ops.quantize_rows_e4m3(a)
    w8, sw = ops.quantize_rows_e4m3(w)
    x = torch.zeros(M, N, device=dev) if epi == "res" else None
    e8 = {"bias": _lib.EPI_BIAS, "gelu": _lib.EPI_BIAS_GELU, "res": _lib.EPI_RESIDUAL}[epi]
    out8 = None if epi == "res" else torch.empty(M, N, dtype=torch.float16, device=dev)
    t8 = timeit(lambda: ops.gemm_fp8(a8, w8, e8, rowscale=sa, colscale=sw, bias=bias, out=x if epi == "res" else out8))
    out16 = None if epi == "res" else torch.empty(M, N, dtype=torch.float16, device=dev)
    t16 = timeit(lambda: ops.gemm(a, w, e8, bias=bias, out=x if epi == "res" else out16))
    tq = timeit(lambda: ops.quantize_rows_e4m3(a))
    fl = 2.0 * M * N * K
    tot8, tot16, totq = tot8 + t8, tot16 + t16, totq + tq
    print(f"{name:26s} {t8:9.1f} {fl / t8 / 1e6:8.0f} {fl / t8 / 1e6 / 5000:8.3f} {t16:9.1f} {fl / t16 / 1e6:8.0f} {t16 / t8:9.2f} {tq:14.1f}")
    del a, w, a8, w8, x, out8, out16
print(f"{'block (4 GEMMs)':26s} {tot8:9.1f} {'':8s} {'':8s} {tot16:9.1f} {'':8s} {tot16 / tot8:9.2f} {totq:14.1f}")
print(f"with the activations quantised by separate launches: {tot16 / (tot8 + totq):.2f}x the fp16 GEMMs")
