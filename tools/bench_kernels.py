"""Micro-benchmarks of the individual HIP kernels at the ViT-L/14 shapes (GPU box only)."""
import argparse
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from stamp_amd import _lib, ops  # noqa: E402


def timeit(fn, iters=10, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e-3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=128)
    ap.add_argument("--dtype", default="f16")
    ap.add_argument("--cfgs", default="0,3,7,8")
    ap.add_argument("--rounds", type=int, default=1)
    a = ap.parse_args()
    dt = torch.float16 if a.dtype == "f16" else torch.bfloat16
    dev = torch.device("cuda:0")
    B, T, D, Hd = a.batch, 257, 1024, 4096
    M = B * T
    g = torch.Generator().manual_seed(0)
    x = torch.randn(M, D, generator=g).to(dev)
    h = torch.randn(M, D, generator=g).to(dev, dt)
    u = torch.randn(M, Hd, generator=g).to(dev, dt)
    shapes = {"qkv": (3 * D, D, _lib.EPI_BIAS, h), "proj": (D, D, _lib.EPI_RESIDUAL, h),
              "fc1": (Hd, D, _lib.EPI_BIAS_GELU, h), "fc2": (D, Hd, _lib.EPI_RESIDUAL, u)}
    for name, (N, K, epi, A) in shapes.items():
        w = (torch.randn(N, K, generator=g) / K ** 0.5).to(dev, dt)
        bias = torch.randn(N, generator=g).to(dev)
        out = x.clone() if epi == _lib.EPI_RESIDUAL else torch.empty(M, N, dtype=dt, device=dev)
        for cfg in [int(c) for c in a.cfgs.split(",")] * a.rounds:
            t = timeit(lambda: ops.gemm(A, w, epi, bias=bias, out=out, cfg=cfg))
            print(f"gemm {name:5s} M={M} N={N} K={K} cfg={cfg}: {t*1e6:9.1f} us  {2*M*N*K/t/1e12:8.1f} TF/s", flush=True)
    gam, bet = torch.ones(D, device=dev), torch.zeros(D, device=dev)
    t = timeit(lambda: ops.layernorm(x, gam, bet, 1e-6, dt))
    print(f"layernorm rows={M} cols={D}: {t*1e6:9.1f} us  {M*D*6/t/1e9:8.1f} GB/s")
    qkv = torch.randn(M, 3 * D, generator=g).to(dev, dt)
    t = timeit(lambda: ops.attention_vit(qkv, B, T, 16))
    print(f"attention B={B} T={T} H=16: {t*1e6:9.1f} us  {4*B*16*T*T*64/t/1e12:8.1f} TF/s  ({M*4*D*2/t/1e9:.0f} GB/s)")
    tiles = torch.randint(0, 256, (B, 224, 224, 3), dtype=torch.uint8, generator=g).to(dev)
    t = timeit(lambda: ops.tile_im2col_u8(tiles, 14, 640, dt))
    print(f"im2col B={B}: {t*1e6:9.1f} us  {(B*150528+B*256*640*2)/t/1e9:8.1f} GB/s")


if __name__ == "__main__":
    main()
