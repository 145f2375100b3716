"""A/B of the fused qkv + attention kernel against the two launches it replaces, alternating, at the headline shape:
   python tools/qkv_attn_ab.py [B=1020] [H=16] [reps=5]
prints per-variant microseconds (events on the stream around 10 back-to-back calls) for each repetition."""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from stamp_amd import _lib, ops  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1020
H = int(sys.argv[2]) if len(sys.argv) > 2 else 16
REPS = int(sys.argv[3]) if len(sys.argv) > 3 else 5
D = H * 64
g = torch.Generator().manual_seed(0)
x = (torch.randn(B * 257, D, generator=g) * 1.5).to("cuda", torch.float16)
w = (torch.randn(3 * D, D, generator=g) / D ** 0.5).to("cuda", torch.float16)
bias = (torch.randn(3 * D, generator=g) * 0.2).cuda()
xf = x.float()
rstd = (xf.var(1, unbiased=False) + 1e-6).rsqrt()
rowstat = torch.stack([rstd, -xf.mean(1) * rstd], 1).contiguous()
colsum = w.float().sum(1)
del xf
qkv = torch.empty(B * 257, 3 * D, dtype=torch.float16, device="cuda")
tail = torch.empty(B * 257, 3 * D, dtype=torch.float16, device="cuda")


def two():
    ops.gemm_lnfold(x, w, _lib.EPI_BIAS, out=qkv, bias=bias, rowstat=rowstat, colsum=colsum)
    return ops.attention_vit(qkv, B, 257, H)


def fused():
    xt, st = ops.gather_token_rows16(x, rowstat, B, 257, 256)
    ops.gemm_lnfold(xt, w, _lib.EPI_BIAS, out=tail[256::257], bias=bias, rowstat=st, colsum=colsum)
    return ops.qkv_attention_vit257(x, w, bias, B, H, rowstat=rowstat, colsum=colsum, qkv_tail=tail)


def fused_only():
    return ops.qkv_attention_vit257(x, w, bias, B, H, rowstat=rowstat, colsum=colsum, qkv_tail=tail)


def timed(fn, n=10):
    for _ in range(2):
        fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


same = torch.equal(two(), fused())
print(f"B={B} H={H} D={D}: fused == two launches bit for bit: {same}")
for r in range(REPS):
    print(f"rep {r}: gemm_lnfold + attention_vit {timed(two):8.1f} us | gather + tail gemm + fused {timed(fused):8.1f} us | fused kernel alone {timed(fused_only):8.1f} us")
