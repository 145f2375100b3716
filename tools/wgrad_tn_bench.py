"""amds_wgrad_tn (token-major operands, kernel id 15) against the form it replaces (two transposes + split-K batched GEMM), per shape of the MIL `vit`
training step:  python tools/wgrad_tn_bench.py [tokens]"""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from stamp_amd import _lib, ops, train_ops as T  # noqa: E402

tokens = int(sys.argv[1]) if len(sys.argv) > 1 else 65600
split_k = 32
lib = _lib.lib()


def timed(fn, n=30):
    for _ in range(5):
        fn()
    a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


for N, K in ((512, 512), (1536, 512), (512, 1024), (256, 256)):
    dy = (torch.randn(tokens, N, device="cuda") * 0.5).bfloat16()
    x = (torch.randn(tokens, K, device="cuda") * 0.5).bfloat16()
    part = torch.empty(split_k, N, K, dtype=torch.float32, device="cuda")
    unit = 64 * split_k
    Mp = (tokens + unit - 1) // unit * unit
    chunk = Mp // split_k
    dyT, xT = T.transpose16(dy, Mp), T.transpose16(x, Mp)
    st = torch.cuda.current_stream().cuda_stream
    t_tn = timed(lambda: _lib.check(lib.amds_wgrad_tn(dy.data_ptr(), N, x.data_ptr(), K, tokens, N, K, split_k, _lib.BF16, part.data_ptr(), st), "tn"))
    t_g = timed(lambda: _lib.check(lib.amds_gemm_batched(dyT.data_ptr(), Mp, chunk, xT.data_ptr(), Mp, chunk, N, K, chunk, split_k, _lib.BF16, _lib.EPI_BIAS_F32,
                                                         part.data_ptr(), K, N * K, None, 1.0, st), "g"))
    t_tr = timed(lambda: (T.transpose16(dy, Mp, out=dyT), T.transpose16(x, Mp, out=xT)))
    fl = 2.0 * tokens * N * K
    print(f"N={N} K={K}: wgrad_tn {t_tn:7.1f} us ({fl / t_tn / 1e6:5.0f} TF/s) | batched GEMM on transposed operands {t_g:7.1f} us ({fl / t_g / 1e6:5.0f} TF/s) + transposes {t_tr:6.1f} us", flush=True)
