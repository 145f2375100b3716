"""fp32 batched GEMM: amds_bgemm_f32 against the vendor library (torch.bmm -> hipBLASLt / rocBLAS), yardstick only.  python tools/bgemm_f32_yardstick.py"""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from stamp_amd import transmil_core as tc  # noqa: E402

torch.backends.cuda.matmul.allow_tf32 = False
for Z, M, N, K in ((512, 256, 256, 256), (512, 1280, 256, 64), (512, 256, 64, 1280), (512, 1280, 64, 256), (64, 1536, 512, 1280), (1, 65600, 1536, 512)):
    A = torch.randn(Z, M, K, device="cuda")
    B = torch.randn(Z, K, N, device="cuda")
    out = torch.empty(Z, M, N, device="cuda")

    def timeit(fn, n=20):
        for _ in range(3):
            fn()
        t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0.record()
        for _ in range(n):
            fn()
        t1.record(); torch.cuda.synchronize()
        return t0.elapsed_time(t1) / n * 1e3

    us_a = timeit(lambda: tc._mm(A, B, False, out=out))
    us_v = timeit(lambda: torch.bmm(A, B, out=out))
    fl = 2.0 * Z * M * N * K
    err = ((tc._mm(A, B, False).double() - A.double() @ B.double()).norm() / (A.double() @ B.double()).norm()).item()
    errv = ((torch.bmm(A, B).double() - A.double() @ B.double()).norm() / (A.double() @ B.double()).norm()).item()
    # the other operand layouts of the same product: B stored [N][K] (transb), A stored [K][M] (transa), both
    At, Bt = A.transpose(1, 2).contiguous(), B.transpose(1, 2).contiguous()
    lay = []
    for ta, tb in ((False, True), (True, False), (True, True)):
        us = timeit(lambda: tc._mm(At if ta else A, Bt if tb else B, tb, out=out, transa=ta))
        lay.append((ta, tb, us))
    print("    layouts (TF/s): " + "  ".join(f"{'At' if ta else 'A'}.{'Bt' if tb else 'B'} {fl / us / 1e6:5.1f}" for ta, tb, us in lay), flush=True)
    print(f"Z={Z} M={M} N={N} K={K}: amds {us_a:8.1f} us {fl / us_a / 1e6:6.1f} TF/s (err {err:.1e}) | vendor {us_v:8.1f} us {fl / us_v / 1e6:6.1f} TF/s (err {errv:.1e})", flush=True)
