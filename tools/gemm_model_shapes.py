"""The four GEMMs of one ViT-L block at one 510-tile chunk, library-default kernels, a few launches each: a target for
rocprofv3 --pmc passes (tools/prof_cmd.sh).  python tools/gemm_model_shapes.py"""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from stamp_amd import _lib, ops  # noqa: E402

M = 131070
for name, N, K, epi in (("qkv", 3072, 1024, _lib.EPI_BIAS), ("proj", 1024, 1024, _lib.EPI_RESIDUAL), ("fc1", 4096, 1024, _lib.EPI_BIAS_GELU),
                        ("fc2", 1024, 4096, _lib.EPI_RESIDUAL)):
    a = torch.randn(M, K, device="cuda").half()
    w = (torch.randn(N, K, device="cuda") / K ** 0.5).half()
    b = torch.zeros(N, device="cuda")
    out = torch.zeros(M, N, device="cuda") if epi == _lib.EPI_RESIDUAL else None
    for _ in range(4):
        ops.gemm(a, w, epi, bias=b, out=out)
    torch.cuda.synchronize()
    del a, w, out
