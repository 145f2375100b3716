"""One shape of amds_bgemm_f32 in a loop (for rocprofv3):  python tools/bgemm_f32_one.py Z M N K [n]"""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from stamp_amd import transmil_core as tc  # noqa: E402

Z, M, N, K = (int(v) for v in sys.argv[1:5])
n = int(sys.argv[5]) if len(sys.argv) > 5 else 10
A = torch.randn(Z, M, K, device="cuda")
B = torch.randn(Z, K, N, device="cuda")
out = torch.empty(Z, M, N, device="cuda")
for _ in range(n):
    tc._mm(A, B, False, out=out)
torch.cuda.synchronize()
