"""The MIL `vit` head's deploy forward alone (bags of 1024 x 1024-d fp16, batch 64), for rocprofv3:  python tools/mil_fwd_only.py [n]"""
import sys
import time
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch  # noqa: E402

from stamp_amd.mil import VisionTransformer as HipMil  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
torch.manual_seed(1)
mil = HipMil(dim_output=2, dim_input=1024, dim_model=512, n_layers=2, n_heads=8, dim_feedforward=512, dropout=0.25, use_alibi=False).eval()
bags = torch.randn(64, 1024, 1024, generator=torch.Generator().manual_seed(1)).half().cuda()
with torch.no_grad():
    for _ in range(3):
        mil(bags, coords=None, mask=None)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        y = mil(bags, coords=None, mask=None)
    torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / n
print(f"MIL vit forward: {dt * 1e3:.3f} ms, {64 / dt:.0f} bags/s")
