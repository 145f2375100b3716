"""Time the tile-encoder attention kernel alone:  python tools/attn_only.py [B] [T] [H]"""
import sys
import time
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from stamp_amd import ops  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 510
T = int(sys.argv[2]) if len(sys.argv) > 2 else 257
H = int(sys.argv[3]) if len(sys.argv) > 3 else 16
HD = int(sys.argv[4]) if len(sys.argv) > 4 else 64
qkv = torch.randn(B * T, 3 * H * HD, generator=torch.Generator().manual_seed(0)).to("cuda", torch.float16)
for _ in range(4):
    ops.attention_vit(qkv, B, T, H, HD)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(20):
    ops.attention_vit(qkv, B, T, H, HD)
torch.cuda.synchronize()
us = (time.perf_counter() - t0) / 20 * 1e6
print(f"attention B={B} T={T} H={H} hd={HD}: {us:.1f} us, {4.0 * B * H * T * T * HD / us / 1e6:.0f} TFLOP/s-equivalent")
