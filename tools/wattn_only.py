"""Time the window-attention kernel alone (stage shapes of CTransPath):  python tools/wattn_only.py [B]"""
import sys
import time
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch  # noqa: E402

from stamp_amd import ops  # noqa: E402
from stamp_amd.swin import rel_bias_lane_table, shift_mask_bits  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
mask = shift_mask_bits().cuda()
for grid, heads in ((56, 3), (28, 6), (14, 12), (7, 24)):
    qkv = torch.randn(B * grid * grid, 3 * heads * 32, device="cuda").half()
    bias = rel_bias_lane_table(torch.randn(169, heads)).cuda()
    for shift in (0, 3 if grid > 7 else 0):
        for _ in range(3):
            ops.window_attention(qkv, bias, mask, B, grid, heads, shift)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10):
            ops.window_attention(qkv, bias, mask, B, grid, heads, shift)
        torch.cuda.synchronize()
        us = (time.perf_counter() - t0) / 10 * 1e6
        mb = qkv.numel() * 2 * 4 / 3 / 1e6
        print(f"grid {grid:2d} heads {heads:2d} shift {shift}: {us:8.1f} us  {mb / us * 1e-6 * 1e6 / 1e6:.2f} TB/s ({mb:.0f} MB)")
