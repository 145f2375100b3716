import sys
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from stamp_amd import ops
B, T, H = 255, 257, 16
qkv = torch.randn(B * T, 3 * H * 64, generator=torch.Generator().manual_seed(0)).to("cuda", torch.float16)
for _ in range(4):
    ops.attention_vit(qkv, B, T, H)
torch.cuda.synchronize()
