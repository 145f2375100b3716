import sys, torch
import os; sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from stamp_amd import ops
from tools.gap_only import weights
dev = torch.device('cuda:0')
w = weights(768, 512, 256, dev)
g = torch.Generator().manual_seed(1)
for N in (1024, 4096, 5000, 8192, 16384):
    x = torch.randn(N, 768, generator=g).to(dev)
    for mode in ("slab", "split"):
        if mode == "split" and N > 12288: continue
        for _ in range(4):
            ops.gated_attn_pool(x, w, mode=mode)
for B in (8, 16, 32, 64):
    xb = torch.randn(B, 1024, 768, generator=g).to(dev)
    for _ in range(4):
        ops.gated_attn_pool_batched(xb, [1024] * B, w, mode="slab")
torch.cuda.synchronize()
