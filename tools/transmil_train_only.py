"""One TransMIL training step (fwd + hand-derived bwd + AdamW) in a loop, for rocprofv3:  python tools/transmil_train_only.py [B] [steps]"""
import sys
import time
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from stamp_amd.mil import TransMIL
import os
torch.set_float32_matmul_precision(os.environ.get('AMDS_MATMUL', 'high'))      # the reference's training setting (train.py:519), which the library follows; AMDS_MATMUL=highest: exact fp32  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
torch.manual_seed(0)
tm = TransMIL(dim_output=2, dim_input=1024, dim_hidden=512).cuda().train()
opt = torch.optim.AdamW(tm.parameters(), lr=1e-4)
bags = torch.randn(B, 1024, 1024, device="cuda")
tg = torch.nn.functional.one_hot(torch.arange(B, device="cuda") % 2, 2).float()


def step():
    opt.zero_grad()
    loss = torch.nn.functional.cross_entropy(tm(bags), tg)
    loss.backward()
    opt.step()
    return loss.detach()        # (an attached loss pins the step's 15 GB arena: bench.py's note)


for _ in range(3):
    step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(steps):
    l = step()
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / steps
print(f"TransMIL train B={B}: {B / dt:.0f} bags/s  {dt * 1e3:.1f} ms/step  loss {l.item():.4f}")
