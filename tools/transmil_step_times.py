"""Per-step wall times of the TransMIL training step (fwd / bwd / AdamW separately synchronised), with the caching allocator's counters beside
them -- to find what makes some 4-step blocks of bench.py's leg 2.3x slower than others:  python tools/transmil_step_times.py [steps] [churn]
churn = 1: allocate and free a few large tensors of other sizes first (what the bench's earlier legs leave in the allocator's cache)."""
import sys
import time
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from stamp_amd.mil import TransMIL  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 24
churn = int(sys.argv[2]) if len(sys.argv) > 2 else 0
B = 64
torch.manual_seed(0)
if churn:
    junk = [torch.empty(int(s * 2 ** 20), dtype=torch.uint8, device="cuda") for s in (5400, 2700, 1300, 900, 640, 300, 150, 77)]
    del junk
tm = TransMIL(dim_output=2, dim_input=1024, dim_hidden=512).cuda().train()
opt = torch.optim.AdamW(tm.parameters(), lr=1e-4)
bags = torch.randn(B, 1024, 1024, device="cuda")
tg = torch.nn.functional.one_hot(torch.arange(B, device="cuda") % 2, 2).float()


def sync():
    torch.cuda.synchronize()
    return time.perf_counter()


for i in range(steps):
    st = torch.cuda.memory_stats()
    a0, r0, n0 = st["allocation.all.allocated"], st["reserved_bytes.all.current"], st.get("num_device_alloc", 0)
    t0 = sync()
    opt.zero_grad(set_to_none=True)
    loss = torch.nn.functional.cross_entropy(tm(bags), tg)
    t1 = sync()
    loss.backward()
    t2 = sync()
    opt.step()
    t3 = sync()
    st = torch.cuda.memory_stats()
    print(f"step {i:2d}: fwd {1e3 * (t1 - t0):7.1f}  bwd {1e3 * (t2 - t1):7.1f}  opt {1e3 * (t3 - t2):6.1f} ms   allocs {st['allocation.all.allocated'] - a0:4d}  "
          f"device allocs {st.get('num_device_alloc', 0) - n0:3d}  reserved {st['reserved_bytes.all.current'] / 2 ** 30:6.2f} GiB ({(st['reserved_bytes.all.current'] - r0) / 2 ** 20:+.0f} MiB)", flush=True)
