import sys, time, torch
sys.path.insert(0, "/root/repo")
from stamp_amd.mil import TransMIL
import os
torch.set_float32_matmul_precision(os.environ.get('AMDS_MATMUL', 'high'))      # the reference's training setting (train.py:519), which the library follows; AMDS_MATMUL=highest: exact fp32
tm = TransMIL(dim_output=2, dim_input=1024, dim_hidden=512).eval().cuda()
for B in (8, 32, 64):
    bags = torch.randn(B, 1024, 1024, device="cuda")
    with torch.no_grad():
        tm(bags); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(3): y = tm(bags)
        torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 3
    print(B, f"{B/dt:.0f} bags/s  {dt*1e3:.1f} ms")
