"""Run only the HIP MIL-vit training step (for rocprofv3):  python tools/train_only.py [steps]"""
import sys
import time
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch  # noqa: E402

from stamp_amd.mil import VisionTransformer as HipMil  # noqa: E402
from stamp_amd.mil_train import HipMilVitTrainer  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 4
torch.manual_seed(1)
mil = HipMil(dim_output=2, dim_input=1024, dim_model=512, n_layers=2, n_heads=8, dim_feedforward=512, dropout=float(sys.argv[2]) if len(sys.argv) > 2 else 0.25, use_alibi=False).eval()
bags = torch.randn(64, 1024, 1024, generator=torch.Generator().manual_seed(1)).half().cuda()
trn = HipMilVitTrainer(mil, device="cuda", total_steps=100, sched_interval="step")
tg = torch.nn.functional.one_hot(torch.arange(64) % 2, 2).float()
cw = torch.tensor([0.5, 0.5])
for _ in range(2):
    trn.step(bags, tg, cw)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(steps):
    loss, _ = trn.step(bags, tg, cw)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / steps
print(f"train step: {dt*1e3:.2f} ms, {64/dt:.0f} bags/s, loss {float(loss):.4f}")
