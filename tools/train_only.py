"""Run only the HIP MIL-vit training step (for rocprofv3):  python tools/train_only.py [steps] [dropout] [alibi 0/1] [precision high/medium] [cls_tail 0/1]"""
import sys
import time
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch  # noqa: E402

from stamp_amd.mil import VisionTransformer as HipMil  # noqa: E402
from stamp_amd.mil_train import HipMilVitTrainer  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 4
torch.manual_seed(1)
mil = HipMil(dim_output=2, dim_input=1024, dim_model=512, n_layers=2, n_heads=8, dim_feedforward=512, dropout=float(sys.argv[2]) if len(sys.argv) > 2 else 0.25, use_alibi=len(sys.argv) > 3 and sys.argv[3] == "1").eval()
bags = torch.randn(64, 1024, 1024, generator=torch.Generator().manual_seed(1)).half().cuda()
prec = sys.argv[4] if len(sys.argv) > 4 else "high"
if len(sys.argv) > 5:
    from stamp_amd import ops
    ops.set_mil_cls_tail(sys.argv[5] == "1")
trn = HipMilVitTrainer(mil, device="cuda", total_steps=100, sched_interval="step", precision=prec)
crd = (torch.rand(64, 1024, 2, generator=torch.Generator().manual_seed(2)) * 4e4).cuda() if mil.use_alibi else None
tg = torch.nn.functional.one_hot(torch.arange(64) % 2, 2).float()
cw = torch.tensor([0.5, 0.5])
for _ in range(2):
    trn.step(bags, tg, cw, coords=crd)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(steps):
    loss, _ = trn.step(bags, tg, cw, coords=crd)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / steps
print(f"train step (alibi={mil.use_alibi}, {prec}): {dt*1e3:.2f} ms, {64/dt:.0f} bags/s, loss {float(loss):.4f}")
