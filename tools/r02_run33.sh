set -x
R=$GRAFT_REPO_ROOT; mkdir -p $R/gpurun_out
cd $R
timeout 300 python - <<'PY'
import torch
from stamp_amd import _lib, ops
g = torch.Generator().manual_seed(3)
for (M, N, K, epi) in ((5000, 4096, 1024, _lib.EPI_BIAS_GELU), (70000, 3072, 1024, _lib.EPI_BIAS), (33000, 1024, 4096, _lib.EPI_BIAS_F32)):
    a = torch.randn(M, K, generator=g).cuda().half(); w = (torch.randn(N, K, generator=g) / K ** 0.5).cuda().half(); b = torch.randn(N, generator=g).cuda()
    ref = ops.gemm(a, w, epi, bias=b, cfg=12)
    for cfg in (14, 15, 16):
        print(M, N, K, cfg, "bit-equal", torch.equal(ops.gemm(a, w, epi, bias=b, cfg=cfg), ref))
PY
timeout 900 python tools/gemm_sched_ab.py 262140 12,14,15,16 > gpurun_out/r02_run33_ab.log 2>&1
cat gpurun_out/r02_run33_ab.log
