# round-2 GPU call 9: attention dephase sweep; residual epilogue with the early batch
set -x
R=$GRAFT_REPO_ROOT; mkdir -p $R/gpurun_out
cd $R
for d in 0 2 4 6 8 12 16 24; do
  AMDS_ATTN_DEPHASE=$d timeout 120 python tools/attn_only.py 1020 257 16 2>&1 | tail -1
done > gpurun_out/r02_run9_attn.log 2>&1
cat gpurun_out/r02_run9_attn.log
( timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_vit.py -q -m gpu -x 2>&1 | tail -5 ) > gpurun_out/r02_run9_pytest.log 2>&1
tail -3 gpurun_out/r02_run9_pytest.log
timeout 600 python bench.py --no-secondary --no-cpu-baseline --e2e-tiles 0 > gpurun_out/r02_run9_bench.log 2>&1
grep -o '"value": [0-9.]*' gpurun_out/r02_run9_bench.log | head -1
