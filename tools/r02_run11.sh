set -x
R=$GRAFT_REPO_ROOT; mkdir -p $R/gpurun_out
cd $R
( timeout 900 python -m pytest tests/test_gpu_kernels.py -q -m gpu -x 2>&1 | tail -8 ) > gpurun_out/r02_run11_pytest.log 2>&1
tail -4 gpurun_out/r02_run11_pytest.log
timeout 600 python tools/gemm_sched_ab.py > gpurun_out/r02_run11_ab.log 2>&1
cat gpurun_out/r02_run11_ab.log
for c in 12 13 14; do
AMDS_GEMM_CFG=$c timeout 300 python bench.py --no-secondary --no-cpu-baseline --e2e-tiles 0 2>&1 | grep -o '"value": [0-9.]*' | head -1
done > gpurun_out/r02_run11_bench.log 2>&1
cat gpurun_out/r02_run11_bench.log
