set -x
R=$GRAFT_REPO_ROOT; mkdir -p $R/gpurun_out
cd $R
( timeout 1500 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_vit.py -q -m gpu -x 2>&1 | tail -6 ) > gpurun_out/r02_run24_pytest.log 2>&1
tail -3 gpurun_out/r02_run24_pytest.log
for f in 1 2; do
timeout 300 python bench.py --no-secondary --no-cpu-baseline --e2e-tiles 0 2>&1 | grep -o '"value": [0-9.]*' | head -1
done > gpurun_out/r02_run24_bench.log 2>&1
cat gpurun_out/r02_run24_bench.log
