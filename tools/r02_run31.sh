set -x
R=$GRAFT_REPO_ROOT; mkdir -p $R/gpurun_out
cd $R
( timeout 1500 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_swin.py tests/test_gpu_mil.py tests/test_gpu_train.py -q -m gpu 2>&1 | tail -6 ) > gpurun_out/r02_run31_pytest.log 2>&1
tail -3 gpurun_out/r02_run31_pytest.log
for i in 1 2; do timeout 300 python bench.py --model ctranspath --no-cpu-baseline --no-secondary --e2e-tiles 0 2>&1 | grep -o '"value": [0-9.]*' | head -1; done
