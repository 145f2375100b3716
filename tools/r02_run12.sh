set -x
R=$GRAFT_REPO_ROOT; mkdir -p $R/gpurun_out
cd $R
( timeout 900 python -m pytest tests/test_gpu_kernels.py -q -m gpu -x 2>&1 | tail -8 ) > gpurun_out/r02_run12_pytest.log 2>&1
tail -4 gpurun_out/r02_run12_pytest.log
timeout 600 python tools/gemm_sched_ab.py > gpurun_out/r02_run12_ab.log 2>&1
cat gpurun_out/r02_run12_ab.log
