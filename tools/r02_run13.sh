set -x
R=$GRAFT_REPO_ROOT; mkdir -p $R/gpurun_out
cd $R
( timeout 900 python -m pytest tests/test_gpu_kernels.py -q -m gpu -x -k "lnfold" 2>&1 | tail -30 ) > gpurun_out/r02_run13_pytest.log 2>&1
tail -30 gpurun_out/r02_run13_pytest.log
