set -x
R=$GRAFT_REPO_ROOT; mkdir -p $R/gpurun_out
cd $R
( timeout 1500 python -m pytest tests/test_gpu_vit.py tests/test_gpu_kernels.py tests/test_gpu_seams.py -q -m gpu 2>&1 | tail -30 ) > gpurun_out/r02_run14_pytest.log 2>&1
tail -12 gpurun_out/r02_run14_pytest.log
