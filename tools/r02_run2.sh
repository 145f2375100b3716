# round-2 GPU call 2: MIL seam tests
set -x
R=$GRAFT_REPO_ROOT; mkdir -p $R/gpurun_out
cd $R
timeout 2400 python -m pytest tests/test_gpu_mil_seam.py tests/test_gpu_mil.py tests/test_gpu_train.py -q -m gpu -s 2>&1 | tail -400 > gpurun_out/r02_pytest_mil.log
tail -60 gpurun_out/r02_pytest_mil.log
