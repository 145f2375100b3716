set -x
R=$GRAFT_REPO_ROOT; mkdir -p $R/gpurun_out
cd $R
( timeout 900 python -m pytest tests/test_gpu_transmil_train.py tests/test_gpu_mil.py -q -m gpu 2>&1 | tail -8 ) > gpurun_out/r02_run23_pytest.log 2>&1
tail -5 gpurun_out/r02_run23_pytest.log
