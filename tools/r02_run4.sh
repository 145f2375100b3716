set -x
R=$GRAFT_REPO_ROOT; mkdir -p $R/gpurun_out
cd $R
( time timeout 1500 python -m pytest tests/test_gpu_transmil_train.py -q -m gpu -s 2>&1 | tail -120 ) > gpurun_out/r02_pytest_transmil.log 2>&1
tail -50 gpurun_out/r02_pytest_transmil.log
