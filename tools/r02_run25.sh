set -x
R=$GRAFT_REPO_ROOT; mkdir -p $R/gpurun_out
cd $R
( timeout 900 python -m pytest tests/test_gpu_transmil_train.py tests/test_gpu_mil.py tests/test_gpu_seams.py -q -m gpu 2>&1 | tail -6 ) > gpurun_out/r02_run25_pytest.log 2>&1
tail -4 gpurun_out/r02_run25_pytest.log
timeout 600 python tools/bgemm_f32_yardstick.py 2>&1 | tail -6
timeout 300 python tools/transmil_train_only.py 64 3 2>&1 | tail -1
timeout 300 python tools/transmil_only.py 2>&1 | tail -3
