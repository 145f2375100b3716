cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_mil.py tests/test_gpu_mil_seam.py tests/test_gpu_train.py -x -q -s  2>&1 | grep -v "^$" | grep "largest\|passed\|failed\|Error" | cut -c1-1500 > gpurun_out/mil_tests.txt
cat gpurun_out/mil_tests.txt
