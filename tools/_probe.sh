cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_vit.py -x -q -s 2>&1 | grep -v "^$" | tail -40 > gpurun_out/vit_tests.txt
cat gpurun_out/vit_tests.txt
