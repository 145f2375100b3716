cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_fp8.py -x -q -s 2>&1 | tail -25
