cd $GRAFT_REPO_ROOT
timeout 240 python -m pytest tests/test_gpu_tiling.py -x -q 2>&1 | tail -2
