cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_seams.py tests/test_gpu_vit.py -x -q -k "titan or diagnostics" 2>&1 | tail -25
