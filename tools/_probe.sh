cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 600 python tools/parity_probe.py uni2_h 2 5 9 1 4 24 > gpurun_out/probe_uni2_split.txt 2>&1
AMDS_VIT_PATCH_SPLIT=0 timeout 600 python tools/parity_probe.py uni2_h 2 5 9 1 4 24 > gpurun_out/probe_uni2_nosplit.txt 2>&1
timeout 600 python -m pytest tests/test_gpu_train.py tests/test_gpu_vit.py tests/test_gpu_mil_seam.py -x -q 2>&1 | tail -5
cat gpurun_out/probe_uni2_split.txt gpurun_out/probe_uni2_nosplit.txt
