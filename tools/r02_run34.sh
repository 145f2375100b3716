set -x
R=$GRAFT_REPO_ROOT; mkdir -p $R/gpurun_out
cd $R
( timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_vit.py -q -m gpu -k "attention or hd80 or virchow or tiny" 2>&1 | tail -4 ) > gpurun_out/r02_run34_pytest.log 2>&1
tail -3 gpurun_out/r02_run34_pytest.log
for n in 512 256 512 256; do echo "threads=$n"; AMDS_ATTN80_THREADS=$n timeout 120 python tools/attn_only.py 1020 261 16 80 2>&1 | tail -1; done
