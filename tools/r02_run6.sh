set -x
R=$GRAFT_REPO_ROOT; mkdir -p $R/gpurun_out
cd $R
timeout 1200 python -m pytest tests/test_gpu_tiling.py tests/test_gpu_kernels.py tests/test_gpu_vit.py -q -m gpu -k "tiling or attention or vit" 2>&1 | tail -15 > gpurun_out/r02_pytest_attn.log
cat gpurun_out/r02_pytest_attn.log
timeout 300 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-secondary > gpurun_out/r02_bench_attn257.log 2>&1
grep -o '"value": [0-9.]*, "unit": "tiles/s", "n_gpus"' gpurun_out/r02_bench_attn257.log; grep -o '"time_share": {[^}]*}' gpurun_out/r02_bench_attn257.log
AMDS_ATTN_257=0 timeout 300 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-secondary > gpurun_out/r02_bench_attn_old.log 2>&1
grep -o '"value": [0-9.]*, "unit": "tiles/s", "n_gpus"' gpurun_out/r02_bench_attn_old.log; grep -o '"time_share": {[^}]*}' gpurun_out/r02_bench_attn_old.log
