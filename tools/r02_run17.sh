set -x
R=$GRAFT_REPO_ROOT; mkdir -p $R/gpurun_out
cd $R
( timeout 1500 python -m pytest tests/test_gpu_vit.py tests/test_gpu_seams.py -q -m gpu -x 2>&1 | tail -12 ) > gpurun_out/r02_run17_pytest.log 2>&1
tail -5 gpurun_out/r02_run17_pytest.log
for t in 1 0 1 0; do
AMDS_VIT_TAIL=$t timeout 400 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --e2e-tiles 0 2>&1 | grep -o '"drop_in_b64": {[^}]*}' | grep -o '"value": [0-9.]*, "unit": "tiles/s", "batches": [0-9]*, "hbm_resident_b64": [0-9.]*'
done > gpurun_out/r02_run17_b64.log 2>&1
cat gpurun_out/r02_run17_b64.log
