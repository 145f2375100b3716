set -x
R=$GRAFT_REPO_ROOT; mkdir -p $R/gpurun_out
cd $R
( timeout 1500 python -m pytest tests/test_gpu_vit.py tests/test_gpu_seams.py -q -m gpu -x 2>&1 | tail -6 ) > gpurun_out/r02_run29_pytest.log 2>&1
tail -3 gpurun_out/r02_run29_pytest.log
for t in 1 0 1; do
echo "tail=$t"; AMDS_VIT_TAIL=$t timeout 300 python tools/b64_only.py 24 2>&1 | tail -1
done > gpurun_out/r02_run29.log 2>&1
cat gpurun_out/r02_run29.log
timeout 300 python bench.py --no-secondary --no-cpu-baseline --e2e-tiles 0 2>&1 | grep -o '"value": [0-9.]*' | head -1
