# round-2 GPU call 36: LN-fold consumer with the row terms in the initial accumulators -- parity tests, then old/new library A/B
set -x
R=$GRAFT_REPO_ROOT; mkdir -p $R/gpurun_out
cd $R
( timeout 1200 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_vit.py -q -m gpu -x 2>&1 | tail -5 )
for i in 1 2 3; do
  for L in build/ab/libamdstamp_old.so stamp_amd/lib/libamdstamp.so; do
    echo "lib=$L"
    AMDSTAMP_LIB=$R/$L timeout 300 python bench.py --no-secondary --no-cpu-baseline --e2e-tiles 0 2>&1 | grep -o '"value": [0-9.]*, "unit": "tiles/s", "n_gpus"'
  done
done
