# round-2 GPU call 50: fc1 -> fc2 over row ranges (Infinity Cache) -- parity + A/B by AMDS_VIT_MLP_ROWS
set -x
R=$GRAFT_REPO_ROOT; mkdir -p $R/gpurun_out
cd $R
( timeout 900 python -m pytest tests/test_gpu_vit.py -q -m gpu -x 2>&1 | tail -3 )
for i in 1 2; do
  for r in 0 16384 32768 8192 16384 0; do
    echo "AMDS_VIT_MLP_ROWS=$r"
    AMDS_VIT_MLP_ROWS=$r timeout 300 python bench.py --no-secondary --no-cpu-baseline --e2e-tiles 0 2>&1 | grep -o '"value": [0-9.]*, "unit": "tiles/s", "n_gpus"'
  done
done
