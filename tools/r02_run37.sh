# round-2 GPU call 37: T = 257 attention with the odd query on the MFMA pipe -- parity, then old/new library A/B
set -x
R=$GRAFT_REPO_ROOT; mkdir -p $R/gpurun_out
cd $R
( timeout 900 python -m pytest tests/test_gpu_kernels.py -q -m gpu -x -k "attention" 2>&1 | tail -5 )
for i in 1 2; do
  for L in build/ab/libamdstamp_old.so stamp_amd/lib/libamdstamp.so; do
    echo "lib=$L"
    AMDSTAMP_LIB=$R/$L timeout 120 python tools/attn_only.py 1020 257 16 2>&1 | tail -1
  done
done
( timeout 900 python -m pytest tests/test_gpu_vit.py -q -m gpu -x 2>&1 | tail -3 )
for L in build/ab/libamdstamp_old.so stamp_amd/lib/libamdstamp.so; do
  echo "lib=$L"
  AMDSTAMP_LIB=$R/$L timeout 300 python bench.py --no-secondary --no-cpu-baseline --e2e-tiles 0 2>&1 | grep -o '"value": [0-9.]*, "unit": "tiles/s", "n_gpus"'
done
