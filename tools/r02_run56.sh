# round-2 GPU call 56: head_dim-80 attention with the ninth query block split over the waves by key tile -- parity, A/B, Virchow2 bench
set -x
R=$GRAFT_REPO_ROOT; cd $R
( timeout 900 python -m pytest tests/test_gpu_kernels.py -q -m gpu -x -k "head_dim_80" 2>&1 | tail -3 )
for i in 1 2; do
  for L in build/ab/libamdstamp_prev.so stamp_amd/lib/libamdstamp.so; do
    echo "lib=$L"
    AMDSTAMP_LIB=$R/$L timeout 120 python tools/attn_only.py 1020 261 16 80 2>&1 | tail -1
  done
done
( timeout 900 python -m pytest tests/test_gpu_vit.py -q -m gpu -x 2>&1 | tail -3 )
for L in build/ab/libamdstamp_prev.so stamp_amd/lib/libamdstamp.so build/ab/libamdstamp_prev.so stamp_amd/lib/libamdstamp.so; do
  echo "lib=$L"
  AMDSTAMP_LIB=$R/$L timeout 400 python bench.py --model virchow2 --steps 4 --warmup 1 --no-cpu-baseline --no-secondary --e2e-tiles 0 2>&1 | grep -o '"value": [0-9.]*, "unit": "tiles/s", "n_gpus"'
done
