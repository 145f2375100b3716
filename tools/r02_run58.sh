# round-2 GPU call 58: persistent kernel for head_dim 64, T != 257 (UNI2-h 265, H-optimus 261) -- parity, A/B by AMDS_ATTN_PERSIST64
set -x
R=$GRAFT_REPO_ROOT; cd $R
( timeout 900 python -m pytest tests/test_gpu_kernels.py -q -m gpu -x -k "attention" 2>&1 | tail -3 )
for i in 1 2; do
  for e in 0 1; do
    echo "AMDS_ATTN_PERSIST64=$e"
    AMDS_ATTN_PERSIST64=$e timeout 120 python tools/attn_only.py 1020 265 24 2>&1 | tail -1
    AMDS_ATTN_PERSIST64=$e timeout 120 python tools/attn_only.py 1020 261 24 2>&1 | tail -1
  done
done
( timeout 900 python -m pytest tests/test_gpu_vit.py -q -m gpu -x 2>&1 | tail -3 )
for e in 0 1 0 1; do
  echo "AMDS_ATTN_PERSIST64=$e"
  AMDS_ATTN_PERSIST64=$e timeout 400 python bench.py --model uni2_h --steps 4 --warmup 1 --no-cpu-baseline --no-secondary --e2e-tiles 0 2>&1 | grep -o '"value": [0-9.]*, "unit": "tiles/s", "n_gpus"'
done
