# round-2 GPU call 48: pipelined attention, LDS prefetch distance 2 / 3, against the old library; then ViT parity + whole-path A/B
set -x
R=$GRAFT_REPO_ROOT; mkdir -p $R/gpurun_out
cd $R
for i in 1 2; do
  for L in build/ab/libamdstamp_old.so stamp_amd/lib/libamdstamp.so build/ab/libamdstamp_d3.so; do
    echo "lib=$L"
    AMDSTAMP_LIB=$R/$L timeout 120 python tools/attn_only.py 1020 257 16 2>&1 | tail -1
  done
done
( timeout 900 python -m pytest tests/test_gpu_vit.py tests/test_gpu_kernels.py -q -m gpu -x 2>&1 | tail -3 )
for i in 1 2 3; do
  for L in stamp_amd/lib/libamdstamp.so build/ab/libamdstamp_old.so; do
    echo "lib=$L"
    AMDSTAMP_LIB=$R/$L timeout 300 python bench.py --no-secondary --no-cpu-baseline --e2e-tiles 0 2>&1 | grep -o '"value": [0-9.]*, "unit": "tiles/s", "n_gpus"'
  done
done
