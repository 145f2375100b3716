# round-2 GPU call 47: software-pipelined T = 257 attention -- ViT parity + whole-path A/B (alternating libraries)
set -x
R=$GRAFT_REPO_ROOT; mkdir -p $R/gpurun_out
cd $R
( timeout 900 python -m pytest tests/test_gpu_vit.py tests/test_gpu_kernels.py -q -m gpu -x 2>&1 | tail -3 )
for i in 1 2 3; do
  for L in stamp_amd/lib/libamdstamp.so build/ab/libamdstamp_old.so; do
    echo "lib=$L"
    AMDSTAMP_LIB=$R/$L timeout 300 python bench.py --no-secondary --no-cpu-baseline --e2e-tiles 0 2>&1 | grep -o '"value": [0-9.]*, "unit": "tiles/s", "n_gpus"'
  done
done
