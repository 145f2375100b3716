# round-2 GPU call 42: fp32 batched GEMM with the XCD-aware tile order -- yardstick shapes, TransMIL train, parity
set -x
R=$GRAFT_REPO_ROOT; mkdir -p $R/gpurun_out
cd $R
for x in 0 1; do echo "AMDS_BGEMM_XCD=$x"; AMDS_BGEMM_XCD=$x timeout 300 python tools/bgemm_f32_yardstick.py 2>&1 | sed 's/| vendor.*//'; done
for x in 0 1 0 1; do echo "AMDS_BGEMM_XCD=$x"; AMDS_BGEMM_XCD=$x timeout 200 python tools/transmil_train_only.py 64 6 2>&1 | tail -1; done
( timeout 900 python -m pytest tests/test_gpu_transmil_train.py tests/test_gpu_transmil.py -q -m gpu -x 2>&1 | tail -3 )
