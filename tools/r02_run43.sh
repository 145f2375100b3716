# round-2 GPU call 43: fp32 batched GEMM, k-major LDS image for k-major operands -- parity, yardstick, TransMIL train, counters
set -x
R=$GRAFT_REPO_ROOT; mkdir -p $R/gpurun_out
cd $R
( timeout 900 python -m pytest tests -q -m gpu -x -k "transmil or bgemm or nystrom or pinv" 2>&1 | tail -3 )
timeout 300 python tools/bgemm_f32_yardstick.py 2>&1 | grep -v amdgpu.ids
for L in build/ab/libamdstamp_old.so stamp_amd/lib/libamdstamp.so build/ab/libamdstamp_old.so stamp_amd/lib/libamdstamp.so; do
  echo "lib=$L"
  AMDSTAMP_LIB=$R/$L timeout 200 python tools/transmil_train_only.py 64 6 2>&1 | tail -1
  AMDSTAMP_LIB=$R/$L timeout 200 python tools/transmil_only.py 2>&1 | tail -1
done
