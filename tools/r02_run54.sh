# round-2 GPU call 54: fp32 batched GEMM at four workgroups per CU (__launch_bounds__(256, 4)) -- yardstick, TransMIL, parity
set -x
R=$GRAFT_REPO_ROOT; cd $R
timeout 300 python tools/bgemm_f32_yardstick.py 2>&1 | grep -v amdgpu.ids
for i in 1 2; do
  timeout 200 python tools/transmil_train_only.py 64 6 2>&1 | tail -1
  timeout 200 python tools/transmil_only.py 2>&1 | tail -1
done
( timeout 900 python -m pytest tests -q -m gpu -x -k "transmil or bgemm" 2>&1 | tail -3 )
