# round-2 GPU call 60: fp32 batched GEMM with a check-free K loop for interior tiles -- parity, yardstick, TransMIL
set -x
R=$GRAFT_REPO_ROOT; cd $R
( timeout 900 python -m pytest tests -q -m gpu -x -k "transmil or bgemm" 2>&1 | tail -3 )
timeout 300 python tools/bgemm_f32_yardstick.py 2>&1 | grep "^Z=" | sed 's/(err [^)]*)//g'
for i in 1 2; do
  timeout 200 python tools/transmil_train_only.py 64 6 2>&1 | tail -1
  timeout 200 python tools/transmil_only.py 2>&1 | tail -1
done
