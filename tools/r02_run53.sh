# round-2 GPU call 53: wave-per-row softmax (forward / backward) in the TransMIL head -- parity, timing
set -x
R=$GRAFT_REPO_ROOT; cd $R
( timeout 900 python -m pytest tests -q -m gpu -x -k "transmil or softmax" 2>&1 | tail -3 )
for i in 1 2; do
  timeout 200 python tools/transmil_train_only.py 64 6 2>&1 | tail -1
  timeout 200 python tools/transmil_only.py 2>&1 | tail -1
done
