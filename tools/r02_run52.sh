# round-2 GPU call 52: register-window depthwise sequence convolution -- parity, A/B
set -x
R=$GRAFT_REPO_ROOT; cd $R
( timeout 900 python -m pytest tests -q -m gpu -x -k "transmil or dwconv" 2>&1 | tail -3 )
for w in 0 1 0 1; do
  echo "AMDS_DWCONV_WIN=$w"
  AMDS_DWCONV_WIN=$w timeout 200 python tools/transmil_train_only.py 64 6 2>&1 | tail -1
  AMDS_DWCONV_WIN=$w timeout 200 python tools/transmil_only.py 2>&1 | tail -1
done
