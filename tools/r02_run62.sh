set -x
R=$GRAFT_REPO_ROOT; cd $R
( timeout 1500 python -m pytest tests -q -m gpu -x -k "mil or train or layernorm or ln_" 2>&1 | tail -3 )
for i in 1 2 3; do timeout 200 python tools/train_only.py 8 2>&1 | tail -1; done
timeout 200 python tools/transmil_train_only.py 64 6 2>&1 | tail -1
