# round-2 GPU call 59: MIL training GEMMs with the ragged last row tile as its own launch -- parity, A/B by AMDS_GEMM_SPLIT
set -x
R=$GRAFT_REPO_ROOT; cd $R
( timeout 1500 python -m pytest tests -q -m gpu -x -k "mil or train" 2>&1 | tail -3 )
for i in 1 2 3; do for e in 0 1; do echo "AMDS_GEMM_SPLIT=$e"; AMDS_GEMM_SPLIT=$e timeout 200 python tools/train_only.py 8 2>&1 | tail -1; done; done
