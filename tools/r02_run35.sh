set -x
R=$GRAFT_REPO_ROOT; mkdir -p $R/gpurun_out
cd $R
for i in 1 2; do for sp in 0 1; do echo "split=$sp"; AMDS_ATTN_SPLIT=$sp timeout 120 python tools/attn_only.py 1020 257 16 2>&1 | tail -1; done; done
( timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_vit.py -q -m gpu -k "attention or large_matches or invariance" 2>&1 | tail -3 )
