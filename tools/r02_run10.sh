set -x
R=$GRAFT_REPO_ROOT; mkdir -p $R/gpurun_out
cd $R
for i in 1 2; do for pk in 0 1; do
echo "pk=$pk"; AMDS_ATTN_PK=$pk timeout 120 python tools/attn_only.py 1020 257 16 2>&1 | tail -1
done; done > gpurun_out/r02_run10_attn.log 2>&1
cat gpurun_out/r02_run10_attn.log
