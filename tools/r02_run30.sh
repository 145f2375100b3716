set -x
R=$GRAFT_REPO_ROOT; mkdir -p $R/gpurun_out
cd $R
for t in 1 4 -1 1 4 -1; do
echo "tail tiles=$t"; AMDS_VIT_TAIL_TILES=$t timeout 300 python tools/b64_only.py 24 2>&1 | tail -1
done > gpurun_out/r02_run30.log 2>&1
cat gpurun_out/r02_run30.log
