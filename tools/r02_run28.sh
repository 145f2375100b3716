set -x
R=$GRAFT_REPO_ROOT; mkdir -p $R/gpurun_out
cd $R
for p in 1 0 1 0; do
echo "side priority high=$p"; AMDS_SIDE_PRIORITY=$p timeout 300 python tools/b64_only.py 24 2>&1 | tail -1
done > gpurun_out/r02_run28.log 2>&1
AMDS_VIT_TAIL=0 timeout 300 python tools/b64_only.py 24 2>&1 | tail -1 >> gpurun_out/r02_run28.log
cat gpurun_out/r02_run28.log
