set -x
R=$GRAFT_REPO_ROOT; mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace -d /tmp/kt -o kt -- python $R/tools/train_only.py > /tmp/kt.log 2>&1 < /dev/null
tail -3 /tmp/kt.log
DB=$(find /tmp/kt -name "*.db" | head -1); echo "db=$DB"
[ -n "$DB" ] && timeout 60 python $R/tools/rocprof_summary.py "$DB" > $R/gpurun_out/r02_mil_train_kernel_stats.txt
head -32 $R/gpurun_out/r02_mil_train_kernel_stats.txt | cut -c1-100,110-200
