cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 300 rocprofv3 --kernel-trace -d /tmp/kt -o kt -- python $R/tools/train_only.py 8 > /tmp/kt.log 2>&1 < /dev/null
DB=$(find /tmp/kt -name "*.db" | head -1); [ -n "$DB" ] && timeout 60 python $R/tools/rocprof_summary.py "$DB" > $R/gpurun_out/r04_rocprofv3_mil_train_kernel_stats.txt
tail -1 /tmp/kt.log >> $R/gpurun_out/r04_rocprofv3_mil_train_kernel_stats.txt
