# TransMIL training step under rocprofv3 (kernel trace): per-kernel and per-shape tables into gpurun_out/ (copy to profiles/).  gpurun -- 'bash tools/r06_transmil_trace.sh'
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 300 rocprofv3 --kernel-trace -d /tmp/kt -o kt -- python $R/tools/transmil_train_only.py 64 5 > /tmp/kt.log 2>&1 < /dev/null
DB=$(find /tmp/kt -name "*.db" | head -1); [ -n "$DB" ] && timeout 60 python $R/tools/rocprof_summary.py "$DB" > $R/gpurun_out/r06_transmil_train_high_kernel_stats.txt
tail -1 /tmp/kt.log >> $R/gpurun_out/r06_transmil_train_high_kernel_stats.txt
[ -n "$DB" ] && timeout 60 python $R/tools/rocprof_summary.py "$DB" --by-shape > $R/gpurun_out/r06_transmil_train_high_by_shape.txt
[ -n "$DB" ] && timeout 60 python $R/tools/rocprof_summary.py "$DB" --seq 420 > $R/gpurun_out/r06_transmil_train_high_sequence.txt
