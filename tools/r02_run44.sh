# round-2 GPU call 44: kernel trace of a TransMIL training step after the LDS-image change
set -x
R=$GRAFT_REPO_ROOT; mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace -d /tmp/kt -o kt -- python $R/tools/transmil_train_only.py 64 3 > /tmp/kt.log 2>&1 < /dev/null
tail -1 /tmp/kt.log
DB=$(find /tmp/kt -name "*.db" | head -1)
[ -n "$DB" ] && timeout 60 python $R/tools/rocprof_summary.py "$DB" > $R/gpurun_out/r02_transmil_train_kernel_stats.txt
head -32 $R/gpurun_out/r02_transmil_train_kernel_stats.txt | cut -c1-190
