# round-2 GPU call 46: kernel trace of the MIL `vit` training step
set -x
R=$GRAFT_REPO_ROOT; mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace -d /tmp/kt -o kt -- python $R/tools/train_only.py 6 > /tmp/kt.log 2>&1 < /dev/null
tail -2 /tmp/kt.log
DB=$(find /tmp/kt -name "*.db" | head -1)
[ -n "$DB" ] && timeout 60 python $R/tools/rocprof_summary.py "$DB" > $R/gpurun_out/r02_mil_train_kernel_stats.txt
head -40 $R/gpurun_out/r02_mil_train_kernel_stats.txt | cut -c1-200
