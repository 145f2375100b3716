#!/bin/bash
# Round 6 re-entry: GPU tests of the whole tree, then the MIL vit training rate and a kernel trace of the training step (baseline for the attention-backward work)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
( cd $R && timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/gputests.log 2>&1; echo "pytest rc=$?" >> gpurun_out/gputests.log; tail -4 gpurun_out/gputests.log )
( cd $R && timeout 200 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 )
for rep in 1 2; do ( cd $R && timeout 200 python tools/train_only.py 30 | tail -1 ) >> $R/gpurun_out/r06_train_only_base.txt; done
timeout 300 rocprofv3 --kernel-trace -d /tmp/kt -o kt -- python $R/tools/train_only.py 8 > /tmp/kt.log 2>&1 < /dev/null
DB=$(find /tmp/kt -name "*.db" | head -1); [ -n "$DB" ] && timeout 60 python $R/tools/rocprof_summary.py "$DB" > $R/gpurun_out/r06_mil_train_base_kernel_stats.txt
cat $R/gpurun_out/r06_train_only_base.txt; head -40 $R/gpurun_out/r06_mil_train_base_kernel_stats.txt
