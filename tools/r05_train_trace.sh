#!/bin/bash
# MIL vit training step: GPU tests of the training path, stand-alone rate (alternating, A/B against the undeferred / unfused forms), then a rocprofv3 kernel trace of 10 steps
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
( cd $R && python -m pytest tests/test_gpu_train.py tests/test_gpu_mil_seam.py tests/test_gpu_mil.py -x -q 2>&1 | tail -4 ) > $R/gpurun_out/r05_train_tests.log
for rep in 1 2 3; do
  ( cd $R && python tools/train_only.py 30 | tail -1 | sed "s/^/new   /"; AMDS_COLSUM_DEFER=0 python tools/train_only.py 30 | tail -1 | sed "s/^/column sums on the spot /" ) >> $R/gpurun_out/r05_train_only.txt
done
timeout 300 rocprofv3 --kernel-trace -d /tmp/kt -o kt -- python $R/tools/train_only.py 8 > /tmp/kt.log 2>&1 < /dev/null
DB=$(find /tmp/kt -name "*.db" | head -1); [ -n "$DB" ] && timeout 60 python $R/tools/rocprof_summary.py "$DB" > $R/gpurun_out/r05_rocprofv3_mil_train_kernel_stats.txt
tail -1 /tmp/kt.log >> $R/gpurun_out/r05_rocprofv3_mil_train_kernel_stats.txt
cat $R/gpurun_out/r05_train_tests.log $R/gpurun_out/r05_train_only.txt; grep -E "TOTAL|train step" $R/gpurun_out/r05_rocprofv3_mil_train_kernel_stats.txt
