#!/bin/bash
# Attention kernels after the VALU diet (shared dropout hashes, no bounds tests in full tiles): training-path GPU tests, MIL vit training rate, kernel trace
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
TAG=${1:-a}
mkdir -p $R/gpurun_out
( cd $R && timeout 1200 python -m pytest tests/test_gpu_train.py tests/test_gpu_mil_seam.py tests/test_gpu_mil.py -x -q 2>&1 | tail -4 ) > $R/gpurun_out/r06_attn_${TAG}_tests.log
for rep in 1 2; do ( cd $R && timeout 200 python tools/train_only.py 30 | tail -1; timeout 200 python tools/train_only.py 30 0.25 1 | tail -1 ) >> $R/gpurun_out/r06_attn_${TAG}_train_only.txt; done
timeout 300 rocprofv3 --kernel-trace -d /tmp/kt -o kt -- python $R/tools/train_only.py 8 > /tmp/kt.log 2>&1 < /dev/null
DB=$(find /tmp/kt -name "*.db" | head -1); [ -n "$DB" ] && timeout 60 python $R/tools/rocprof_summary.py "$DB" > $R/gpurun_out/r06_attn_${TAG}_kernel_stats.txt
cat $R/gpurun_out/r06_attn_${TAG}_tests.log $R/gpurun_out/r06_attn_${TAG}_train_only.txt; head -12 $R/gpurun_out/r06_attn_${TAG}_kernel_stats.txt | cut -c1-180
