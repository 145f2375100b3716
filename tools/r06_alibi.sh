#!/bin/bash
# ALiBi head: tests, training rate (AMDSTAMP_LIB = previous build for A/B when present), kernel trace of the training step
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
OUT=$R/gpurun_out/r06_alibi.txt
: > $OUT
( cd $R && timeout 1200 python -m pytest tests/test_gpu_train.py tests/test_gpu_mil_seam.py tests/test_gpu_mil.py tests/test_gpu_kernels.py -x -q 2>&1 | tail -3 ) >> $OUT
for rep in 1 2 3; do
  [ -f $R/stamp_amd/lib/libamdstamp_prev.so ] && ( cd $R && AMDSTAMP_LIB=$R/stamp_amd/lib/libamdstamp_prev.so timeout 200 python tools/train_only.py 30 0.25 1 | tail -1 | sed "s/^/prev /" ) >> $OUT
  ( cd $R && timeout 200 python tools/train_only.py 30 0.25 1 | tail -1 | sed "s/^/new  /" ) >> $OUT
done
rm -rf /tmp/kt; timeout 300 rocprofv3 --kernel-trace -d /tmp/kt -o kt -- python $R/tools/train_only.py 8 0.25 1 > /tmp/kt.log 2>&1 < /dev/null
DB=$(find /tmp/kt -name "*.db" | head -1)
[ -n "$DB" ] && timeout 60 python $R/tools/rocprof_summary.py "$DB" > $R/gpurun_out/r06_alibi_train_kernel_stats.txt
head -22 $R/gpurun_out/r06_alibi_train_kernel_stats.txt | cut -c1-70,110-180 >> $OUT
cat $OUT
