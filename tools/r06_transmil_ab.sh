#!/bin/bash
# TransMIL: tests on the new library, then forward / training rate alternating with stamp_amd/lib/libamdstamp_prev.so, then a kernel trace of the training step
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
OUT=$R/gpurun_out/r06_transmil_ab.txt
: > $OUT
( cd $R && timeout 1200 python -m pytest tests/test_gpu_transmil_train.py tests/test_gpu_mil.py tests/test_gpu_seams.py -x -q 2>&1 | tail -3 ) >> $OUT
for rep in 1 2; do
  ( cd $R && AMDSTAMP_LIB=$R/stamp_amd/lib/libamdstamp_prev.so timeout 300 python tools/transmil_train_only.py 64 6 | tail -1 | sed "s/^/prev train /"; timeout 300 python tools/transmil_train_only.py 64 6 | tail -1 | sed "s/^/new  train /" ) >> $OUT
  ( cd $R && AMDSTAMP_LIB=$R/stamp_amd/lib/libamdstamp_prev.so timeout 300 python tools/transmil_only.py | tail -1 | sed "s/^/prev fwd /"; timeout 300 python tools/transmil_only.py | tail -1 | sed "s/^/new  fwd /" ) >> $OUT
done
rm -rf /tmp/kt; ( cd $R && timeout 300 rocprofv3 --kernel-trace -d /tmp/kt -o kt -- python tools/transmil_train_only.py 64 8 > /tmp/kt.log 2>&1 < /dev/null )
DB=$(find /tmp/kt -name "*.db" | head -1)
[ -n "$DB" ] && timeout 60 python $R/tools/rocprof_summary.py "$DB" > $R/gpurun_out/r06_transmil_train_kernel_stats_b.txt
head -14 $R/gpurun_out/r06_transmil_train_kernel_stats_b.txt | cut -c1-70,100-175 >> $OUT
cat $OUT
