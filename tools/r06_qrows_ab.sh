#!/bin/bash
# the Q third of the tail layer's in-projection gradients on the class rows alone -- tests, alternating rate against libamdstamp_prev.so
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
OUT=$R/gpurun_out/r06_qrows_ab.txt
: > $OUT
( cd $R && timeout 1200 python -m pytest tests/test_gpu_train.py tests/test_gpu_mil_seam.py tests/test_gpu_mil.py -x -q 2>&1 | tail -3 ) >> $OUT
for rep in 1 2 3; do
  ( cd $R && AMDSTAMP_LIB=$R/stamp_amd/lib/libamdstamp_prev.so timeout 200 python tools/train_only.py 30 | tail -1 | sed "s/^/every row       /"; timeout 200 python tools/train_only.py 30 | tail -1 | sed "s/^/Q on class rows /" ) >> $OUT
done
( cd $R && AMDSTAMP_LIB=$R/stamp_amd/lib/libamdstamp_prev.so timeout 200 python tools/train_only.py 30 0.25 1 | tail -1 | sed "s/^/every row       /"; timeout 200 python tools/train_only.py 30 0.25 1 | tail -1 | sed "s/^/Q on class rows /" ) >> $OUT
cat $OUT
