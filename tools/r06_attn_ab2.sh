#!/bin/bash
# NOTE: the AMDS_ATTN_DKDV / _DQ / _FWD switches existed only while both forms of the kernels were in the library (commits 320cdb4, 3e683cf); the first
# forms now live as text under tools/ubench/attic/attention_first_forms/ -- this script documents how profiles/r06_attn_bwd_ab.txt was measured.
# A/B of the two dK/dV kernels in one library (AMDS_ATTN_DKDV=1: first form): tests under both, alternating training rate, kernel trace of each
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
OUT=$R/gpurun_out/r06_attn_ab2.txt
: > $OUT
( cd $R && timeout 900 python -m pytest tests/test_gpu_train.py tests/test_gpu_mil_seam.py -x -q 2>&1 | tail -3 | sed "s/^/new tests: /" ) >> $OUT
for rep in 1 2 3; do
  ( cd $R && AMDS_ATTN_DKDV=1 timeout 200 python tools/train_only.py 30 | tail -1 | sed "s/^/first form  /"; timeout 200 python tools/train_only.py 30 | tail -1 | sed "s/^/second form /" ) >> $OUT
done
( cd $R && AMDS_ATTN_DKDV=1 timeout 200 python tools/train_only.py 30 0.25 1 | tail -1 | sed "s/^/first form  /"; timeout 200 python tools/train_only.py 30 0.25 1 | tail -1 | sed "s/^/second form /" ) >> $OUT
( cd $R && AMDS_ATTN_DKDV=1 timeout 200 python tools/train_only.py 30 0.25 0 medium | tail -1 | sed "s/^/first form  /"; timeout 200 python tools/train_only.py 30 0.25 0 medium | tail -1 | sed "s/^/second form /" ) >> $OUT
for v in 1 2; do
  rm -rf /tmp/kt; AMDS_ATTN_DKDV=$v timeout 300 rocprofv3 --kernel-trace -d /tmp/kt -o kt -- python $R/tools/train_only.py 8 > /tmp/kt.log 2>&1 < /dev/null
  DB=$(find /tmp/kt -name "*.db" | head -1)
  [ -n "$DB" ] && timeout 60 python $R/tools/rocprof_summary.py "$DB" > $R/gpurun_out/r06_attn_ab2_form${v}_kernel_stats.txt
  grep -E "attn_|TOTAL" $R/gpurun_out/r06_attn_ab2_form${v}_kernel_stats.txt | cut -c1-60,110-180 | sed "s/^/form $v /" >> $OUT
done
cat $OUT
