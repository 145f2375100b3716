#!/bin/bash
# NOTE: the AMDS_ATTN_DKDV / _DQ / _FWD switches existed only while both forms of the kernels were in the library (commits 320cdb4, 3e683cf); the first
# forms now live as text under tools/ubench/attic/attention_first_forms/ -- this script documents how profiles/r06_attn_bwd_ab.txt was measured.
# The attention kernels on the LDS-DMA + transpose-read data path: the whole GPU suite on the new forms, then A/B against the first forms (AMDS_ATTN_FWD / _DQ / _DKDV = 1)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
OUT=$R/gpurun_out/r06_attn_ab3.txt
: > $OUT
( cd $R && timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/gputests.log 2>&1; echo "pytest rc=$?" >> gpurun_out/gputests.log; tail -3 gpurun_out/gputests.log | sed "s/^/new forms: /" ) >> $OUT
for rep in 1 2 3; do
  ( cd $R && AMDS_ATTN_FWD=1 AMDS_ATTN_DQ=1 timeout 200 python tools/train_only.py 30 | tail -1 | sed "s/^/fwd+dq first form /"; timeout 200 python tools/train_only.py 30 | tail -1 | sed "s/^/second forms      /" ) >> $OUT
done
for rep in 1 2; do
  ( cd $R && AMDS_ATTN_FWD=1 timeout 200 python tools/mil_fwd_only.py 30 | tail -1 | sed "s/^/fwd first form  /"; timeout 200 python tools/mil_fwd_only.py 30 | tail -1 | sed "s/^/fwd second form /" ) >> $OUT
done
( cd $R && AMDS_ATTN_FWD=1 AMDS_ATTN_DQ=1 timeout 200 python tools/train_only.py 30 0.25 1 | tail -1 | sed "s/^/fwd+dq first form /"; timeout 200 python tools/train_only.py 30 0.25 1 | tail -1 | sed "s/^/second forms      /" ) >> $OUT
for v in 1 2; do
  rm -rf /tmp/kt; AMDS_ATTN_FWD=$v AMDS_ATTN_DQ=$v timeout 300 rocprofv3 --kernel-trace -d /tmp/kt -o kt -- python $R/tools/train_only.py 8 > /tmp/kt.log 2>&1 < /dev/null
  DB=$(find /tmp/kt -name "*.db" | head -1)
  [ -n "$DB" ] && timeout 60 python $R/tools/rocprof_summary.py "$DB" > $R/gpurun_out/r06_attn_ab3_form${v}_kernel_stats.txt
  grep -E "attn_|TOTAL" $R/gpurun_out/r06_attn_ab3_form${v}_kernel_stats.txt | cut -c1-60,110-180 | sed "s/^/form $v /" >> $OUT
done
cat $OUT
