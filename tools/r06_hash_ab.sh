#!/bin/bash
# One-multiply dropout hash: the training-path tests on the new library, then the MIL vit training rate alternating with stamp_amd/lib/libamdstamp_prev.so
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
OUT=$R/gpurun_out/r06_hash_ab.txt
: > $OUT
( cd $R && timeout 1200 python -m pytest tests/test_gpu_train.py tests/test_gpu_mil_seam.py tests/test_gpu_mil.py tests/test_gpu_transmil_train.py -x -q 2>&1 | tail -3 ) >> $OUT
for rep in 1 2 3; do
  ( cd $R && AMDSTAMP_LIB=$R/stamp_amd/lib/libamdstamp_prev.so timeout 200 python tools/train_only.py 30 | tail -1 | sed "s/^/murmur finaliser per pair /"; timeout 200 python tools/train_only.py 30 | tail -1 | sed "s/^/one multiply per pair     /" ) >> $OUT
done
rm -rf /tmp/kt; timeout 300 rocprofv3 --kernel-trace -d /tmp/kt -o kt -- python $R/tools/train_only.py 8 > /tmp/kt.log 2>&1 < /dev/null
DB=$(find /tmp/kt -name "*.db" | head -1)
[ -n "$DB" ] && timeout 60 python $R/tools/rocprof_summary.py "$DB" | grep -E "attn_|TOTAL" | cut -c1-60,110-180 >> $OUT
cat $OUT
