#!/bin/bash
# Where the MIL legs' time is: kernel traces of the deploy forwards (vit head, TransMIL) and the TransMIL training step, busy fraction + top kernels
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
for leg in "mil_fwd tools/mil_fwd_only.py 30" "transmil_fwd tools/transmil_only.py" "transmil_train tools/transmil_train_only.py 64 6"; do
  set -- $leg; name=$1; shift
  rm -rf /tmp/kt; ( cd $R && timeout 300 rocprofv3 --kernel-trace -d /tmp/kt -o kt -- python "$@" > /tmp/kt_$name.log 2>&1 < /dev/null )
  DB=$(find /tmp/kt -name "*.db" | head -1)
  O=$R/gpurun_out/r06_trace_$name.txt
  tail -3 /tmp/kt_$name.log | grep -v rocprofv3 > $O
  [ -n "$DB" ] && ( cd $R/tools && timeout 60 python rocprof_gaps.py "$DB" 30 | head -12 ) >> $O
  [ -n "$DB" ] && timeout 60 python $R/tools/rocprof_summary.py "$DB" | head -28 | cut -c1-80,110-180 >> $O
  head -45 $O
done
