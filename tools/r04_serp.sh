# Round 4: serpentine-K tile walk (kernel id 14) against id 12: time (sustained) and L2-side fetch bytes per launch.
set -x
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
timeout 400 python $R/tools/gemm_sched_ab.py 262140 12,14 > $O/r04_serp_time.txt 2>&1
cat $O/r04_serp_time.txt
for s in qkv fc1 proj; do for c in 12 14; do
  rm -rf /tmp/pm; timeout 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d /tmp/pm -o pm -- python $R/tools/gemm_only.py $s $c 1020 > /tmp/pm.log 2>&1 < /dev/null
  DB=$(find /tmp/pm -name "*.db" | head -1)
  echo "== $s cfg $c" >> $O/r04_serp_fetch.txt
  [ -n "$DB" ] && timeout 60 python $R/tools/pmc_dump.py "$DB" gemm >> $O/r04_serp_fetch.txt
done; done
cat $O/r04_serp_fetch.txt
