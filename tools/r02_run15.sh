set -x
R=$GRAFT_REPO_ROOT; mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace -d /tmp/ks -o ks -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-secondary --e2e-tiles 0 > /tmp/ks.log 2>&1 < /dev/null
DB=$(find /tmp/ks -name "*.db" | head -1); echo "db=$DB"
[ -n "$DB" ] && timeout 60 python $R/tools/rocprof_summary.py "$DB" --by-shape 2 > $R/gpurun_out/r02_run15_by_shape.txt
head -24 $R/gpurun_out/r02_run15_by_shape.txt | cut -c1-60,100-220
