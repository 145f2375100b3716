# round-2 GPU call 55: kernel traces of the Virchow2 and UNI2-h presets
set -x
R=$GRAFT_REPO_ROOT; mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
for m in virchow2 uni2_h; do
  timeout 400 rocprofv3 --kernel-trace -d /tmp/k_$m -o ks -- python $R/bench.py --model $m --steps 2 --warmup 1 --no-cpu-baseline --no-secondary --e2e-tiles 0 > /tmp/k_$m.log 2>&1 < /dev/null
  DB=$(find /tmp/k_$m -name "*.db" | head -1)
  [ -n "$DB" ] && timeout 60 python $R/tools/rocprof_summary.py "$DB" > $R/gpurun_out/r02_kernel_stats_$m.txt
  head -14 $R/gpurun_out/r02_kernel_stats_$m.txt | cut -c1-185
done
