set -x
cd /tmp && export TMPDIR=/tmp
R=/root/repo
timeout 400 rocprofv3 --kernel-trace -d /tmp/ks -o ks -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > /tmp/ks.log 2>&1 < /dev/null
tail -c 300 /tmp/ks.log
DB=$(find /tmp/ks -name "*.db" | head -1); echo "db=$DB"
[ -n "$DB" ] && timeout 60 python $R/tools/rocprof_summary.py "$DB" > $R/gpurun_out/kernel_stats.txt
timeout 400 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d /tmp/pf -o pf -- python $R/bench.py --steps 1 --warmup 1 --tiles 1020 --no-cpu-baseline > /tmp/pf.log 2>&1 < /dev/null
timeout 400 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d /tmp/pw -o pw -- python $R/bench.py --steps 1 --warmup 1 --tiles 1020 --no-cpu-baseline > /tmp/pw.log 2>&1 < /dev/null
F=$(find /tmp/pf -name "*.db" | head -1); W=$(find /tmp/pw -name "*.db" | head -1); echo "f=$F w=$W"
[ -n "$F" ] && [ -n "$W" ] && timeout 60 python $R/tools/pmc_summary.py "$F" "$W" gemm layernorm attn > $R/gpurun_out/pmc_summary.txt
head -30 $R/gpurun_out/pmc_summary.txt
