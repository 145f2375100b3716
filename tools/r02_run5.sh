# round-2 GPU call 5: tiling tests, ViT-only kernel trace (per shape), PMC FETCH / WRITE passes
set -x
R=$GRAFT_REPO_ROOT; mkdir -p $R/gpurun_out
cd $R
timeout 900 python -m pytest tests/test_gpu_tiling.py tests/test_gpu_seams.py -q -m gpu 2>&1 | tail -15 > gpurun_out/r02_pytest_tiling.log
cat gpurun_out/r02_pytest_tiling.log
cd /tmp && export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace -d /tmp/ks -o ks -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-secondary > /tmp/ks.log 2>&1 < /dev/null
grep '^{"metric"' /tmp/ks.log > $R/gpurun_out/r02_vitonly_bench.json
DB=$(find /tmp/ks -name "*.db" | head -1); echo "db=$DB"
[ -n "$DB" ] && timeout 60 python $R/tools/rocprof_summary.py "$DB" > $R/gpurun_out/r02_vitonly_kernel_stats.txt
[ -n "$DB" ] && timeout 60 python $R/tools/rocprof_summary.py "$DB" --by-shape 2 > $R/gpurun_out/r02_vitonly_kernel_stats_by_shape.txt
head -12 $R/gpurun_out/r02_vitonly_kernel_stats_by_shape.txt
timeout 400 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d /tmp/pf -o pf -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-secondary > /tmp/pf.log 2>&1 < /dev/null
timeout 400 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d /tmp/pw -o pw -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-secondary > /tmp/pw.log 2>&1 < /dev/null
F=$(find /tmp/pf -name "*.db" | head -1); W=$(find /tmp/pw -name "*.db" | head -1); echo "f=$F w=$W"
[ -n "$F" ] && [ -n "$W" ] && timeout 60 python $R/tools/pmc_summary.py "$F" "$W" > $R/gpurun_out/r02_pmc_summary.txt
head -20 $R/gpurun_out/r02_pmc_summary.txt
