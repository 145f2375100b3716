# round-2 GPU call 16: full GPU suite, default bench line (LayerNorm folded), ViT-only kernel trace, PMC traffic passes, other presets
set -x
R=$GRAFT_REPO_ROOT; mkdir -p $R/gpurun_out
cd $R
( time timeout 3000 python -m pytest tests -q -m gpu 2>&1 | tail -40 ) > gpurun_out/r02_pytest_all.log 2>&1
grep -E "passed|failed" gpurun_out/r02_pytest_all.log | tail -3
( time timeout 900 python bench.py ) > gpurun_out/r02_bench_default.log 2>&1
tail -c 1200 gpurun_out/r02_bench_default.log
AMDS_VIT_LNFOLD=0 timeout 300 python bench.py --no-secondary --no-cpu-baseline --e2e-tiles 0 > gpurun_out/r02_bench_nofold.log 2>&1
grep -o '"value": [0-9.]*, "unit": "tiles/s", "n_gpus"' gpurun_out/r02_bench_nofold.log
for m in uni2_h virchow2; do
  timeout 400 python bench.py --model $m --steps 4 --warmup 1 --no-cpu-baseline --no-secondary --e2e-tiles 0 > gpurun_out/r02_bench_$m.log 2>&1
  grep -o '"value": [0-9.]*, "unit": "tiles/s", "n_gpus"' gpurun_out/r02_bench_$m.log
done
timeout 300 python -c "
import __graft_entry__ as g; g.smoke()" > gpurun_out/r02_smoke.log 2>&1; tail -2 gpurun_out/r02_smoke.log
cd /tmp && export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace -d /tmp/ks -o ks -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-secondary --e2e-tiles 0 > /tmp/ks.log 2>&1 < /dev/null
DB=$(find /tmp/ks -name "*.db" | head -1); echo "db=$DB"
[ -n "$DB" ] && timeout 60 python $R/tools/rocprof_summary.py "$DB" > $R/gpurun_out/r02_kernel_stats.txt
[ -n "$DB" ] && timeout 60 python $R/tools/rocprof_summary.py "$DB" --by-shape 2 > $R/gpurun_out/r02_kernel_stats_by_shape.txt
timeout 400 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d /tmp/pf -o pf -- python $R/bench.py --steps 1 --warmup 1 --tiles 1020 --no-cpu-baseline --no-secondary --e2e-tiles 0 > /tmp/pf.log 2>&1 < /dev/null
timeout 400 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d /tmp/pw -o pw -- python $R/bench.py --steps 1 --warmup 1 --tiles 1020 --no-cpu-baseline --no-secondary --e2e-tiles 0 > /tmp/pw.log 2>&1 < /dev/null
F=$(find /tmp/pf -name "*.db" | head -1); W=$(find /tmp/pw -name "*.db" | head -1); echo "f=$F w=$W"
[ -n "$F" ] && [ -n "$W" ] && timeout 60 python $R/tools/pmc_summary.py "$F" "$W" gemm layernorm attn ln_ > $R/gpurun_out/r02_pmc_summary.txt
head -30 $R/gpurun_out/r02_pmc_summary.txt
