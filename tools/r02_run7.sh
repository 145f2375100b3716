# round-2 GPU call 7: full GPU suite, default bench line, other presets, MIL train kernel stats
set -x
R=$GRAFT_REPO_ROOT; mkdir -p $R/gpurun_out
cd $R
( time timeout 3000 python -m pytest tests -q -m gpu 2>&1 | tail -40 ) > gpurun_out/r02_pytest_all.log 2>&1
grep -E "passed|failed" gpurun_out/r02_pytest_all.log | tail -3
( time timeout 900 python bench.py ) > gpurun_out/r02_bench_default.log 2>&1
tail -c 1500 gpurun_out/r02_bench_default.log
for m in uni2_h virchow2; do
  timeout 400 python bench.py --model $m --steps 4 --warmup 1 --no-cpu-baseline --no-secondary > gpurun_out/r02_bench_$m.log 2>&1
  grep -o '"value": [0-9.]*, "unit": "tiles/s", "n_gpus"' gpurun_out/r02_bench_$m.log
done
timeout 300 python -c "
import __graft_entry__ as g; g.smoke()" > gpurun_out/r02_smoke.log 2>&1; tail -2 gpurun_out/r02_smoke.log
