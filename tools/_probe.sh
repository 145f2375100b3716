cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_vit.py tests/test_gpu_seams.py -x -q 2>&1 | tail -4
timeout 600 python bench.py --no-cpu-baseline --no-secondary --e2e-tiles 0 2> gpurun_out/bench_err.log | tail -1 > gpurun_out/bench_quick.json
python - <<'PY'
import json; d=json.load(open('gpurun_out/bench_quick.json')); print(d['value'], d['roofline']['achieved'], d['roofline']['time_share'], d.get('exact_mode'))
PY
timeout 600 python bench.py --no-cpu-baseline --no-secondary --e2e-tiles 0 --model uni2_h 2>> gpurun_out/bench_err.log | tail -1 > gpurun_out/bench_quick_uni2.json
python - <<'PY'
import json; d=json.load(open('gpurun_out/bench_quick_uni2.json')); print(d['value'], d['roofline']['achieved'], d.get('exact_mode'))
PY
tail -3 gpurun_out/bench_err.log
