# Round 4 follow-up lease: MIL-train launch deletions (tests + trace), UNI2-h by-shape trace, guard A/B on one box.
set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_train.py tests/test_gpu_mil.py tests/test_gpu_mil_seam.py tests/test_gpu_kernels.py -x -q 2>&1 | tail -4 > $O/r04_followup_tests.log; cat $O/r04_followup_tests.log
for i in 1 2; do timeout 200 python tools/train_only.py 20 2>&1 | tail -1; done > $O/r04_train_only.txt; cat $O/r04_train_only.txt
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 300 rocprofv3 --kernel-trace -d /tmp/kt -o kt -- python $R/tools/train_only.py 8 > /tmp/kt.log 2>&1 < /dev/null
DB=$(find /tmp/kt -name "*.db" | head -1); [ -n "$DB" ] && timeout 60 python $R/tools/rocprof_summary.py "$DB" > $R/$O/r04_rocprofv3_mil_train_kernel_stats.txt
tail -1 /tmp/kt.log >> $R/$O/r04_rocprofv3_mil_train_kernel_stats.txt
timeout 500 rocprofv3 --kernel-trace -d /tmp/ku -o ku -- python $R/bench.py --model uni2_h --steps 3 --warmup 1 --no-cpu-baseline --no-secondary --e2e-tiles 0 > /tmp/ku.log 2>&1 < /dev/null
DB=$(find /tmp/ku -name "*.db" | head -1); [ -n "$DB" ] && timeout 60 python $R/tools/rocprof_summary.py "$DB" --by-shape 2 > $R/$O/r04_rocprofv3_uni2h_kernel_stats_by_shape.txt
cd $R
for i in 1 2; do for c in fallback off; do
  echo "check=$c: $(timeout 300 python bench.py --check $c --steps 10 --warmup 2 --no-cpu-baseline --no-secondary --e2e-tiles 0 2>/dev/null | tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["value"], d["roofline"]["frac"])')"
done; done > $O/r04_guard_ab.txt; cat $O/r04_guard_ab.txt
for i in 1 2; do for e in 1 0; do echo "AMDS_ATTN_26X=$e: $(AMDS_ATTN_26X=$e timeout 300 python bench.py --model uni2_h --steps 6 --warmup 2 --no-cpu-baseline --no-secondary --e2e-tiles 0 2>/dev/null | tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["value"])')"; done; done > $O/r04_uni2h_attn_ab.txt; cat $O/r04_uni2h_attn_ab.txt
