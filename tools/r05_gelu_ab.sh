#!/bin/bash
# A/B of the GELU polynomial's degree in the fc1 epilogue (verdict r04 item 1 (f)): libraries built with make GELU_DEG=8 / 7 / 6 (build/ab/libamdstamp_g8.so, _g7.so, _g6.so),
# alternating: full-size feature error against the fp32 oracle (tests/test_gpu_vit.py prints it) and the headline leg of bench.py
cd "$(dirname "$0")/.."
run() { AMDSTAMP_LIB=$PWD/build/ab/libamdstamp_$1.so "${@:2}"; }
if [ "$1" != "speed" ]; then
for v in g8 g7 g6; do
  echo "== $v: feature error (ViT-L/14 and the GELU presets)"
  run $v python -m pytest tests/test_gpu_vit.py -q -s -k "vit_large_matches or full_size" 2>&1 | grep -E "rel-L2|passed|failed"
done
fi
for rep in 1 2 3 4; do
  for v in g8 g7 g6; do
    r=$(run $v python bench.py --steps 12 --warmup 3 --no-cpu-baseline --no-secondary --e2e-tiles 0 --slide-tiles 0 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])")
    echo "rep $rep $v: $r"
  done
done
