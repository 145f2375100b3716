#!/bin/bash
# A/B: v_fma_mix_f32 in the residual-planes epilogue (this tree) against build/ab/libamdstamp_g7.so (the tree before it, same GELU degree), alternating headline legs
cd "$(dirname "$0")/.."
for rep in 1 2 3 4; do
  for v in new old; do
    if [ $v = old ]; then export AMDSTAMP_LIB=$PWD/build/ab/libamdstamp_g7.so; else unset AMDSTAMP_LIB; fi
    r=$(python bench.py --steps 12 --warmup 3 --no-cpu-baseline --no-secondary --e2e-tiles 0 --slide-tiles 0 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])")
    echo "rep $rep $v: $r"
  done
done
