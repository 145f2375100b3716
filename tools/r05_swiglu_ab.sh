#!/bin/bash
# A/B of the packed SWIGLU epilogue (this tree) against build/ab/libamdstamp_head.so (the tree before), alternating, UNI2-h and Virchow2 shapes
cd "$(dirname "$0")/.."
for m in uni2_h virchow2; do
for rep in 1 2 3; do
  for v in new old; do
    if [ $v = old ]; then export AMDSTAMP_LIB=$PWD/build/ab/libamdstamp_head.so; else unset AMDSTAMP_LIB; fi
    r=$(python bench.py --model $m --steps 8 --warmup 2 --no-cpu-baseline --no-secondary --e2e-tiles 0 --slide-tiles 0 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])")
    echo "$m rep $rep $v: $r"
  done
done
done
