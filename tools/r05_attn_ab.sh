#!/bin/bash
# A/B of the stand-alone T = 257 attention kernel: this tree's library against build/ab/libamdstamp_old.so (the tree without the change), alternating:
# the kernel alone (tools/attn_only.py 1020 257 16) and the headline leg of bench.py
cd "$(dirname "$0")/.."
for rep in 1 2 3; do
  echo "rep $rep new: $(python tools/attn_only.py 1020 257 16 | tail -1)"
  echo "rep $rep old: $(AMDSTAMP_LIB=$PWD/build/ab/libamdstamp_old.so python tools/attn_only.py 1020 257 16 | tail -1)"
done
for rep in 1 2 3; do
  for v in new old; do
    if [ $v = old ]; then export AMDSTAMP_LIB=$PWD/build/ab/libamdstamp_old.so; else unset AMDSTAMP_LIB; fi
    r=$(python bench.py --steps 12 --warmup 3 --no-cpu-baseline --no-secondary --e2e-tiles 0 --slide-tiles 0 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])")
    echo "rep $rep $v: $r"
  done
done
