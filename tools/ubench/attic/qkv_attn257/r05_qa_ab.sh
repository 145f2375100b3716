#!/bin/bash
# Alternating A/B of the headline with the fused qkv + attention kernel on (default) and off (AMDS_VIT_QKVATTN=0): tiles/s of bench.py's headline leg only.
cd "$(dirname "$0")/.."
for rep in 1 2 3; do
  for v in 1 0; do
    r=$(AMDS_VIT_QKVATTN=$v python bench.py --steps 12 --warmup 3 --no-cpu-baseline --no-secondary --e2e-tiles 0 --slide-tiles 0 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])")
    echo "rep $rep AMDS_VIT_QKVATTN=$v: $r"
  done
done
