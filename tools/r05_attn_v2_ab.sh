#!/bin/bash
# (the v1p / v2 variants are not in the tree: the run recorded in profiles/r05_attn257_v2_ab.txt was made with tools/ubench/attic/attention_vit257_v2.hip in place of
# stamp_amd/csrc/attention_vit257.hip, which reads AMDS_ATTN257_V2)
# A/B of the stand-alone T = 257 attention kernel, alternating: "old" = build/ab/libamdstamp_old.so (the tree before), "v1p" = this tree's default
# (the round-2 pipeline with packed fma / add), "v2" = AMDS_ATTN257_V2=1 (half-resident scores, maximum subtracted on the matrix pipe):
# parity tests, the kernel alone (tools/attn_only.py 1020 257 16) and the headline leg of bench.py
cd "$(dirname "$0")/.."
python -m pytest tests/test_gpu_kernels.py -x -q -k "attention" 2>&1 | tail -1
AMDS_ATTN257_V2=1 python -m pytest tests/test_gpu_kernels.py -x -q -k "attention" 2>&1 | tail -1
run() { case $1 in old) AMDSTAMP_LIB=$PWD/build/ab/libamdstamp_old.so "${@:2}";; v1p) AMDS_ATTN257_V2=0 "${@:2}";; v2) AMDS_ATTN257_V2=1 "${@:2}";; esac; }
for rep in 1 2 3; do
  for v in old v1p v2; do echo "rep $rep $v: $(run $v python tools/attn_only.py 1020 257 16 2>/dev/null | tail -1)"; done
done
for rep in 1 2 3; do
  for v in old v1p v2; do
    r=$(run $v python bench.py --steps 12 --warmup 3 --no-cpu-baseline --no-secondary --e2e-tiles 0 --slide-tiles 0 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])")
    echo "rep $rep $v: $r"
  done
done
