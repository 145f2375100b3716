# round-2 GPU call 38: odd query on the MFMA pipe -- whole-path A/B (alternating) and the new phase timeline
set -x
R=$GRAFT_REPO_ROOT; mkdir -p $R/gpurun_out
cd $R
for i in 1 2 3; do
  for L in stamp_amd/lib/libamdstamp.so build/ab/libamdstamp_old.so; do
    echo "lib=$L"
    AMDSTAMP_LIB=$R/$L timeout 300 python bench.py --no-secondary --no-cpu-baseline --e2e-tiles 0 2>&1 | grep -o '"value": [0-9.]*, "unit": "tiles/s", "n_gpus"'
  done
done
hipcc --offload-arch=gfx950 -O3 -std=c++17 -DA7_TRACE -Iinclude -Istamp_amd/csrc tools/ubench/attn257_trace.hip -o /tmp/a7t && timeout 60 /tmp/a7t
