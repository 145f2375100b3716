# round-2 GPU call 49: attention -- previous commit (bpermute exchanges) against DPP / v_permlane32_swap exchanges, one box
set -x
R=$GRAFT_REPO_ROOT; mkdir -p $R/gpurun_out
cd $R
timeout 30 build/xor32_check
for i in 1 2 3; do
  for L in build/ab/libamdstamp_old.so build/ab/libamdstamp_prev.so stamp_amd/lib/libamdstamp.so; do
    echo "lib=$L"
    AMDSTAMP_LIB=$R/$L timeout 120 python tools/attn_only.py 1020 257 16 2>&1 | tail -1
  done
done
