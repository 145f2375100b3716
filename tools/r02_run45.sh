set -x
R=$GRAFT_REPO_ROOT; cd $R
for pad in 0 20000 45000; do echo "PADLDS=$pad"; AMDS_BGEMM_PADLDS=$pad timeout 300 python tools/bgemm_f32_yardstick.py 2>&1 | grep "^Z=" | sed 's/| vendor.*//'; done
