set -x
R=$GRAFT_REPO_ROOT; cd $R
( timeout 900 python -m pytest tests -q -m gpu -x -k "transmil or ppeg" 2>&1 | tail -3 )
for L in build/ab/libamdstamp_prev.so stamp_amd/lib/libamdstamp.so build/ab/libamdstamp_prev.so stamp_amd/lib/libamdstamp.so; do
  echo "lib=$L"
  AMDSTAMP_LIB=$R/$L timeout 200 python tools/transmil_train_only.py 64 6 2>&1 | tail -1
  AMDSTAMP_LIB=$R/$L timeout 200 python tools/transmil_only.py 2>&1 | tail -1
done
