# LDS bank-conflict scan over the secondary paths: which kernels spend their LDS time in conflicts?
set -x
cd /tmp && export TMPDIR=/tmp
R=/root/repo
i=0
for cmd in "tools/train_only.py" "tools/swin_only.py" "tools/transmil_train_only.py 64 2" "tools/attn_only.py 510 261 16 80" "tools/transmil_only.py"; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE -d /tmp/ls$i -o ls -- python $R/$cmd > /tmp/ls$i.log 2>&1 < /dev/null
  DB=$(find /tmp/ls$i -name "*.db" | head -1)
  echo "== $cmd =="
  [ -n "$DB" ] && timeout 60 python $R/tools/pmc_dump.py "$DB" | python -c "
import sys
name=None; d={}
def flush():
    if name and d.get('SQ_LDS_IDX_ACTIVE',0)>0:
        print(f\"{name[:90]:90s} lds_active {d['SQ_LDS_IDX_ACTIVE']/1e6:9.2f}M conflict {d.get('SQ_LDS_BANK_CONFLICT',0)/1e6:9.2f}M ({100*d.get('SQ_LDS_BANK_CONFLICT',0)/d['SQ_LDS_IDX_ACTIVE']:.0f}%) lds/gui {d['SQ_LDS_IDX_ACTIVE']/256/max(1,d.get('GRBM_GUI_ACTIVE',1)/8):.2f}\")
for l in sys.stdin:
    if not l.startswith('    '):
        flush(); name=l.strip(); d={}
    else:
        k,v=l.split(); d[k]=float(v)
flush()
" || tail -5 /tmp/ls$i.log
done > $R/gpurun_out/lds_scan.txt 2>&1
cat $R/gpurun_out/lds_scan.txt
