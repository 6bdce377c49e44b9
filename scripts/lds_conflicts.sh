#!/bin/bash
# LDS bank-conflict share of the conv kernels (PMC pass on the micro-benchmark): SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE
REPO="${GRAFT_REPO_ROOT:-/root/repo}"
cd /tmp && export TMPDIR=/tmp
for so in ${SOS:-libfisr_hip.so}; do
  rm -rf /tmp/ldsc; 
  FISR_HIP_SO=$REPO/fisr_amd/$so timeout 600 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace -d /tmp/ldsc -o l -- python $REPO/scripts/conv_bench.py ${PRECS:-bf16x3 f16f8 fp32 fp32w} > /tmp/ldsc.log 2>&1
  echo "== $so"
  python - <<'PY'
import sqlite3,glob,collections
f=glob.glob('/tmp/ldsc/**/*.db',recursive=True)[0]
db=sqlite3.connect(f)
tabs=[r[0] for r in db.execute("select name from sqlite_master where type='table'")]
def t(s): return [x for x in tabs if s in x][0]
pmc=t('pmc_event'); info=t('info_pmc'); kd=t('kernel_dispatch'); ks=t('kernel_symbol')
cols=[r[1] for r in db.execute(f"pragma table_info({pmc})")]
q=f"select s.kernel_name, i.name, sum(e.value) from {pmc} e join {info} i on e.pmc_id=i.id join {kd} d on e.event_id=d.event_id join {ks} s on d.kernel_id=s.id group by 1,2"
agg=collections.defaultdict(dict)
for k,n,v in db.execute(q): agg[k][n]=v
for k,v in agg.items():
    if 'conv3x3' in k and v.get('SQ_LDS_IDX_ACTIVE'):
        print(f"{k[:70]:70s} conflict/active = {v.get('SQ_LDS_BANK_CONFLICT',0)/v['SQ_LDS_IDX_ACTIVE']:.3f}")
PY
done
