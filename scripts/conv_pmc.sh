#!/bin/bash
# PMC passes over the conv micro-benchmark: scripts/conv_pmc.sh "<precisions>" "<shapes n,h,w,ci,co,flags,res;...>"  -> per-kernel counter sums
REPO="${GRAFT_REPO_ROOT:-/root/repo}"
PRECS="${1:-f16f8 f16f8r}"; export CONV_SHAPES="${2:-12,272,496,128,128,3,1}"
cd /tmp && export TMPDIR=/tmp
pass() {
  rm -rf /tmp/cpmc; timeout 600 rocprofv3 --pmc $1 --kernel-trace -d /tmp/cpmc -o l -- python $REPO/scripts/conv_bench.py $PRECS > /tmp/cpmc.log 2>&1
  python - <<'PY'
import sqlite3,glob,collections
f=glob.glob('/tmp/cpmc/**/*.db',recursive=True)[0]
db=sqlite3.connect(f)
tabs=[r[0] for r in db.execute("select name from sqlite_master where type='table'")]
def t(s): return [x for x in tabs if s in x][0]
pmc=t('pmc_event'); info=t('info_pmc'); kd=t('kernel_dispatch'); ks=t('kernel_symbol')
q=f"select s.kernel_name, i.name, sum(e.value), count(distinct d.id) from {pmc} e join {info} i on e.pmc_id=i.id join {kd} d on e.event_id=d.event_id join {ks} s on d.kernel_id=s.id group by 1,2"
agg=collections.defaultdict(dict); nd={}
for k,n,v,c in db.execute(q): agg[k][n]=v; nd[k]=c
for k,v in agg.items():
    if 'conv3x3' in k:
        print(k[:90], 'dispatches', nd[k])
        for n,x in sorted(v.items()): print(f"    {n:32s} {x/nd[k]:.4g} per dispatch")
PY
}
pass "SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE SQ_INSTS_MFMA"
pass "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_LDS"
