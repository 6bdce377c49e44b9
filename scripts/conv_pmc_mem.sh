#!/bin/bash
# HBM-side traffic of the conv micro-benchmark per launch: scripts/conv_pmc_mem.sh "<precisions>" "<shapes>"  (FETCH_SIZE and WRITE_SIZE in separate passes;
# FETCH_SIZE counts 64-byte units on gfx950: x 2 = bytes / 32, see MI355X_MICROARCH.md -- printed as the guide prescribes: KB units x 2)
REPO="${GRAFT_REPO_ROOT:-/root/repo}"
PRECS="${1:-f16f8 f16f8r}"; export CONV_SHAPES="${2:-12,544,992,64,64,0,0}"
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
for k,n,v,c in db.execute(q):
    if 'conv3x3' in k: print(f"{k[:80]:80s} {n:12s} {v/c:.4g} per dispatch ({c} dispatches)")
PY
}
pass "FETCH_SIZE"
pass "WRITE_SIZE"
