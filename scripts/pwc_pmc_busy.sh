#!/bin/bash
# Shader clock and matrix-pipe occupancy of the flow engine's largest dispatches: scripts/pwc_pmc_busy.sh [fp16|fp32]
# (GRBM_GUI_ACTIVE / duration = clock; SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x GRBM_GUI_ACTIVE / 8) = busy fraction, as scripts/summarize_prof.py)
REPO="${GRAFT_REPO_ROOT:-/root/repo}"; PR="${1:-fp16}"
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/ppmcb; timeout 600 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES --kernel-trace -d /tmp/ppmcb -o l -- python $REPO/scripts/pwc_prof.py 2 $PR > /tmp/ppmcb.log 2>&1
python - <<'PY'
import sqlite3,glob,collections
f=glob.glob('/tmp/ppmcb/**/*.db',recursive=True)[0]
db=sqlite3.connect(f)
tabs=[r[0] for r in db.execute("select name from sqlite_master where type='table'")]
def t(s): return [x for x in tabs if s in x][0]
pmc=t('pmc_event'); info=t('info_pmc'); kd=t('kernel_dispatch'); ks=t('kernel_symbol')
q=f"select d.id, s.kernel_name, d.grid_size_x, i.name, sum(e.value), d.end-d.start from {pmc} e join {info} i on e.pmc_id=i.id join {kd} d on e.event_id=d.event_id join {ks} s on d.kernel_id=s.id group by d.id, i.name order by d.id"
rows=collections.OrderedDict()
for i,k,g,n,v,dur in db.execute(q):
    r=rows.setdefault(i,{'k':k,'g':g,'dur':dur}); r[n]=v
R=list(rows.values())
preps=[i for i,r in enumerate(R) if 'pwc_prep_kernel' in r['k']]
R=R[preps[-5]:]
print("us, clock MHz, mfma busy frac, grid, kernel   (the 16 longest dispatches of the last stack)")
for r in sorted(R,key=lambda r:-r['dur'])[:16]:
    gui=r.get('GRBM_GUI_ACTIVE',0); 
    print(f"{r['dur']/1e3:8.1f} {gui/(r['dur']/1e3):7.0f} {r.get('SQ_VALU_MFMA_BUSY_CYCLES',0)/(1024*gui/8) if gui else 0:6.3f} {r['g']:9d} {r['k'][:60]}")
PY
