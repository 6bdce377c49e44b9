#!/bin/bash
# Per-kernel shader clock and MFMA-busy share INSIDE the network: scripts/net_pmc_clock.sh <precision>
# (one PMC pass over a short bench run: GRBM_GUI_ACTIVE is summed over the 8 XCDs; busy = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x cycles))
REPO="${GRAFT_REPO_ROOT:-/root/repo}"; PR="${1:-mixed}"
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/npmc; timeout 900 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES --kernel-trace -d /tmp/npmc -o l -- python $REPO/bench.py --precision $PR --others "" --no-cpu-baseline --no-flow --no-train --no-parity --no-roofline --steps 3 --warmup 1 > /tmp/npmc.log 2>&1
python - <<'PY'
import sqlite3,glob,collections
f=glob.glob('/tmp/npmc/**/*.db',recursive=True)[0]
db=sqlite3.connect(f)
tabs=[r[0] for r in db.execute("select name from sqlite_master where type='table'")]
def t(s): return [x for x in tabs if s in x][0]
pmc=t('pmc_event'); info=t('info_pmc'); kd=t('kernel_dispatch'); ks=t('kernel_symbol')
q=f"select s.kernel_name, i.name, sum(e.value), count(distinct d.id), sum(d.end-d.start) from {pmc} e join {info} i on e.pmc_id=i.id join {kd} d on e.event_id=d.event_id join {ks} s on d.kernel_id=s.id group by 1,2"
agg=collections.defaultdict(dict)
for k,n,v,c,dur in db.execute(q): agg[k][n]=(v,c,dur)
rows=[]
for k,v in agg.items():
    if 'GRBM_GUI_ACTIVE' in v and 'SQ_VALU_MFMA_BUSY_CYCLES' in v:
        g,c,dur=v['GRBM_GUI_ACTIVE']; m=v['SQ_VALU_MFMA_BUSY_CYCLES'][0]
        cyc=g/8.0
        rows.append((dur/1e6, c, cyc/(dur/1e3) if dur else 0, m/(1024.0*cyc) if cyc else 0, k))
rows.sort(reverse=True)
for ms,c,mhz,busy,k in rows[:14]: print(f"{ms:8.2f} ms total  {c:4d} launches  clock {mhz:6.0f} MHz  MFMA busy {busy:5.3f}  {k[:100]}")
PY
