#!/bin/bash
# The fp32 heads under rocprofv3 PMC passes, shipped kernel (FISR_HEAD_STRIP=0) against the strip-walking one (=1; diagnostics build):
# HBM bytes per launch (FETCH_SIZE x 2 + WRITE_SIZE, KB units as MI355X_MICROARCH.md prescribes) and the vector-ALU share.
REPO="${GRAFT_REPO_ROOT:-/root/repo}"
export FISR_HIP_SO=$REPO/build_ab/libfisr_hip_diag.so
cd /tmp && export TMPDIR=/tmp
for v in 0 1; do
 for pm in "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY"; do
  rm -rf /tmp/hpmc; FISR_HEAD_STRIP=$v timeout 600 rocprofv3 --pmc $pm --kernel-trace -d /tmp/hpmc -o l -- python $REPO/scripts/head_bench.py > /tmp/hpmc.log 2>&1
  V=$v python - <<'PY'
import sqlite3,glob,collections,os
f=glob.glob('/tmp/hpmc/**/*.db',recursive=True)[0]
db=sqlite3.connect(f)
tabs=[r[0] for r in db.execute("select name from sqlite_master where type='table'")]
def t(s): return [x for x in tabs if s in x][0]
pmc=t('pmc_event'); info=t('info_pmc'); kd=t('kernel_dispatch'); ks=t('kernel_symbol')
q=f"select s.kernel_name, i.name, sum(e.value), count(distinct d.id) from {pmc} e join {info} i on e.pmc_id=i.id join {kd} d on e.event_id=d.event_id join {ks} s on d.kernel_id=s.id where s.kernel_name like '%head_conv%' group by 1,2"
for k,n,v,c in db.execute(q): print(f"FISR_HEAD_STRIP={os.environ['V']} {k[:60]:60s} {n:22s} {v/c:.5g} per launch ({c} launches, all map sizes)")
PY
 done
done
