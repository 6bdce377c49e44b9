#!/bin/bash
# A/B of HBM traffic (FETCH_SIZE / WRITE_SIZE PMC passes) and speed between two library builds.
REPO="${GRAFT_REPO_ROOT:-/root/repo}"; cd /tmp; export TMPDIR=/tmp
for tag in ${AB_TAGS:-g0 g1}; do
  export FISR_HIP_SO=$REPO/fisr_amd/libfisr_hip_$tag.so
  CMD="python $REPO/bench.py --steps 1 --warmup 1 --precision ${PREC:-bf16x3} --no-cpu-baseline --no-roofline --no-fp32-ref"
  for c in FETCH_SIZE WRITE_SIZE; do
    rm -rf /tmp/pm_$tag_$c; timeout 600 rocprofv3 --pmc $c --kernel-trace -d /tmp/pm_${tag}_$c -o p -- $CMD > /dev/null 2>&1
    python - <<PY
import sqlite3, glob
f=glob.glob("/tmp/pm_${tag}_$c/**/*.db", recursive=True)[0]
cur=sqlite3.connect(f).cursor()
rows=list(cur.execute("select name, count(*), sum(counter_value) from pmc_events where counter_name='$c' group by name order by sum(counter_value) desc"))
tot=sum(r[2] for r in rows)
print("$tag $c total %.1f GB(KiB-units); top:" % (tot*1024/1e9), [(r[0][:40], r[1], round(r[2]*1024/1e9,1)) for r in rows[:2]])
PY
  done
  python $REPO/bench.py --steps 3 --warmup 1 --precision ${PREC:-bf16x3} --no-cpu-baseline --no-fp32-ref 2>/dev/null | python -c "import sys,json; l=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$tag fps', l['value'], 'conv TF', l['roofline']['achieved'])"
done
