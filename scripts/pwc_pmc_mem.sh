#!/bin/bash
# HBM bytes of the flow engine's kernels, per dispatch of one stack: scripts/pwc_pmc_mem.sh [fp16|fp32]  (FETCH_SIZE KiB x 2 on gfx950, WRITE_SIZE KiB)
REPO="${GRAFT_REPO_ROOT:-/root/repo}"; PR="${1:-fp16}"
cd /tmp && export TMPDIR=/tmp
for pm in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/ppmc_$pm; timeout 600 rocprofv3 --pmc $pm --kernel-trace -d /tmp/ppmc_$pm -o l -- python $REPO/scripts/pwc_prof.py 2 $PR > /tmp/ppmc.log 2>&1
done
python - <<'PY'
import sqlite3,glob,collections
def load(pm):
    f=glob.glob(f'/tmp/ppmc_{pm}/**/*.db',recursive=True)[0]
    db=sqlite3.connect(f)
    tabs=[r[0] for r in db.execute("select name from sqlite_master where type='table'")]
    def t(s): return [x for x in tabs if s in x][0]
    pmc=t('pmc_event'); kd=t('kernel_dispatch'); ks=t('kernel_symbol')
    q=f"select d.id, s.kernel_name, d.grid_size_x, sum(e.value), d.end-d.start from {pmc} e join {kd} d on e.event_id=d.event_id join {ks} s on d.kernel_id=s.id group by d.id order by d.id"
    return list(db.execute(q))
F=load('FETCH_SIZE'); W=load('WRITE_SIZE')
preps=[i for i,r in enumerate(F) if 'pwc_prep_kernel' in r[1]]
n=preps[-5]    # the last stack starts with its five frame preparations
rows=[]
for (i,k,g,fv,dur),(_,_,_,wv,_) in zip(F[n:],W[n:]):
    rows.append((dur/1e3, fv*2048/1e9, wv*1024/1e9, g, k[:48]))
tot=collections.defaultdict(lambda:[0,0,0,0])
for d,f,w,g,k in rows:
    t=tot[k]; t[0]+=d; t[1]+=f; t[2]+=w; t[3]+=1
print(f"last stack: {len(rows)} dispatches, {sum(r[0] for r in rows)/1e3:.2f} ms (PMC pass), fetch {sum(r[1] for r in rows):.1f} GB, write {sum(r[2] for r in rows):.1f} GB")
print("per kernel: launches, us (PMC pass), fetch GB, write GB")
for k,t in sorted(tot.items(), key=lambda kv:-kv[1][0])[:10]: print(f"{t[3]:4d} {t[0]:10.1f} {t[1]:8.3f} {t[2]:8.3f}  {k}")
print("largest conv3x3_dma dispatches: us, fetch GB, write GB, grid")
for d,f,w,g,k in sorted([r for r in rows if 'dma_f16' in r[4]], reverse=True)[:14]: print(f"{d:9.1f} {f:8.3f} {w:8.3f} {g:9d} {k}")
PY
