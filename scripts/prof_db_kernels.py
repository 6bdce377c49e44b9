#!/usr/bin/env python3
"""Per-kernel numbers out of a `rocprofv3 --kernel-trace` database (rocpd, *_results.db):
  python scripts/prof_db_kernels.py <dir with *_results.db> <reps> <name substrings...>     (name, grid) groups: launches, average us
  python scripts/prof_db_kernels.py <dir> <reps> --dispatches                               the dispatches of the LAST repetition in
                                                                                            launch order (profiles/r04_flow_*_dispatches.txt)"""
import collections, glob, sqlite3, sys
f = glob.glob(sys.argv[1] + '/*.db')[0]
reps = int(sys.argv[2])
c = sqlite3.connect(f)
rows = list(c.execute("select name, start, end, grid_x, workgroup_x from kernels order by start"))
if "--dispatches" in sys.argv[3:]:
    # a stack starts with the pre-processing of its frames (pwc_prep_kernel x frames): the last such run opens the last repetition
    starts = [i for i, r in enumerate(rows) if "pwc_prep_kernel" in r[0] and (i == 0 or "pwc_prep_kernel" not in rows[i - 1][0])]
    last = rows[starts[-1]:] if starts else rows[-(len(rows) // reps):]
    n = len(last)
    t0 = last[0][1]
    print("dispatches per stack", n, " total ms", round(sum(e - s for _, s, e, _, _ in last) / 1e6, 2))
    for name, s, e, gx, wx in last:
        print("%8.1f us  + %8.1f  grid %8d wg %4d  %s" % ((e - s) / 1e3, (s - t0) / 1e3, gx, wx, name.replace("void fisr::", "")[:110]))
    sys.exit(0)
g = collections.defaultdict(lambda: [0, 0.0])
for n, s, e, gx, _ in rows:
    g[(n[:60], gx)][0] += 1
    g[(n[:60], gx)][1] += (e - s) / 1e3
print('total ms per stack', sum(v[1] for v in g.values()) / reps / 1e3)
for k, v in sorted(g.items(), key=lambda kv: -kv[1][1]):
    if not sys.argv[3:] or any(w in k[0] for w in sys.argv[3:]):
        print('%-62s grid %9d n=%3d avg %.1f us' % (k[0], k[1], v[0], v[1] / v[0]))
