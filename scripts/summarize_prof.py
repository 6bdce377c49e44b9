#!/usr/bin/env python3
"""Condense rocprofv3 CSV output (kernel stats + PMC passes) into a short per-kernel summary."""
import csv
import glob
import os
import sys
from collections import defaultdict

root = sys.argv[1]


def short(name):
    name = name.replace("void fisr::", "").replace("fisr::", "")
    return name[:80]


def find(pattern):
    return sorted(glob.glob(os.path.join(root, "**", pattern), recursive=True))


print("# rocprofv3 summary of", os.path.basename(root))
for f in find("*kernel_stats.csv"):
    print("\n## kernel stats (", os.path.relpath(f, root), ")")
    with open(f) as fh:
        rows = list(csv.DictReader(fh))
    for r in rows[:14]:
        print(f"{short(r['Name']):82s} calls {r['Calls']:>6s} total_ms {float(r['TotalDurationNs'])/1e6:10.3f} "
              f"avg_us {float(r['AverageNs'])/1e3:10.2f} pct {r['Percentage']:>6s}")

for f in find("*counter_collection.csv"):
    print("\n## counters (", os.path.relpath(f, root), ")")
    agg = defaultdict(lambda: defaultdict(float))
    cnt = defaultdict(set)
    with open(f) as fh:
        for r in csv.DictReader(fh):
            k = short(r["Kernel_Name"])
            agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
            cnt[k].add(r["Dispatch_Id"])
    for k, d in sorted(agg.items(), key=lambda kv: -sum(kv[1].values()))[:8]:
        n = len(cnt[k])
        print(f"{k:82s} dispatches {n}")
        for c, v in sorted(d.items()):
            print(f"    {c:32s} total {v:.6g}  per_dispatch {v / max(n, 1):.6g}")
