#!/usr/bin/env python3
"""Condense rocprofv3 output (rocpd sqlite .db of ROCm 7.2, or CSV) into a short per-kernel
summary: kernel stats from the trace pass, counter totals / per-dispatch averages from PMC passes."""
import glob
import os
import re
import sqlite3
import sys
from collections import defaultdict

root = sys.argv[1]


def short(name):
    name = re.sub(r"\bvoid\s+", "", name).replace("fisr::", "")
    name = re.sub(r"\(.*", "", name)
    return name[:70]


print("# rocprofv3 summary of", os.path.basename(os.path.abspath(root)))
for f in sorted(glob.glob(os.path.join(root, "**", "*.db"), recursive=True)):
    db = sqlite3.connect(f)
    cur = db.cursor()
    rel = os.path.relpath(f, root)
    try:
        rows = list(cur.execute("select name, count(*), sum(duration), avg(duration), min(duration), max(duration) "
                                "from kernels group by name order by sum(duration) desc"))
    except sqlite3.Error as e:
        print(rel, "no kernels view:", e)
        continue
    tot = sum(r[2] for r in rows) or 1
    pmc = list(cur.execute("select name, counter_name, count(*), sum(counter_value) from pmc_events "
                           "group by name, counter_name"))
    if not pmc:
        print(f"\n## kernel trace stats ({rel}); durations from the GPU timestamps of each dispatch")
        print(f"{'kernel':72s} {'calls':>6s} {'total_ms':>10s} {'avg_us':>10s} {'min_us':>9s} {'max_us':>9s} {'pct':>6s}")
        for n, c, s, a, mn, mx in rows:              # every kernel: warp / maxpool rows must not be cut
            print(f"{short(n):72s} {c:6d} {s / 1e6:10.3f} {a / 1e3:10.2f} {mn / 1e3:9.2f} {mx / 1e3:9.2f} {100 * s / tot:6.2f}")
    else:
        print(f"\n## counters ({rel})")
        agg = defaultdict(dict)
        for n, cn, c, s in pmc:
            agg[short(n)][cn] = (c, s)
        dur = {short(r[0]): (r[1], r[2]) for r in rows}
        for k in sorted(agg, key=lambda k: -dur.get(k, (0, 0))[1]):
            print(f"{k}  dispatches {dur.get(k, (0, 0))[0]}  total_ms {dur.get(k, (0, 0))[1] / 1e6:.3f}")
            for cn, (c, s) in sorted(agg[k].items()):
                print(f"    {cn:30s} sum {s:.6g}   per_dispatch {s / max(c, 1):.6g}")

# ---- HBM traffic per launch of each kernel (for bench.py's roofline.traffic) ----
# FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE reports 1/2 of the bytes of wide
# coalesced streaming reads (MI355X_MICROARCH.md, HBM section) -> doubled here.
import json

traffic = {}
for f in sorted(glob.glob(os.path.join(root, "**", "*.db"), recursive=True)):
    cur = sqlite3.connect(f).cursor()
    try:
        rows = list(cur.execute("select name, counter_name, count(*), sum(counter_value) from pmc_events "
                                "where counter_name in ('FETCH_SIZE','WRITE_SIZE') group by name, counter_name"))
    except sqlite3.Error:
        continue
    for n, cn, c, s in rows:
        traffic.setdefault(short(n), {})[cn] = {"dispatches": c, "kib_per_dispatch": s / max(c, 1)}
durations = {}   # average dispatch duration (ns) per kernel, from the kernel-trace pass (no counters attached)
for f in sorted(glob.glob(os.path.join(root, "**", "*.db"), recursive=True)):
    cur = sqlite3.connect(f).cursor()
    try:
        if list(cur.execute("select count(*) from pmc_events"))[0][0]:
            continue
        for n, c, a in cur.execute("select name, count(*), avg(duration) from kernels group by name"):
            durations[short(n)] = (c, a)
    except sqlite3.Error:
        continue
out = {}
for k, d in traffic.items():
    if "FETCH_SIZE" in d and "WRITE_SIZE" in d:
        out[k] = {"fetch_bytes_per_launch_raw": d["FETCH_SIZE"]["kib_per_dispatch"] * 1024,
                  "fetch_bytes_per_launch_corrected_x2": d["FETCH_SIZE"]["kib_per_dispatch"] * 2048,
                  "write_bytes_per_launch": d["WRITE_SIZE"]["kib_per_dispatch"] * 1024,
                  "hbm_bytes_per_launch": d["FETCH_SIZE"]["kib_per_dispatch"] * 2048 + d["WRITE_SIZE"]["kib_per_dispatch"] * 1024,
                  "dispatches": d["FETCH_SIZE"]["dispatches"]}
        if k in durations:
            out[k]["avg_duration_us_trace_pass"] = durations[k][1] / 1e3
            out[k]["hbm_gbps"] = out[k]["hbm_bytes_per_launch"] / durations[k][1]
            out[k]["frac_of_8tbps"] = out[k]["hbm_gbps"] / 8000.0
# ---- share of the matrix pipe busy: SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x GRBM_GUI_ACTIVE / 8 XCDs) per kernel, and the shader
# clock while it ran (GRBM_GUI_ACTIVE of one XCD / duration) -- bench.py's roofline.pmc_mfma_busy_frac
for f in sorted(glob.glob(os.path.join(root, "**", "*.db"), recursive=True)):
    cur = sqlite3.connect(f).cursor()
    try:
        rows = list(cur.execute("select name, counter_name, count(*), sum(counter_value), sum(duration) from pmc_events "
                                "where counter_name in ('SQ_VALU_MFMA_BUSY_CYCLES','GRBM_GUI_ACTIVE','SQ_INSTS_VALU') group by name, counter_name"))
    except sqlite3.Error:
        try:
            rows = [r + (None,) for r in cur.execute("select name, counter_name, count(*), sum(counter_value) from pmc_events "
                                                     "where counter_name in ('SQ_VALU_MFMA_BUSY_CYCLES','GRBM_GUI_ACTIVE','SQ_INSTS_VALU') group by name, counter_name")]
        except sqlite3.Error:
            continue
    agg = defaultdict(dict)
    for n, cn, c, s_, dsum in rows:
        agg[short(n)][cn] = (c, s_, dsum)
    for k, d in agg.items():
        if "SQ_VALU_MFMA_BUSY_CYCLES" in d and "GRBM_GUI_ACTIVE" in d and d["GRBM_GUI_ACTIVE"][1] > 0:
            e = out.setdefault(k, {"dispatches": d["GRBM_GUI_ACTIVE"][0]})
            e["mfma_busy_frac"] = d["SQ_VALU_MFMA_BUSY_CYCLES"][1] / (1024.0 * d["GRBM_GUI_ACTIVE"][1] / 8.0)
            e["gui_active_cycles_per_launch"] = d["GRBM_GUI_ACTIVE"][1] / max(d["GRBM_GUI_ACTIVE"][0], 1)
            if "SQ_INSTS_VALU" in d:      # a wave64 vector instruction holds its SIMD's vector ALU for 4 cycles (packed fp32 included)
                e["valu_issue_frac"] = 4.0 * d["SQ_INSTS_VALU"][1] / (1024.0 * d["GRBM_GUI_ACTIVE"][1] / 8.0)
            # shader clock while the kernel ran: cycles and duration OF THE SAME PASS (r06: cycles of the counter pass over durations
            # of the trace pass gave 3.4 - 5.7 GHz on short kernels).  GRBM_GUI_ACTIVE also counts the dispatch's ramp-up and drain,
            # which a 20-us kernel does not amortise: a clock outside 0.5 - 2.5 GHz cannot be true -- the row's cycle-derived columns
            # are nulled instead of published (tests/test_host.py checks the committed tables).
            dsum = d["GRBM_GUI_ACTIVE"][2]
            clk = e["gui_active_cycles_per_launch"] / (dsum / max(d["GRBM_GUI_ACTIVE"][0], 1) / 1e3) if dsum else None
            e["shader_clock_mhz"] = clk
            if clk is None or not (500.0 <= clk <= 2500.0):
                e["shader_clock_mhz"] = None
                e["cycle_columns_nulled"] = "no same-pass duration" if clk is None else f"derived clock {clk:.0f} MHz outside 500-2500"
                e["mfma_busy_frac"] = None
                e["valu_issue_frac"] = None
# ---- which run this is: the library string (it carries the sha256 of csrc/ + include/fisr.h) and the number of steps, read from the
# bench line of the trace pass -- bench.py drops counter-derived fields whose run is not the running library's, or whose launch
# population (dispatches per step) is not the one it sees
meta = {"source": "scripts/gpu_profile.sh -> scripts/summarize_prof.py"}
for lf in sorted(glob.glob(os.path.join(root, "*.log"))):
    try:
        for line in open(lf, errors="replace"):
            if line.startswith("{") and '"library"' in line:
                d = json.loads(line)
                meta["library"] = d.get("library")
                meta["steps_per_pass"] = int(d.get("steps", 0)) + int(d.get("warmup", 0))
                meta["engines"] = [d.get("config", {}).get("engine", "")[:40]] + sorted((d.get("other_precisions") or {}).keys())
                break
    except (OSError, ValueError):
        continue
    if "library" in meta:
        break
if out:
    out["_meta"] = meta
    with open(os.path.join(root, "pmc_traffic.json"), "w") as fh:
        json.dump(out, fh, indent=1)
    print("\n## pmc_traffic.json written:", ", ".join(out))
