#!/usr/bin/env python3
"""Record per-workgroup timestamps of one conv launch (diagnostics): python scripts/trace_conv.py out.bin n h w cin cout flags res [prec]"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
out, n, h, w, ci, co, fl, rs = sys.argv[1], *map(int, sys.argv[2:9])
prec = {"fp32": 0, "fp16": 1, "bf16x3": 2, "f16f8": 3, "fp32w": 4, "fp32w4": 8}[sys.argv[9] if len(sys.argv) > 9 else "bf16x3"]
# needs a diagnostics build (-DFISR_DIAG: the shipped library has no trace hook): FISR_HIP_SO=build_ab/libfisr_hip_diag.so,
# made by `python -c "from fisr_amd import lib; lib.build(diag=True)"` (or any A/B build of scripts/gpu_ablate.sh)
os.environ.setdefault("FISR_HIP_SO", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "build_ab", "libfisr_hip_diag.so"))
from fisr_amd import lib
L = lib.lib()
if not hasattr(L, "fisr_diag_bench_conv"):
    sys.exit(f"{lib.SO_PATH} is not a FISR_DIAG build")
L.fisr_diag_bench_conv.argtypes = [ctypes.c_int] * 9 + [ctypes.POINTER(ctypes.c_double), ctypes.c_int, ctypes.c_uint, ctypes.c_char_p]
us = ctypes.c_double()
rc = L.fisr_diag_bench_conv(prec, n, h, w, ci, co, fl, rs, 3, ctypes.byref(us), int(os.environ.get("BENCH_ZERO", "0")),
                            int(os.environ.get("BENCH_LOMASK", "ffff"), 16), out.encode())
print("rc", rc, "us", us.value, L.fisr_last_error(None) if rc else "")

import numpy as np
a = np.fromfile(out, dtype=np.uint64).reshape(-1, 8).astype(np.int64)
nwg = int((a[:, 7] > 0).sum())
if prec == 4 and len(a) >= 2 * nwg and nwg > 0 and (a[nwg:2 * nwg, 3] > a[nwg:2 * nwg, 0]).all():
    x = a[nwg:2 * nwg]
    print("output stage: combine in LDS %.0f  read back %.0f  relu + stores %.0f" % (
        np.median(x[:, 1] - x[:, 0]), np.median(x[:, 2] - x[:, 1]), np.median(x[:, 3] - x[:, 2])))
    a = a[:nwg]
a = a[a[:, 2] > a[:, 0]]
t0, tm, te, tf, e1, e2, e3 = a[:, 0], a[:, 1], a[:, 2], a[:, 4], a[:, 5], a[:, 6], a[:, 7]
med = lambda x: float(np.median(x))
clk = (te - t0) / np.maximum(e2 - e1, 1) * 100.0   # s_memtime ticks per 100 MHz s_memrealtime tick -> MHz
print(f"shader clock while the kernel runs: median {med(clk):.0f} MHz (10%..90%: {np.percentile(clk,10):.0f}..{np.percentile(clk,90):.0f})")
if (a[:, 3] > tm).all():   # persistent kernel: the exchange barrier of the traced item
    print(f"epilogue up to the exchange barrier {med(a[:, 3] - tm):.0f}  output stage {med(te - a[:, 3]):.0f}  first iteration {med(tf - t0):.0f}  items/wg {med(e3):.0f}")
print(f"blocks {len(a)}  life {med(te - t0):.0f}  prologue {med(tf - t0):.0f}  main(after prologue) {med(tm - tf):.0f}  epilogue {med(te - tm):.0f}")
