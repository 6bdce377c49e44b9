#!/usr/bin/env python3
"""Per-shape micro-benchmark of the conv kernel: python scripts/conv_bench.py [precision ...]
Uses $FISR_HIP_SO if set (A/B of kernel builds)."""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fisr_amd import lib

SHAPES = [  # n, h, w, cin, cout, flags, res
    (12, 544, 992, 64, 64, 3, 1),
    (12, 544, 992, 64, 64, 0, 0),
    (12, 272, 496, 128, 128, 3, 1),
    (12, 136, 248, 256, 256, 3, 1),
    (12, 68, 124, 512, 512, 3, 1),
    (12, 544, 992, 64, 256, 7, 0),
    (12, 136, 248, 512, 256, 2, 0),
    (12, 136, 248, 64, 64, 3, 1),
    (1, 544, 992, 64, 64, 3, 1),
]
if os.environ.get("CONV_SHAPES"):   # "n,h,w,cin,cout,flags,res;..."
    SHAPES = [tuple(int(v) for v in t.split(",")) for t in os.environ["CONV_SHAPES"].split(";")]
PID = {"fp32": 0, "fp16": 1, "bf16x3": 2, "f16f8": 3, "fp32w": 4, "fp16r": 6, "fp32w4": 8, "f16f8r": 9}
L = lib.lib()
for prec in (sys.argv[1:] or ["bf16x3"]):
    for (n, h, w, ci, co, fl, rs) in SHAPES:
        us = ctypes.c_double()
        iters = 5 if prec.startswith("fp32") else 10
        rc = L.fisr_bench_conv(PID[prec], n, h, w, ci, co, fl, rs, iters, ctypes.byref(us))
        if rc:
            print(prec, (n, h, w, ci, co), "ERR", L.fisr_last_error(None))
            continue
        fl_ = 2.0 * 9 * ci * co * n * h * w
        ab = 2 if prec.startswith("fp16") else 4
        by = n * h * w * (ci * 1.0 + co * (2 if rs else 1)) * ab
        print(f"{os.environ.get('TAG', '')} {prec:7s} {n:2d}x{h}x{w} {ci:3d}->{co:3d} f{fl} r{rs}: {us.value:9.1f} us  {fl_ / us.value / 1e6:7.1f} TF  "
              f"alg {by / us.value / 1e3:6.0f} GB/s")
