"""Gate of the split-bf16 Winograd kernel (conv3x3_wino8b.h, FISR_PREC_F32WB).

Per-launch time of the three dominant hi-region shapes on every kernel that could run them (fisr_bench_conv: three warm-ups,
`iters` launches back to back, HIP events) and the new kernel's error against the fp64 direct convolution on O(1) data
(N(0,1) inputs, He-scaled weights -- the op tests' distribution), next to conv3x3_dma_fs's on format-exact operands.
Gate: >= 25 % faster than FISR_PREC_F16F8 per launch on all three shapes, max |err| <= 5e-5.

    python scripts/probes/winob_gate.py [--iters 10] [--out gpurun_out/winob_gate.json]
"""
import argparse
import ctypes
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, os.path.join(ROOT, "tests"))

import torch  # noqa: E402

from fisr_amd import lib as flib  # noqa: E402

SHAPES = [  # n, h, w, cin, cout, flags, with_res, label
    (12, 544, 992, 64, 64, 3, 0, "64->64 @12x544x992 relu-on-load"),
    (12, 544, 992, 64, 64, 0, 1, "64->64 @12x544x992 residual"),
    (12, 272, 496, 128, 128, 3, 0, "128->128 @12x272x496 relu-on-load"),
    (12, 544, 992, 64, 256, 7, 0, "64->256 @12x544x992 d2s"),
    (12, 544, 992, 128, 64, 2, 0, "128->64 @12x544x992"),
]
PRECS = [("f16f8", 3), ("f32wb", 10), ("fp32w4", 8), ("fp32w", 4), ("bf16x3", 2), ("fp16", 1)]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=10)
    ap.add_argument("--out", default=None)
    ap.add_argument("--precs", default=",".join(p for p, _ in PRECS))
    a = ap.parse_args()
    L = flib.lib()
    L.fisr_bench_conv.argtypes = [ctypes.c_int] * 9 + [ctypes.POINTER(ctypes.c_double)]
    want = a.precs.split(",")
    rows = []
    for n, h, w, ci, co, fl, wr, label in SHAPES:
        row = {"shape": label, "gflop": 2.0 * 9 * ci * co * n * h * w * 1e-9}
        for name, pid in PRECS:
            if name not in want:
                continue
            us = ctypes.c_double(0)
            rc = L.fisr_bench_conv(pid, n, h, w, ci, co, fl, wr, a.iters, ctypes.byref(us))
            row[name + "_us"] = round(us.value, 1) if rc == 0 else None
        if row.get("f16f8_us") and row.get("f32wb_us"):
            row["f32wb_vs_f16f8"] = round(row["f16f8_us"] / row["f32wb_us"], 3)
        print(json.dumps(row), flush=True)
        rows.append(row)

    # accuracy on O(1) data against the fp64 direct convolution
    import test_gpu_parity as T  # noqa: E402  (helpers only: hip_conv / ref_conv)
    T.PREC_ID["f32wb"] = 10
    acc = []
    for (n, h, w, ci, co, fl, use_res) in [(1, 64, 96, 64, 64, 3, False), (1, 64, 96, 64, 64, 0, True), (1, 48, 64, 128, 128, 3, False),
                                           (1, 32, 64, 64, 256, 7, False), (1, 32, 64, 128, 64, 2, False), (1, 16, 32, 512, 256, 2, False)]:
        rng = np.random.default_rng(ci * 1000 + co + fl)
        x = rng.standard_normal((n, h, w, ci)).astype(np.float32)
        wt = (rng.standard_normal((3, 3, ci, co)) * np.sqrt(2.0 / (9 * ci))).astype(np.float32)
        b = rng.standard_normal(co).astype(np.float32)
        res = rng.standard_normal((n, h, w, co)).astype(np.float32) if use_res else None
        exp = T.ref_conv(x, wt, b, None, res, fl)
        r = {"shape": f"{ci}->{co} @{n}x{h}x{w} flags {fl} res {int(use_res)}"}
        for prec in ("f32wb", "fp32w", "f16f8"):
            xin = T.splitfmt.from_fsplit(T.splitfmt.to_fsplit(x)) if prec == "f16f8" else x      # format-exact operands for f16f8
            e = T.ref_conv(xin, wt, b, None, res, fl) if prec == "f16f8" else exp
            got = T.hip_conv(xin if prec == "f16f8" else x, wt, b, None, res, fl, prec=prec)
            err = np.abs(got.astype(np.float64) - e)
            r[prec] = {"max": float(err.max()), "rms": float(np.sqrt((err ** 2).mean())), "nan": int(np.isnan(got).sum())}
        print(json.dumps(r), flush=True)
        acc.append(r)
    out = {"library": L.fisr_version().decode(), "device": torch.cuda.get_device_name(0), "iters": a.iters, "timing": rows, "accuracy": acc}
    if a.out:
        os.makedirs(os.path.dirname(a.out), exist_ok=True)
        json.dump(out, open(a.out, "w"), indent=1)


if __name__ == "__main__":
    T = None
    torch.empty(1, device="cuda")
    flib.lib().fisr_version.restype = ctypes.c_char_p
    # the test helpers register PREC_ID["fp32w"] etc.; f32wb tensors are plain fp32
    import test_gpu_parity as _T
    _orig_to, _orig_from, _orig_empty = _T.to_dev, _T.from_dev, _T.empty_dev
    _T.to_dev = lambda x, prec: _orig_to(x, "fp32" if prec == "f32wb" else prec)
    _T.from_dev = lambda t, prec, shape: _orig_from(t, "fp32" if prec == "f32wb" else prec, shape)
    _T.empty_dev = lambda shape, prec: _orig_empty(shape, "fp32" if prec == "f32wb" else prec)
    main()
