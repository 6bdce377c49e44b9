#!/usr/bin/env python3
"""Per-workgroup cycle budget of the persistent f16f8 LDS-DMA kernel (conv3x3_dma_fs.h; diagnostics build):
python scripts/trace_fs.py n h w cin cout flags res  ->  life, K loops, epilogues, barrier waits per item (median over the workgroups)."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
n, h, w, ci, co, fl, rs = map(int, sys.argv[1:8])
os.environ.setdefault("FISR_HIP_SO", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "build_ab", "libfisr_hip_diag.so"))
from fisr_amd import lib
import numpy as np
L = lib.lib()
if not hasattr(L, "fisr_diag_bench_conv"):
    sys.exit(f"{lib.SO_PATH} is not a FISR_DIAG build")
L.fisr_diag_bench_conv.argtypes = [ctypes.c_int] * 9 + [ctypes.POINTER(ctypes.c_double), ctypes.c_int, ctypes.c_uint, ctypes.c_char_p]
us = ctypes.c_double()
out = "/tmp/fs_trace.bin"
rc = L.fisr_diag_bench_conv(3, n, h, w, ci, co, fl, rs, 3, ctypes.byref(us), int(os.environ.get("BENCH_ZERO", "0")), 0xffff, out.encode())
if rc:
    sys.exit(f"rc {rc} {L.fisr_last_error(None)}")
a = np.fromfile(out, dtype=np.uint64).reshape(-1, 16).astype(np.int64)
a = a[a[:, 4] > 0]
life, ck, cep, items, cw = a[:, 2] - a[:, 0], a[:, 1], a[:, 3], a[:, 4], a[:, 7]
clk = life / np.maximum(a[:, 6] - a[:, 5], 1) * 100.0
med = lambda x: float(np.median(x))
nch = ci // 16
print(f"{n}x{h}x{w} {ci}->{co} f{fl} r{rs}: {us.value:.1f} us  {2.0 * 9 * ci * co * n * h * w / us.value / 1e6:.1f} TF  workgroups {len(a)}  items/wg {med(items):.1f}  clock {med(clk):.0f} MHz")
print(f"  per item: life {med(life / items):.0f}  K loop {med(ck / items):.0f} ({med(ck / items) / nch:.0f} per chunk; 4864 MFMA cycles)  "
      f"epilogue {med(cep / items):.0f}")
print(f"  per chunk: waiting for the copies + barrier {med(cw / items) / nch:.0f}  multiplying {med(a[:, 8] / items) / nch:.0f}  barrier behind it {med(a[:, 9] / items) / nch:.0f}  "
      f"requesting the next chunk {med(a[:, 10] / items) / nch:.0f}")
