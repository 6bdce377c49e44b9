#!/usr/bin/env python3
"""Per-workgroup cycle accounting of the weight-gradient kernel (diagnostics build: FISR_HIP_SO=build_ab/libfisr_hip_diag.so):
where wave 0 of each workgroup spends its cycles (staging between the barriers / MFMA section / reduction / atomics).
python scripts/wgrad_trace.py [dense|relu]"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from fisr_amd import lib
L = lib.lib()
vp = ctypes.c_void_p
L.fisr_diag_wgrad_trace.argtypes = [vp, ctypes.c_int, vp, ctypes.c_int, ctypes.c_int, vp, ctypes.c_int, vp, vp] + [ctypes.c_int] * 5 + [vp, vp]
mode = sys.argv[1] if len(sys.argv) > 1 else "relu"
for ci, co, r in [(64, 64, 96), (128, 64, 96), (256, 256, 24), (64, 64, 48), (64, 64, 24), (512, 512, 6)]:
    n = 32
    x = torch.randn(n, r, r, ci, device="cuda"); g = torch.randn(n, r, r, co, device="cuda")
    if mode == "relu":
        x = torch.relu(x); g = g * (torch.rand_like(g) > 0.5)
    dw = torch.zeros(3, 3, ci, co, device="cuda"); db = torch.zeros(co, device="cuda")
    tr = torch.zeros(1024 * 8, dtype=torch.int64, device="cuda")
    for _ in range(3):
        rc = L.fisr_diag_wgrad_trace(vp(x.data_ptr()), ci, None, 0, 1, vp(g.data_ptr()), co, vp(dw.data_ptr()), vp(db.data_ptr()), ci, co, n, r, r, None, vp(tr.data_ptr()))
        assert rc == 0
    torch.cuda.synchronize()
    t = tr.cpu().numpy().reshape(-1, 8)
    t = t[t[:, 5] != 0]
    total = (t[:, 5] - t[:, 0]).astype(np.float64)
    real = (t[:, 7] - t[:, 6]).astype(np.float64) / 100.0          # us (100 MHz)
    span = (t[:, 7].max() - t[:, 6].min()) / 100.0
    print(f"ci {ci} co {co} map {r} ({mode}): {len(t)} workgroups, kernel span {span:.1f} us; per workgroup (mean): life {real.mean():.1f} us = "
          f"{total.mean():.0f} cycles ({total.mean() / real.mean() / 1e3:.2f} GHz): staging {100 * (t[:, 1] / total).mean():.1f} %, MFMA section "
          f"{100 * (t[:, 2] / total).mean():.1f} %, reduction {100 * ((t[:, 4] - t[:, 3]) / total).mean():.1f} %, atomics {100 * ((t[:, 5] - t[:, 4]) / total).mean():.1f} %, "
          f"start spread {(t[:, 6].max() - t[:, 6].min()) / 100.0:.1f} us")
