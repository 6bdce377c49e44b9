#!/usr/bin/env python3
"""One-off campaign: a persistent Winograd kernel -- F(2x2), FISR_PREC_F32W (default), or F(4x4), FISR_PREC_F32W4 (third argument
"f4") -- against the direct exact-fp32 kernel (fp32d) on random LARGER shapes -- many work items per workgroup, ragged H / W,
concat, residual (also in place), relu, depth_to_space and, for F(4x4), the fused x2 bilinear.  python scripts/wino_campaign.py [cases] [seed] [f4]"""
import ctypes, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from fisr_amd import lib as flib
L = flib.lib()
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 100
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
fp = lambda a: a.ctypes.data_as(ctypes.POINTER(ctypes.c_float))
F4 = len(sys.argv) > 3 and sys.argv[3] == "f4"
PREC, TOL = (flib.PREC_F32W4, 4e-4) if F4 else (flib.PREC_F32W, 5e-5)      # (F(4,3)'s transforms: ~10x the rounding error per conv)
worst = 0.0
for case in range(cases):
    n = int(rng.integers(1, 5))
    h, w = int(rng.integers(8, 260)), int(rng.integers(8, 400))
    c0 = 16 * int(rng.integers(2, 9))
    c1 = 16 * int(rng.integers(1, 5)) if rng.random() < 0.3 else 0
    cout = 64 * int(rng.integers(1, 5))
    flags = int(rng.integers(0, 4)) | (4 if rng.random() < 0.25 and cout in (64, 128, 256) else 0)
    use_res = (not flags & 4) and rng.random() < 0.5
    inplace = use_res and rng.random() < 0.5
    # F(4x4) only: the fused x2 bilinear (FISR_CONV_UP2_IN) against upsample2 + the direct kernel
    ups = F4 and rng.random() < 0.3
    if ups:
        h, w, c1, use_res, inplace = h + (h & 1), w + (w & 1), 0, False, False
        flags &= 2
    x0 = torch.randn(n, h, w, c0, device="cuda")
    x1 = torch.randn(n, h, w, c1, device="cuda") if c1 else None
    wt = (rng.standard_normal((3, 3, c0 + c1, cout)) * np.sqrt(2.0 / (9 * (c0 + c1)))).astype(np.float32)
    b = rng.standard_normal(cout).astype(np.float32)
    res = torch.randn(n, h, w, cout, device="cuda") if use_res else None
    oshape = (n, 2 * h, 2 * w, cout // 4) if flags & 4 else (n, h, w, cout)
    outs = []
    if ups:
        xs = torch.randn(n, h // 2, w // 2, c0, device="cuda")
        rc = L.fisr_op_upsample2(ctypes.c_void_p(xs.data_ptr()), ctypes.c_void_p(x0.data_ptr()), n, h // 2, w // 2, c0, 0, None)
        assert rc == 0, (rc, L.fisr_last_error(None))
        torch.cuda.synchronize()
    for prec in (PREC, flib.PREC_F32):
        r = res.clone() if use_res else None
        out = r if inplace else torch.empty(oshape, device="cuda")
        fused = ups and prec == PREC
        rc = L.fisr_op_conv3x3(ctypes.c_void_p((xs if fused else x0).data_ptr()), c0, ctypes.c_void_p(x1.data_ptr() if c1 else 0), c1, fp(wt), fp(b), cout,
                               ctypes.c_void_p(r.data_ptr() if use_res else 0), ctypes.c_void_p(out.data_ptr()), n, h, w,
                               flags | (flib.CONV_UP2_IN if fused else 0), prec, 0, None)
        assert rc == 0, (rc, L.fisr_last_error(None))
        torch.cuda.synchronize()
        outs.append(out)
    err = float((outs[0] - outs[1]).abs().max())
    worst = max(worst, err)
    assert err < TOL and not torch.isnan(outs[0]).any(), f"case {case}: n{n} {h}x{w} {c0}+{c1}->{cout} flags {flags} res {use_res} inplace {inplace} ups {ups}: {err}"
print(f"{cases} cases ok, worst |winograd - direct| = {worst:.2e}")
