#!/usr/bin/env python3
"""Weight-gradient kernel (train_kernels.h) on the layer shapes of one training step (batch 8 x four passes = 32 images):
python scripts/wgrad_bench.py [reps].  FISR_HIP_SO selects a diagnostics build (lib.build(diag=True, defines=["FISR_GABL=.."]))."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from fisr_amd import lib
L = lib.lib()
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
shapes = [(64, 64, 96), (128, 64, 96), (128, 128, 48), (256, 256, 24), (512, 512, 12), (64, 64, 48), (256, 256, 12), (512, 512, 6),
          (64, 64, 24), (128, 128, 12), (256, 256, 6), (512, 512, 3)]
vp = ctypes.c_void_p
tot = 0.0
for ci, co, r in shapes:
    n = 32
    x = torch.randn(n, r, r, ci, device="cuda"); g = torch.randn(n, r, r, co, device="cuda")
    dw = torch.zeros(3, 3, ci, co, device="cuda"); db = torch.zeros(co, device="cuda")
    st = vp(torch.cuda.current_stream().cuda_stream)
    def run():
        rc = L.fisr_train_wgrad(vp(x.data_ptr()), ci, None, 0, 1, vp(g.data_ptr()), co, vp(dw.data_ptr()), vp(db.data_ptr()), ci, co, n, r, r, st)
        assert rc == 0
    for _ in range(3): run()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): run()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / reps * 1e3
    fl = 2.0 * 9 * ci * co * r * r * n
    tot += us
    print(f"ci {ci:4d} co {co:4d} map {r:3d}x{r:<3d} x{n}: {us:8.1f} us  {fl / us / 1e6:7.1f} TFLOP/s")
print(f"sum {tot:.0f} us  ({L.fisr_version().decode()})")
