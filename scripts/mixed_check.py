#!/usr/bin/env python3
"""Mixed-precision engine: accuracy against the fp32 engine on the three weight sets, and speed on the 1080p workload
(diagnostics; the parity tests proper are in tests/test_gpu_weightsets.py)."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from fisr_amd import weights
from fisr_amd.fisrnet import FISRnet
from tests_support import make_full_size_input

def psnr_shift(rms, db): return 4.343 * (rms / 10 ** (-db / 20)) ** 2

for ws in ("default", "survey_spec", "harsh"):
    W = weights.synthetic_weights(2020) if ws == "default" else weights.WEIGHT_SETS[ws]()
    x = torch.from_numpy(make_full_size_input(12, 96, 160, 2)).cuda()
    ref = None
    for prec in ("fp32", "fp16", "mixed", "bf16x3"):
        net = FISRnet(device="cuda:0", precision=prec); net.set_weights(W)
        out = net.model(x)[2].float().cpu().numpy(); net.close()
        if ref is None: ref = out; continue
        e = np.clip(out, 0, 1) - np.clip(ref, 0, 1)
        sr = float(np.sqrt((e[..., 3:6] ** 2).mean())); fi = float(np.sqrt((e[..., 0:3] ** 2).mean()))
        print(f"{ws:12s} {prec:7s} SR rms {sr:.2e} (dPSNR~{psnr_shift(sr, 48.07):.4f} dB)  FI rms {fi:.2e} (dPSNR~{psnr_shift(fi, 37.86):.4f} dB)", flush=True)

W = weights.synthetic_weights(2020)
x = torch.rand(12, 544, 992, 29, device="cuda")
for prec in ("bf16x3", "f16f8", "fp16", "mixed"):
    net = FISRnet(device="cuda:0", precision=prec); net.set_weights(W)
    for _ in range(2): net.model(x)
    torch.cuda.synchronize(); t = time.time()
    for _ in range(4): net.model(x)
    torch.cuda.synchronize(); ms = (time.time() - t) / 4 * 1e3
    print(f"{prec:7s} {ms:7.2f} ms per 12-tile forward (3 windows x 4 tiles = 7 unique frames) -> {7 / (ms / 1e3):.1f} frames/s", flush=True)
    net.close()
