#!/usr/bin/env python3
"""Training step of the reference's configuration (batch 8, 96x96 LR patches of 5 frames; main.py:74, FISRnet.py:187):
four weight-sharing passes forward + backward + Adam.  python scripts/train_bench.py [batch] [patch] [steps] [serial]
("serial": weight gradients on the main stream like everything else -- for per-kernel profiles whose durations add up)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
import numpy as np, torch
from fisr_amd import train, weights
b = int(sys.argv[1]) if len(sys.argv) > 1 else 8
p = int(sys.argv[2]) if len(sys.argv) > 2 else 96
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 5
r = np.random.default_rng(0)
f32 = np.float32
batch = train.to_device_batch(dict(data15=r.random((b, p, p, 15), dtype=f32), label21=r.random((b, 2 * p, 2 * p, 21), dtype=f32),
                                   flow16=(r.standard_normal((b, p, p, 16)) * 0.02).astype(f32), warp24=r.random((b, p, p, 24), dtype=f32),
                                   flow_ss2=(r.standard_normal((b, p, p, 8)) * 0.04).astype(f32), warp_ss2=r.random((b, p, p, 12), dtype=f32)))
net = train.TrainNet(weights.synthetic_weights(2020))
net.overlap_wgrad = "serial" not in sys.argv[4:]
for _ in range(2):
    net.train_step(batch, 1e-4)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(steps):
    loss, _ = net.train_step(batch, 1e-4)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / steps
flop = 4 * b * 5288328.0 * p * p * 3          # forward + data gradient + weight gradient of every conv, four passes
print(("two streams" if net.overlap_wgrad else "one stream") + f": batch {b} x {p}x{p}: {dt * 1e3:.1f} ms/step, {b / dt:.1f} samples/s, {flop / dt / 1e12:.1f} TFLOP/s (3 x forward FLOPs), loss {loss:.4f}")
