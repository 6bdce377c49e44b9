#!/usr/bin/env python3
"""Per-call timing of the conv / weight-gradient calls of one training step (HIP events around every library call, a
synchronize between calls), grouped by shape: where the step's conv time goes.  python scripts/train_calls.py [batch] [patch]"""
import collections, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
import numpy as np, torch
from fisr_amd import train, weights
b = int(sys.argv[1]) if len(sys.argv) > 1 else 8
p = int(sys.argv[2]) if len(sys.argv) > 2 else 96
r = np.random.default_rng(0); f32 = np.float32
batch = train.to_device_batch(dict(data15=r.random((b, p, p, 15), dtype=f32), label21=r.random((b, 2 * p, 2 * p, 21), dtype=f32),
                                   flow16=(r.standard_normal((b, p, p, 16)) * 0.02).astype(f32), warp24=r.random((b, p, p, 24), dtype=f32),
                                   flow_ss2=(r.standard_normal((b, p, p, 8)) * 0.04).astype(f32), warp_ss2=r.random((b, p, p, 12), dtype=f32)))
net = train.TrainNet(weights.synthetic_weights(2020))
for _ in range(2):
    net.train_step(batch, 1e-4)
torch.cuda.synchronize()
log = []

class Proxy:
    def __init__(self, L): self._L = L
    def __getattr__(self, k):
        f = getattr(self._L, k)
        if k not in ("fisr_train_conv3x3", "fisr_train_wgrad"):
            return f
        def timed(*a):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize(); e0.record(); rc = f(*a); e1.record(); torch.cuda.synchronize()
            if k == "fisr_train_wgrad":      # (x0,c0,x1,c1,relu,g,cg,dw,db,ci,co,n,h,w,st)
                key = ("wgrad", a[9], a[10], a[12])
                fl = 2.0 * 9 * a[9] * a[10] * a[11] * a[12] * a[13]
            else:                            # (in0,c0,in1,c1,pk,b,cout,res,out,n,h,w,...)
                key = ("conv", a[1] + a[3], a[6], a[10])
                fl = 2.0 * 9 * (a[1] + a[3]) * a[6] * a[9] * a[10] * a[11]
            log.append((key, e0.elapsed_time(e1) * 1e3, fl))
            return rc
        return timed
net.L = Proxy(net.L)
net.train_step(batch, 1e-4)
agg = collections.OrderedDict()
for key, us, fl in log:
    v = agg.setdefault(key, [0, 0.0, 0.0]); v[0] += 1; v[1] += us; v[2] += fl
for kind in ("conv", "wgrad"):
    tot = sum(v[1] for k, v in agg.items() if k[0] == kind)
    print(f"== {kind}: {tot / 1e3:.2f} ms in {sum(v[0] for k, v in agg.items() if k[0] == kind)} calls")
    for k, v in sorted(((k, v) for k, v in agg.items() if k[0] == kind), key=lambda kv: -kv[1][1]):
        print(f"  ci {k[1]:4d} co {k[2]:4d} map {k[3]:3d}: {v[0]:3d} calls {v[1] / 1e3:7.2f} ms  avg {v[1] / v[0]:7.1f} us  {v[2] / v[1] / 1e6:6.1f} TFLOP/s")
