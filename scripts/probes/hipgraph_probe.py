#!/usr/bin/env python3
"""Does a captured HIP graph of one forward (12 tiles, 149 launches) beat the eager stream?  The host enqueues a step in 0.8 ms, far ahead
of the GPU's 113 ms, so only the GPU-side gaps between dependent launches are at stake.  python scripts/probes/hipgraph_probe.py [prec]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from fisr_amd import weights
from fisr_amd.fisrnet import FISRnet
prec = sys.argv[1] if len(sys.argv) > 1 else "fp32"
net = FISRnet(device="cuda:0", precision=prec)
net.set_weights(weights.synthetic_weights(2020))
g = torch.Generator(device="cuda").manual_seed(1)
x = torch.rand((12, 544, 992, 29), device="cuda", generator=g)
x[..., 9:17] = (x[..., 9:17] - 0.5) * 0.4
ref = net.model(x, want_all=False)[-1].clone()
torch.cuda.synchronize()
def timeit(fn, n=5):
    fn(); torch.cuda.synchronize()
    t = time.time()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.time() - t) / n * 1e3
print(prec, "eager: %.2f ms per 12 tiles" % timeit(lambda: net.model(x, want_all=False)))
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for _ in range(2):
        net.model(x, want_all=False)
torch.cuda.current_stream().wait_stream(s)
gr = torch.cuda.CUDAGraph()
try:
    with torch.cuda.graph(gr):
        out = net.model(x, want_all=False)[-1]
except Exception as e:
    print("capture failed:", repr(e)[:400]); sys.exit(0)
gr.replay(); torch.cuda.synchronize()
print("graph output identical:", bool(torch.equal(out, ref)))
print(prec, "graph: %.2f ms per 12 tiles" % timeit(gr.replay))
print(prec, "eager: %.2f ms per 12 tiles" % timeit(lambda: net.model(x, want_all=False)))
print(prec, "graph: %.2f ms per 12 tiles" % timeit(gr.replay))
