#!/usr/bin/env python3
"""Does splitting the 12-tile batch of a stack into two half batches on two HIP streams fill the tails of the persistent conv
launches (one workgroup per CU; 7.5 / 13.5 / 25.5 items per CU leave the last round of a launch half empty)?
One engine per stream (own workspace), same weights; compares 1 x 12 tiles with 2 x 6 tiles concurrently.  python scripts/two_stream_probe.py [prec]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from fisr_amd import weights
from fisr_amd.fisrnet import FISRnet
prec = sys.argv[1] if len(sys.argv) > 1 else "fp32"
W = weights.synthetic_weights(2020)
nets = [FISRnet(device="cuda:0", precision=prec) for _ in range(4)]
for n in nets:
    n.set_weights(W)
g = torch.Generator(device="cuda").manual_seed(1)
x = torch.rand((12, 544, 992, 29), device="cuda", generator=g)
x[..., 9:17] = (x[..., 9:17] - 0.5) * 0.4
def run_one():
    nets[0].model(x, want_all=False)
ss = [torch.cuda.Stream() for _ in range(4)]
def run_two(k=2):
    cur = torch.cuda.current_stream()
    parts = x.chunk(k)
    for i, st in enumerate(ss[:k]):
        st.wait_stream(cur)
        with torch.cuda.stream(st):
            nets[i].model(parts[i], want_all=False)
    for st in ss[:k]:
        cur.wait_stream(st)
for name, fn in (("1 x 12 tiles, one stream", run_one), ("2 x 6 tiles, two streams", run_two), ("3 x 4 tiles, three streams", lambda: run_two(3)), ("4 x 3 tiles, four streams", lambda: run_two(4)), ("1 x 12 tiles, one stream", run_one), ("2 x 6 tiles, two streams", run_two), ("3 x 4 tiles, three streams", lambda: run_two(3))):
    fn(); torch.cuda.synchronize()
    t = time.time()
    for _ in range(4):
        fn()
    torch.cuda.synchronize()
    print(f"{prec} {name}: {(time.time() - t) / 4 * 1e3:.2f} ms per 12 tiles")
