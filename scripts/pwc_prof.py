#!/usr/bin/env python3
"""The flow of one 5-frame 1080p stack (8 directions, one fisr_pwc_flow_stack call), for timing and rocprofv3 --kernel-trace --stats:
python scripts/pwc_prof.py [reps] [fp32|fp16]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from fisr_amd import pwcnet
dev = "cuda:0"
prec = sys.argv[2] if len(sys.argv) > 2 else "fp32"
pwc = pwcnet.PWCNet(dev, precision=prec)
pwc.set_weights(pwcnet.synthetic_weights(595000))
g = torch.Generator().manual_seed(1)
fr = [torch.randint(0, 256, (1080, 1920, 3), generator=g, dtype=torch.uint8).to(dev) for _ in range(5)]
out = None
for _ in range(int(sys.argv[1]) if len(sys.argv) > 1 else 2):
    t = time.time(); out = pwc.flow_stack(fr, out=out); torch.cuda.synchronize(); print(prec, "flow of a 5-frame stack (8 directions): ms", round((time.time() - t) * 1e3, 2))
