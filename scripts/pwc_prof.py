#!/usr/bin/env python3
"""One PWC-Net-large flow pair (both directions) on a 1080p frame pair, for rocprofv3 --kernel-trace --stats."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from fisr_amd import pwcnet
dev = "cuda:0"
pwc = pwcnet.PWCNet(dev)
pwc.set_weights(pwcnet.synthetic_weights(595000))
g = torch.Generator().manual_seed(1)
a = torch.randint(0, 256, (1080, 1920, 3), generator=g, dtype=torch.uint8).to(dev)
b = torch.randint(0, 256, (1080, 1920, 3), generator=g, dtype=torch.uint8).to(dev)
for _ in range(int(sys.argv[1]) if len(sys.argv) > 1 else 2):
    t = time.time(); pwc.flow_pair(a, b); torch.cuda.synchronize(); print("flow pair ms", (time.time() - t) * 1e3)
