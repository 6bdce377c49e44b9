#!/usr/bin/env python3
"""Time of the fp32 heads inside one step (library profile), for A/B builds via $FISR_HIP_SO."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from fisr_amd import weights
from fisr_amd.fisrnet import FISRnet
net = FISRnet(device="cuda:0", precision="fp32"); net.set_weights(weights.synthetic_weights(2020))
x = torch.rand(12, 544, 992, 29, device="cuda")
net.model(x); net.profile(2); net.model(x); torch.cuda.synchronize()
for p in net.profile_read():
    if "conv/2" in p["name"] and "1088" in p["name"]:
        print(os.environ.get("FISR_HIP_SO", "in-tree").split("/")[-1], p["name"], round(p["ms"] * 1e3), "us")
