#!/usr/bin/env python3
"""Per-step times of the headline loop (fp32, cfg2): does the step time drift while the chip warms up, and what do power / clock /
temperature read around it?   python scripts/step_times.py [steps]   (on the GPU box)"""
import importlib.util, os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
b = importlib.util.module_from_spec(spec); spec.loader.exec_module(b)
import torch
from fisr_amd import weights
from fisr_amd.fisrnet import FISRnet


def smi():
    try:
        o = subprocess.run(["rocm-smi", "--showpower", "--showclocks", "--showtemp"], capture_output=True, text=True, timeout=20).stdout
        keep = [l.split(":", 1)[-1].strip() + " (" + l.split(":")[1].strip() + ")" for l in o.splitlines()
                if any(k in l for k in ("Package Power", "sclk", "junction"))]
        return "; ".join(keep)
    except Exception as e:       # noqa: BLE001
        return repr(e)


n = int(sys.argv[1]) if len(sys.argv) > 1 else 40
dev = torch.device("cuda", 0)
net = FISRnet(device="cuda:0", precision="fp32")
net.set_weights(weights.synthetic_weights(2020))
wl = b.Workload(torch, dev, 0, (2, 2), "stack")
wl.premake_warps(net)
print("before:", smi())
wl.step(net); torch.cuda.synchronize()
evs = [torch.cuda.Event(enable_timing=True) for _ in range(n + 1)]
t0 = time.perf_counter()
evs[0].record()
for i in range(n):
    wl.step(net)
    evs[i + 1].record()
torch.cuda.synchronize()
wall = (time.perf_counter() - t0) / n * 1e3
ms = [evs[i].elapsed_time(evs[i + 1]) for i in range(n)]
print("after: ", smi())
print("ms per step (HIP events, back to back, no sync in between): " + " ".join("%.1f" % m for m in ms))
print("mean %.2f  first five %.2f  last five %.2f  wall per step %.2f" % (sum(ms) / n, sum(ms[:5]) / 5, sum(ms[-5:]) / 5, wall))
