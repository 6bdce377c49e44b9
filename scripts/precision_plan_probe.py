#!/usr/bin/env python3
"""Which tensors of the network have to be better than fp16?  The graph of oracle/torch_cpu.py on the GPU in fp32 (torch /
MIOpen: a measuring stick, not the product), with fp16 rounding injected where a precision plan would put it:
  fp16      operands and every stored tensor in fp16 everywhere                                   (engine `fp16`:  known 0.026 dB)
  mixed     level 3 at full and half resolution exact, fp16 elsewhere                             (engine `mixed`: known 0.004 dB)
  trunk     like mixed, but the convs of the exact region take fp16 OPERANDS (activations and weights rounded on load)
            and only the residual trunk / skip tensors stay exact; res-block intermediates are stored in fp16
  trunk_full  `trunk` at full resolution only, half resolution all-fp16
  hi_w16    like mixed, but the exact region's WEIGHTS are rounded to fp16 (activations exact)
  hi_a16    like mixed, but the exact region's conv ACTIVATION operands are rounded to fp16 on load (weights exact; storage exact)
  hi_a16s   like hi_a16 and every tensor of the region is also STORED in fp16 (fp16 activations end to end, exact weights)
Prints the PSNR shift of the SR / FI channels per plan (protocol of SURVEY.md 8c-ii: pseudo ground truth at the reference's
published PSNRs).  python scripts/precision_plan_probe.py [h] [w] [n]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch, torch.nn.functional as F
from fisr_amd import weights
from tests_support import make_full_size_input

dev = "cuda"
HI = ("FISRnet/level_3/enc/level_0/", "FISRnet/level_3/enc/level_1/", "FISRnet/level_3/dec/level_1/", "FISRnet/level_3/dec/level_0/",
      "FISRnet/level_3/FI-SR/", "FISRnet/level_3/SR/")
HI_FULL = ("FISRnet/level_3/enc/level_0/", "FISRnet/level_3/dec/level_0/", "FISRnet/level_3/FI-SR/", "FISRnet/level_3/SR/")

def q16(t): return t.half().float()

class Plan:
    def __init__(self, name): self.name = name
    def hi(self, lname):
        if self.name in ("fp32",): return True
        if self.name == "fp16": return False
        reg = HI_FULL if self.name in ("trunk_full", "mixed_full") else HI
        return any(lname.startswith(p) for p in reg)
    def operands16(self, lname):        # conv reads fp16 operands: (activations, weights)
        if not self.hi(lname) or self.name in ("trunk", "trunk_full"): return (True, True)
        if self.name == "hi_w16": return (False, True)
        if self.name in ("hi_a16", "hi_a16s"): return (True, False)
        return (False, False)
    def store(self, t, lname, kind):    # kind: "trunk" (residual stream, skips, level outputs) or "mid" (consumed by one conv)
        if self.name == "fp32": return t
        if not self.hi(lname) or self.name == "hi_a16s": return q16(t)
        if self.name in ("trunk", "trunk_full") and kind == "mid": return q16(t)
        return t

def conv(x, W, name, P):
    w, b = W[name + "/w"], W[name + "/b"]
    a16, w16 = P.operands16(name)
    if a16: x = q16(x)
    if w16: w = q16(w)
    return F.conv2d(x, w, b, padding=1)

def res_block(x, W, name, P, relu_after=False):
    a = P.store(F.relu(conv(F.relu(x), W, name + "/conv/0", P)), name, "mid")
    y = x + conv(a, W, name + "/conv/1", P)
    if relu_after: y = F.relu(y)
    return P.store(y, name, "trunk")

def up2(x):
    n, c, h, w = x.shape
    xr = torch.cat([x[..., 1:], x[..., -1:]], dim=3)
    top = torch.stack([x, x + (xr - x) * 0.5], dim=4).reshape(n, c, h, 2 * w)
    tb = torch.cat([top[:, :, 1:], top[:, :, -1:]], dim=2)
    return torch.stack([top, top + (tb - top) * 0.5], dim=3).reshape(n, c, 2 * h, 2 * w)

def d2s(x):
    n, c4, h, w = x.shape; c = c4 // 4
    return x.reshape(n, 2, 2, c, h, w).permute(0, 3, 4, 1, 5, 2).reshape(n, c, 2 * h, 2 * w)

def enc(x, W, name, P):
    n = P.store(conv(x, W, name + "/conv/0", P), name, "trunk")
    n = res_block(n, W, name + "/res_block/0", P)
    n = res_block(n, W, name + "/res_block/1", P, relu_after=True)
    return F.max_pool2d(n, 2), n

def dec(x, skip, W, name, P):
    n = P.store(F.relu(conv(up2(x), W, name + "/resize", P)), name, "mid")
    n = P.store(conv(torch.cat([n, skip], dim=1), W, name + "/conv/0", P), name, "trunk")
    n = res_block(n, W, name + "/res_block/0", P)
    return res_block(n, W, name + "/res_block/1", P, relu_after=True)

def level(x, W, p, P):
    x = P.store(x, p + "/enc/level_0/", "trunk")
    n, s0 = enc(x, W, p + "/enc/level_0", P)
    n, s1 = enc(n, W, p + "/enc/level_1", P)
    n = P.store(n, p + "/enc/level_2/", "trunk")
    n, s2 = enc(n, W, p + "/enc/level_2", P)
    n = P.store(conv(n, W, p + "/bottleneck/conv/0", P), p + "/bottleneck/", "trunk")
    n = res_block(n, W, p + "/bottleneck/res_block/0", P, relu_after=True)
    n = dec(n, s2, W, p + "/dec/level_2", P)
    n = dec(n, s1, W, p + "/dec/level_1", P)
    n = dec(n, s0, W, p + "/dec/level_0", P)
    outs = []
    for head in ("FI-SR", "SR"):
        h = p + "/" + head
        a = P.store(conv(n, W, h + "/conv/0", P), h + "/", "trunk")
        a = res_block(a, W, h + "/res_block/0", P)
        a = P.store(d2s(F.relu(conv(F.relu(a), W, h + "/conv/1", P))), h + "/", "mid")
        outs.append(conv(a, W, h + "/conv/2", P))
    fisr, sr = outs
    return torch.cat([fisr[:, 0:3], sr, fisr[:, 3:6]], dim=1)

def forward(x, W, P):
    with torch.no_grad():
        p1 = level(x[:, :, ::4, ::4], W, "FISRnet/level_1", P)
        p2 = level(torch.cat([x[:, :, ::2, ::2], p1], dim=1), W, "FISRnet/level_2", P)
        return level(torch.cat([x, p2], dim=1), W, "FISRnet/level_3", P)

def psnr_shift(rms, db): return 4.343 * (rms / 10 ** (-db / 20)) ** 2

h = int(sys.argv[1]) if len(sys.argv) > 1 else 272
w = int(sys.argv[2]) if len(sys.argv) > 2 else 496
n = int(sys.argv[3]) if len(sys.argv) > 3 else 2
torch.backends.cudnn.allow_tf32 = False
for ws in ("default", "survey_spec", "harsh"):
    Wn = weights.synthetic_weights(2020) if ws == "default" else weights.WEIGHT_SETS[ws]()
    W = {k: (torch.from_numpy(np.ascontiguousarray(v)).permute(3, 2, 0, 1).contiguous() if k.endswith("/w") else torch.from_numpy(np.ascontiguousarray(v))).to(dev) for k, v in Wn.items()}
    for seed in (12, 13):
        x = torch.from_numpy(make_full_size_input(seed, h, w, n)).permute(0, 3, 1, 2).contiguous().to(dev)
        ref = forward(x, W, Plan("fp32")).clamp(0, 1)
        for plan in ("fp16", "mixed", "mixed_full", "trunk", "hi_w16", "hi_a16", "hi_a16s"):
            e = forward(x, W, Plan(plan)).clamp(0, 1) - ref
            sr = float((e[:, 3:6] ** 2).mean().sqrt()); fi = float((torch.cat([e[:, 0:3], e[:, 6:9]], 1) ** 2).mean().sqrt())
            print(f"{ws:12s} seed {seed} {plan:11s} SR rms {sr:.2e} dPSNR {psnr_shift(sr, 48.07):.4f} dB   FI rms {fi:.2e} dPSNR {psnr_shift(fi, 37.86):.4f} dB", flush=True)
