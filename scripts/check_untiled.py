import sys, time, numpy as np, torch
sys.path.insert(0, '.')
from fisr_amd import weights
from fisr_amd.fisrnet import FISRnet
net = FISRnet(device="cuda:0", precision="bf16x3"); net.set_weights(weights.synthetic_weights(2020))
x = torch.rand((1, 1056, 1920, 29), device="cuda")
x[..., 9:17] = (x[..., 9:17] - 0.5) * 0.4
t0 = time.time(); full = net.forward_tiled(x, (1, 1)); torch.cuda.synchronize(); t1 = time.time()
full2 = net.forward_tiled(x, (1, 1)); torch.cuda.synchronize(); t2 = time.time()
print("untiled 1056x1920 -> ", tuple(full.shape), "finite", bool(torch.isfinite(full).all()), "first %.3fs second %.3fs" % (t1 - t0, t2 - t1), "deterministic", bool(torch.equal(full, full2)))
# interior of the untiled result vs the 2x2-tiled result on the common 1024 rows: they differ only near the tile seams' zero padding
xt = x[:, :1024].contiguous()
tiled = net.forward_tiled(xt, (2, 2)); torch.cuda.synchronize()
d = (full[:2048] - tiled).abs()
print("max |untiled - tiled| away from seams:", float(d[200:800, 200:1700].max()), " overall:", float(d.max()))
