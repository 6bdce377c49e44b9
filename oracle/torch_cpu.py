"""CPU ORACLE, torch/oneDNN twin -- TEST INFRASTRUCTURE ONLY, never the product path.

Only `tests/` and `bench.py`'s `cpu_baseline` leg may import this file (fisr_amd/ must not).

The same graph as oracle/fisr_oracle.py (FISRnet.py:73-173 over ops.py:7-76), with the convolutions,
pooling and element-wise work done by PyTorch's CPU kernels (oneDNN/MKL-DNN direct + Winograd convolution,
channels-last, all host cores): the closest stand-in available here for the reference's TensorFlow-1.13
Eigen/MKL-DNN CPU path (TensorFlow itself cannot be installed: no network).  Used by bench.py as
`cpu_baseline_onednn` and cross-checked against the numpy /
C oracles in tests/test_oracle.py.  The TF-specific ops (legacy x2 bilinear, DCR depth_to_space, strided
"bicubic" down-sampling) are restated by hand exactly as in fisr_oracle.py -- torch's own interpolate /
pixel_shuffle have different semantics.  PARITY UNPINNED (same status as fisr_oracle.py).
"""
from __future__ import annotations

import numpy as np
import torch
import torch.nn.functional as F


def prepare_weights(W, dtype=torch.float32):
    """TF HWIO [3,3,Ci,Co] -> torch OIHW, channels-last memory format."""
    out = {}
    for k, v in W.items():
        t = torch.from_numpy(np.ascontiguousarray(v)).to(dtype)
        out[k] = t.permute(3, 2, 0, 1).contiguous(memory_format=torch.channels_last) if k.endswith("/w") else t
    return out


def _conv(x, W, name):            # ops.py:7-11
    return F.conv2d(x, W[name + "/w"], W[name + "/b"], stride=1, padding=1)


def _res_block(x, W, name):       # ops.py:39-44
    n = _conv(F.relu(x), W, name + "/conv/0")
    n = _conv(F.relu(n), W, name + "/conv/1")
    return x + n


def _enc(x, W, name):             # ops.py:48-55
    n = _conv(x, W, name + "/conv/0")
    n = _res_block(n, W, name + "/res_block/0")
    n = F.relu(_res_block(n, W, name + "/res_block/1"))
    return F.max_pool2d(n, 2), n


def _up2(x):
    """ops.py:69: TF-1.13 legacy bilinear x2 (no half-pixel centres): even outputs copy, odd outputs average
    with the next sample (clamped at the edge); order of evaluation as in fisr_oracle.resize_bilinear_x2."""
    n, c, h, w = x.shape
    xr = torch.cat([x[..., 1:], x[..., -1:]], dim=3)
    top = torch.stack([x, x + (xr - x) * 0.5], dim=4).reshape(n, c, h, 2 * w)
    tb = torch.cat([top[:, :, 1:], top[:, :, -1:]], dim=2)
    return torch.stack([top, top + (tb - top) * 0.5], dim=3).reshape(n, c, 2 * h, 2 * w)


def _dec(x, skip, W, name):       # ops.py:67-76
    n = F.relu(_conv(_up2(x), W, name + "/resize"))
    n = _conv(torch.cat([n, skip], dim=1), W, name + "/conv/0")
    n = _res_block(n, W, name + "/res_block/0")
    return F.relu(_res_block(n, W, name + "/res_block/1"))


def _d2s(x):
    """FISRnet.py:99 tf.depth_to_space(x,2), NHWC 'DCR': out[2h+i,2w+j,c] = x[h,w,(2i+j)*C+c] (NCHW here)."""
    n, c4, h, w = x.shape
    c = c4 // 4
    return x.reshape(n, 2, 2, c, h, w).permute(0, 3, 4, 1, 5, 2).reshape(n, c, 2 * h, 2 * w)


def _level(x, W, p):              # FISRnet.py:83-108
    n, s0 = _enc(x, W, p + "/enc/level_0")
    n, s1 = _enc(n, W, p + "/enc/level_1")
    n, s2 = _enc(n, W, p + "/enc/level_2")
    n = _conv(n, W, p + "/bottleneck/conv/0")
    n = F.relu(_res_block(n, W, p + "/bottleneck/res_block/0"))
    n = _dec(n, s2, W, p + "/dec/level_2")
    n = _dec(n, s1, W, p + "/dec/level_1")
    n = _dec(n, s0, W, p + "/dec/level_0")
    outs = []
    for head in ("FI-SR", "SR"):
        h = p + "/" + head
        a = _conv(n, W, h + "/conv/0")
        a = _res_block(a, W, h + "/res_block/0")
        a = _d2s(F.relu(_conv(F.relu(a), W, h + "/conv/1")))
        outs.append(_conv(a, W, h + "/conv/2"))          # relu(relu(.)) of FI-SR is idempotent (FISRnet.py:100)
    fisr, sr = outs
    return torch.cat([fisr[:, 0:3], sr, fisr[:, 3:6]], dim=1)   # FISRnet.py:107-108


def forward(img_nhwc, Wt):
    """img [N,H,W,29] numpy/torch -> (pred_l1, pred_l2, pred_l3) numpy NHWC.  Wt from prepare_weights()."""
    dtype = next(iter(Wt.values())).dtype
    x = torch.as_tensor(np.asarray(img_nhwc)).to(dtype).permute(0, 3, 1, 2).contiguous(memory_format=torch.channels_last)
    with torch.no_grad():
        p1 = _level(x[:, :, ::4, ::4], Wt, "FISRnet/level_1")                       # FISRnet.py:81
        p2 = _level(torch.cat([x[:, :, ::2, ::2], p1], dim=1), Wt, "FISRnet/level_2")   # :112-113
        p3 = _level(torch.cat([x, p2], dim=1), Wt, "FISRnet/level_3")               # :144
    return tuple(p.permute(0, 2, 3, 1).contiguous().numpy() for p in (p1, p2, p3))
