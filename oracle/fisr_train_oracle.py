"""CPU ORACLE of the training graph -- TEST INFRASTRUCTURE ONLY, never the product path.

Only `tests/` may import this file (fisr_amd/ must not).

Restates `FISRnet.build_model` (FISRnet.py:175-497) on top of the torch twin of the network
(oracle/torch_cpu.py, FISRnet.py:73-173): the multiple-data-sample strategy -- three stride-1 windows of a
5-frame sample and one stride-2 window through the SAME weights (FISRnet.py:283-314, 394-415) --, the label
pyramid (`tf.image.resize_images(..., BICUBIC)` to exactly 1/2 and 1/4 size, FISRnet.py:262-263: with TF-1.13's
legacy sampling grid the source coordinate of output i is exactly 2i / 4i, where the cubic kernel is (0, 1, 0, 0),
i.e. plain sub-sampling, as at FISRnet.py:81,112), `Groups2Ovlp` (ops.py:119-144), the seven loss terms
(FISRnet.py:316-484, ops.py:24-32) and `tf.train.AdamOptimizer` (FISRnet.py:490-491; TF-1.13 defaults
beta1 0.9, beta2 0.999, epsilon 1e-8, update lr_t = lr * sqrt(1 - b2^t) / (1 - b1^t),
var -= lr_t * m / (sqrt(v) + eps)).  Gradients come from torch autograd in fp64.

PARITY UNPINNED: TensorFlow cannot be installed here (no network) and the reference holds no golden vectors for
training; what pins this file is mathematics -- tests/test_oracle.py checks the analytic gradients against central
finite differences of the restated loss.
"""
from __future__ import annotations

import numpy as np
import torch

import torch_cpu as tw

LAMBDAS = dict(recn=1.0, tm1=1.0, tm2=0.1, tmm=1.0, td=0.1, ss2=1.0)     # main.py:80-85 defaults


def network(x_nchw, Wt):
    """FISRnet.model (FISRnet.py:73-173) on an NCHW 29-channel input -> (pred_l1, pred_l2, pred_l3), NCHW 9 channels."""
    p1 = tw._level(x_nchw[:, :, ::4, ::4], Wt, "FISRnet/level_1")
    p2 = tw._level(torch.cat([x_nchw[:, :, ::2, ::2], p1], dim=1), Wt, "FISRnet/level_2")
    p3 = tw._level(torch.cat([x_nchw, p2], dim=1), Wt, "FISRnet/level_3")
    return p1, p2, p3


def window_input(data15, flow, warp, order):
    """Tensor_slicer_recurrent* (ops.py:92-117): data [B,15,H,W] -> 9 ch from 3*order, flow [B,16,H,W] -> 8 ch from
    4*order, warp [B,24,H,W] -> 12 ch from 6*order; concatenated (FISRnet.py:287-290)."""
    return torch.cat([data15[:, 3 * order:3 * order + 9], flow[:, 4 * order:4 * order + 8],
                      warp[:, 6 * order:6 * order + 12]], dim=1)


def split_frames(p):
    """tf_split_seq_dim (ops.py:156-160): [B, 3*k, H, W] -> list of k frames [B,3,H,W]."""
    return [p[:, 3 * i:3 * i + 3] for i in range(p.shape[1] // 3)]


def groups2ovlp(g):
    """ops.py:119-144 on a list of 9 frames -> 7 frames."""
    return [g[0], g[1], (g[2] + g[3]) / 2, g[4], (g[5] + g[6]) / 2, g[7], g[8]]


def l2(a, b):                      # ops.py:30-32
    return ((a - b) ** 2).mean()


def total_loss(Wt, data15, label21, flow16, warp24, flow_ss2, warp_ss2, lam=LAMBDAS):
    """All tensors NCHW torch (float64 for gradient work).  label21: [B, 21, 2H, 2W] (7 HR frames).
    Returns (total_loss, dict of the terms)."""
    gt = [split_frames(label21[:, :, ::4, ::4]), split_frames(label21[:, :, ::2, ::2]), split_frames(label21)]
    scale = (4.0, 2.0, 1.0)                                        # levels 1, 2, 3 (FISRnet.py:326-328)
    pred = [[], [], []]
    for order in range(3):                                         # FISRnet.py:283-310
        outs = network(window_input(data15, flow16, warp24, order), Wt)
        for l in range(3):
            pred[l] += split_frames(outs[l])
    ov = [groups2ovlp(pred[l]) for l in range(3)]                  # FISRnet.py:312-314
    recn = tm = tmm = td = 0.0
    for l in range(3):
        s = scale[l]
        for i in range(3):                                         # type 1 (:316-328)
            recn = recn + s * l2(torch.cat(pred[l][3 * i:3 * i + 3], 1), torch.cat(gt[l][2 * i:2 * i + 3], 1))
        for i in range(2):                                         # types 2, 3 (:330-358)
            a, b = pred[l][3 * i + 2], pred[l][3 * i + 3]
            tm = tm + s * l2(a, b)
            tmm = tmm + s * l2((a + b) / 2, gt[l][2 * (i + 1)])
        for i in range(6):                                         # type 4 (:360-387)
            td = td + s * l2(ov[l][i + 1] - ov[l][i], gt[l][i + 1] - gt[l][i])
    total_s1 = lam["recn"] * recn + lam["tm1"] * tm + lam["tmm"] * tmm + lam["td"] * td      # :389-391
    # stride 2 (:394-484): frames 0, 2, 4 of the sample, its own flows / warps, GT frames 1, 3, 5
    x2 = torch.cat([data15[:, 0:3], data15[:, 6:9], data15[:, 12:15], flow_ss2, warp_ss2], dim=1)
    outs2 = network(x2, Wt)
    recn2 = td2 = tm2 = 0.0
    for l in range(3):
        s = scale[l]
        p2 = split_frames(outs2[l])
        g2 = [gt[l][1], gt[l][3], gt[l][5]]
        recn2 = recn2 + s * l2(torch.cat(p2, 1), torch.cat(g2, 1))                              # type 5
        for i in range(2):                                                                       # type 6
            td2 = td2 + s * l2(p2[i + 1] - p2[i], g2[i + 1] - g2[i])
        tm2 = tm2 + s * l2(torch.cat(p2, 1), torch.cat([ov[l][1], ov[l][3], ov[l][5]], 1))      # type 7
    total_s2 = lam["recn"] * recn2 + lam["td"] * td2 + lam["tm2"] * tm2                          # :481-482
    total = total_s1 + lam["ss2"] * total_s2                                                    # :485
    terms = dict(recn=recn, tm=tm, tmm=tmm, td=td, recn_ss2=recn2, td_ss2=td2, tm_ss2=tm2, total=total)
    return total, terms


def to_nchw(a, dtype=torch.float64):
    return torch.from_numpy(np.ascontiguousarray(np.asarray(a))).to(dtype).permute(0, 3, 1, 2).contiguous()


def loss_and_grads(W, batch, lam=LAMBDAS, dtype=torch.float64):
    """W: {TF name: numpy HWIO / [Co]}; batch: dict of NHWC numpy arrays data15, label21, flow16, warp24, flow_ss2 (8 ch),
    warp_ss2 (12 ch).  Returns (loss, terms, {name: numpy gradient in the TF layout})."""
    Wt = {}
    for k, v in W.items():
        t = torch.from_numpy(np.ascontiguousarray(v)).to(dtype)
        Wt[k] = (t.permute(3, 2, 0, 1).contiguous() if k.endswith("/w") else t).requires_grad_(True)
    args = [to_nchw(batch[k], dtype) for k in ("data15", "label21", "flow16", "warp24", "flow_ss2", "warp_ss2")]
    total, terms = total_loss(Wt, *args, lam=lam)
    total.backward()
    grads = {k: (t.grad.permute(2, 3, 1, 0) if k.endswith("/w") else t.grad).contiguous().numpy() for k, t in Wt.items()}
    return float(total.detach()), {k: float(v.detach()) for k, v in terms.items()}, grads


def adam_step(W, grads, m, v, step, lr, b1=0.9, b2=0.999, eps=1e-8):
    """tf.train.AdamOptimizer.apply_gradients as TF 1.13 computes it (training/adam.py): in place on numpy dicts."""
    lr_t = lr * np.sqrt(1.0 - b2 ** step) / (1.0 - b1 ** step)
    for k in W:
        m[k] = b1 * m[k] + (1 - b1) * grads[k]
        v[k] = b2 * v[k] + (1 - b2) * grads[k] ** 2
        W[k] = W[k] - lr_t * m[k] / (np.sqrt(v[k]) + eps)


def synthetic_batch(seed, b, h, w):
    """A seeded training batch in the reference's value ranges: frames and labels in [0, 1], flows already divided by
    (patch size * 2) (FISRnet.py:198, 203), warped frames in [0, 1]."""
    r = np.random.default_rng(seed)
    f32 = np.float32
    return dict(data15=r.random((b, h, w, 15), dtype=f32), label21=r.random((b, 2 * h, 2 * w, 21), dtype=f32),
                flow16=(r.standard_normal((b, h, w, 16)) * 0.02).astype(f32), warp24=r.random((b, h, w, 24), dtype=f32),
                flow_ss2=(r.standard_normal((b, h, w, 8)) * 0.04).astype(f32), warp_ss2=r.random((b, h, w, 12), dtype=f32))
