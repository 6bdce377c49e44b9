"""Training graph of FISRnet on the GPU (FISRnet.py:175-497, ops.py:7-160).

The reference builds four weight-sharing forward passes per 5-frame sample (three stride-1 windows and one
stride-2 window, FISRnet.py:283-314, 394-415), seven multi-scale loss terms (FISRnet.py:316-484) and
`tf.train.AdamOptimizer(lr).minimize(total_loss)` (FISRnet.py:490-491) in Python, and TensorFlow's autodiff supplies
the backward graph.  This module keeps that shape: the network is written once as a sequence of ops (`_level`,
the mirror of FISRnet.model) and every op call is recorded on a tape that `backward()` walks in reverse.  Every op --
forward, data gradient, weight gradient, the element-wise adjoints, the loss and Adam -- is a HIP kernel behind the
C-ABI (`fisr_train_*`, include/fisr.h); PyTorch only owns the device memory.  fp32 throughout: the forward and the
data gradients run on the inference engine's direct exact-fp32 MFMA kernel (the data gradient of a 3x3 SAME
convolution is the same convolution with the taps rotated by 180 degrees and Cin/Cout swapped), the weight gradient on
`train_wgrad_kernel` (a GEMM over the pixel axis on the same MFMA).

There is no CPU fallback: without the library and a GPU every call raises.
"""
from __future__ import annotations

import ctypes
import math
from collections import OrderedDict

import numpy as np

from . import lib as _lib
from . import weights as _weights

RELU_IN, RELU_OUT, D2S = _lib.CONV_RELU_IN, _lib.CONV_RELU_OUT, _lib.CONV_D2S
LAMBDAS = dict(recn=1.0, tm1=1.0, tm2=0.1, tmm=1.0, td=0.1, ss2=1.0)      # main.py:80-85
LEVEL_SCALE = {"level_1": 4.0, "level_2": 2.0, "level_3": 1.0}            # FISRnet.py:326-328


def _pad4(c):
    return (c + 3) // 4 * 4


def _pad16(c):
    return (c + 15) // 16 * 16


class _Conv:
    """One conv layer's device state: master weights (TF layout), gradients, Adam slots, the two packed copies."""
    __slots__ = ("name", "ci", "co", "w", "b", "gw", "gb", "mw", "vw", "mb", "vb", "pk", "pk_t", "pkw", "pkw_t")


class TrainNet:
    def __init__(self, weights=None, device="cuda:0", seed=2020, lambdas=None):
        import torch
        self.torch = torch
        self.L = _lib.lib()
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise _lib.FisrError("fisr_amd.train needs a GPU device (there is no CPU fallback)")
        self.lam = dict(LAMBDAS if lambdas is None else lambdas)
        W = weights if weights is not None else _weights.synthetic_weights(seed)
        _weights.check_complete(W)
        self.convs = OrderedDict()
        f32 = torch.float32
        # Weights, gradients and Adam's two slots live in FOUR flat buffers of one layout (views per tensor, every slot a
        # multiple of four floats so that the conv kernels' 16-byte bias reads stay aligned, 64 floats of tail so that a bias
        # read up to the N block's padding stays inside the buffer): the step's all-reduce and Adam are then one call each.
        specs = _weights.conv_specs()
        total = sum(_pad4(9 * ci * co) + _pad4(co) for _, ci, co in specs) + 64
        self.wflat, self.gflat, self.mflat, self.vflat = (torch.zeros(total, dtype=f32, device=self.device) for _ in range(4))
        off = 0
        descs = np.zeros(len(specs), dtype=[("w", "u8"), ("pk", "u8"), ("pk_t", "u8"), ("pkw", "u8"), ("pkw_t", "u8"), ("ci", "i4"), ("co", "i4")])
        for k, (name, ci, co) in enumerate(specs):
            c = _Conv()
            c.name, c.ci, c.co = name, ci, co
            nw_, nb_ = 9 * ci * co, co
            c.w, c.gw, c.mw, c.vw = (f[off:off + nw_].view(3, 3, ci, co) for f in (self.wflat, self.gflat, self.mflat, self.vflat))
            off += _pad4(nw_)
            c.b, c.gb, c.mb, c.vb = (f[off:off + nb_] for f in (self.wflat, self.gflat, self.mflat, self.vflat))
            off += _pad4(nb_)
            c.w.copy_(torch.from_numpy(np.ascontiguousarray(W[name + "/w"], dtype=np.float32)))
            c.b.copy_(torch.from_numpy(np.ascontiguousarray(W[name + "/b"], dtype=np.float32)))
            # one packed layout per conv and orientation: Winograd slabs where the conv is eligible (0 bytes = it is not), the
            # direct kernel's rows otherwise
            nw, nwt = self.L.fisr_train_wino_bytes(ci, co, 0), self.L.fisr_train_wino_bytes(ci, co, 1)
            c.pkw = torch.empty(nw // 4, dtype=f32, device=self.device) if nw else None
            c.pkw_t = torch.empty(nwt // 4, dtype=f32, device=self.device) if nwt else None
            c.pk = None if nw else torch.empty(self.L.fisr_train_packed_bytes(ci, co, 0) // 4, dtype=f32, device=self.device)
            c.pk_t = None if nwt else torch.empty(self.L.fisr_train_packed_bytes(ci, co, 1) // 4, dtype=f32, device=self.device)
            descs[k] = tuple(t.data_ptr() if t is not None else 0 for t in (c.w, c.pk, c.pk_t, c.pkw, c.pkw_t)) + (ci, co)
            self.convs[name] = c
        self._pack_descs = torch.from_numpy(descs.view(np.uint8).copy()).to(self.device)       # fisr_train_pack_desc[], on the device
        self.zero_bias = torch.zeros(1024, dtype=f32, device=self.device)
        self.step_count = 0
        self.tape = []
        self.keep_preds = False        # loss_and_grads() then leaves the four passes' predictions in last_preds
        self.grad_scale = 1.0          # data parallel: 1 / world (see allreduce_grads)
        self.last_preds = None
        # Weight gradients run on a second stream: nothing but Adam consumes them, so each wgrad launch overlaps the data-gradient
        # chain of the layers below it and fills the CUs that the small maps of the lower U-Net levels leave idle.
        self.overlap_wgrad = True
        self._side = torch.cuda.Stream(device=self.device)
        self.repack()

    # ------------------------------------------------------------------ plumbing
    def _st(self):
        return ctypes.c_void_p(self.torch.cuda.current_stream(self.device).cuda_stream)

    def _ck(self, rc):
        if rc < 0:
            raise _lib.FisrError(f"fisr_train op failed ({rc}): {self.L.fisr_last_error(None).decode()}")

    @staticmethod
    def _p(t):
        return ctypes.c_void_p(t.data_ptr()) if t is not None else None

    def new(self, *shape):
        return self.torch.empty(shape, dtype=self.torch.float32, device=self.device)

    def zeros(self, *shape):
        return self.torch.zeros(shape, dtype=self.torch.float32, device=self.device)

    def repack(self):
        """Master weights -> the conv kernel's layouts (after construction and after every Adam step)."""
        self._ck(self.L.fisr_train_pack_all(self._p(self._pack_descs), len(self.convs), self._st()))       # one launch for all 138 convs

    def weights_numpy(self):
        out = OrderedDict()
        for c in self.convs.values():
            out[c.name + "/w"] = c.w.cpu().numpy()
            out[c.name + "/b"] = c.b.cpu().numpy()
        return out

    def optimizer_state_numpy(self, b1=0.9, b2=0.999):
        """Adam's state under TensorFlow's names: `<var>/Adam` (m), `<var>/Adam_1` (v), `beta1_power`, `beta2_power`
        (TF 1.13 keeps b^(t+1) after t steps: initialised to b, multiplied once per step)."""
        out = OrderedDict()
        for c in self.convs.values():
            out[c.name + "/w/Adam"] = c.mw.cpu().numpy()
            out[c.name + "/w/Adam_1"] = c.vw.cpu().numpy()
            out[c.name + "/b/Adam"] = c.mb.cpu().numpy()
            out[c.name + "/b/Adam_1"] = c.vb.cpu().numpy()
        out["beta1_power"] = np.float64(b1 ** (self.step_count + 1))
        out["beta2_power"] = np.float64(b2 ** (self.step_count + 1))
        return out

    def load_optimizer_state(self, state, b2=0.999, fallback_step=0):
        """Restore what optimizer_state_numpy() wrote; the step count of the bias correction comes from beta2_power
        (`fallback_step` -- the step in the checkpoint's file name -- when a float32 beta power has underflowed to 0, which
        TensorFlow's own does after ~88 000 steps: the correction is 1 by then)."""
        torch = self.torch
        for c in self.convs.values():
            for dst, key in ((c.mw, "/w/Adam"), (c.vw, "/w/Adam_1"), (c.mb, "/b/Adam"), (c.vb, "/b/Adam_1")):
                a = np.ascontiguousarray(state[c.name + key], np.float32)
                if tuple(a.shape) != tuple(dst.shape):
                    raise ValueError(f"{c.name}{key}: shape {a.shape} != {tuple(dst.shape)}")
                dst.copy_(torch.from_numpy(a))
        p2 = float(state["beta2_power"])
        self.step_count = max(0, int(round(math.log(p2) / math.log(b2))) - 1) if 0.0 < p2 < 1.0 else int(fallback_step)

    def grads_numpy(self):
        out = OrderedDict()
        for c in self.convs.values():
            out[c.name + "/w"] = c.gw.cpu().numpy()
            out[c.name + "/b"] = c.gb.cpu().numpy()
        return out

    def zero_grad(self):
        self.gflat.zero_()

    def allreduce_grads(self, group=None):
        """Data-parallel training: every rank ran loss_and_grads on its shard of the batch with `grad_scale = 1 / world`
        (all seven loss terms are means over the batch), so the SUM over ranks is the gradient of the whole batch.  One
        all-reduce of the flat gradient buffer: RCCL over xGMI under the `nccl` backend; other backends (gloo in the
        one-GPU tests) stage through the host."""
        import torch.distributed as dist
        if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
            return
        if dist.get_backend(group) == "nccl":
            dist.all_reduce(self.gflat, op=dist.ReduceOp.SUM, group=group)
        else:
            h = self.gflat.cpu()
            dist.all_reduce(h, op=dist.ReduceOp.SUM, group=group)
            self.gflat.copy_(h)

    # ------------------------------------------------------------------ forward ops (recorded)
    def conv(self, name, x0, x1=None, res=None, flags=0, scatter=None, out=None):
        """ops.py:7-11 (+ the fused relu / residual / depth_to_space).  scatter = (dst [B,H,W,9], coff, split, gap) for the heads."""
        c = self.convs[name]
        n, h, w, c0 = x0.shape
        c1 = x1.shape[3] if x1 is not None else 0
        if scatter is not None:
            dst, coff, split, gap = scatter
            self._ck(self.L.fisr_train_conv3x3(self._p(x0), c0, self._p(x1), c1, self._p(c.pk), self._p(c.b), c.co, None,
                                               self._p(dst), n, h, w, flags, dst.shape[3], coff, split, gap, None, self._st()))
            self.tape.append(("conv", c, x0, x1, None, dst, flags, scatter))
            return dst
        y = out if out is not None else (self.new(n, 2 * h, 2 * w, c.co // 4) if flags & D2S else self.new(n, h, w, c.co))
        self._ck(self.L.fisr_train_conv3x3(self._p(x0), c0, self._p(x1), c1, self._p(c.pk), self._p(c.b), c.co, self._p(res),
                                           self._p(y), n, h, w, flags, 0, 0, 0, 0, self._p(c.pkw), self._st()))
        self.tape.append(("conv", c, x0, x1, res, y, flags, None))
        return y

    def maxpool(self, x):
        n, h, w, c = x.shape
        y = self.new(n, h // 2, w // 2, c)
        self._ck(self.L.fisr_op_maxpool2(self._p(x), self._p(y), n, h, w, c, _lib.PREC_F32, self._st()))
        self.tape.append(("pool", x, y))
        return y

    def up2(self, x):
        n, h, w, c = x.shape
        y = self.new(n, 2 * h, 2 * w, c)
        self._ck(self.L.fisr_op_upsample2(self._p(x), self._p(y), n, h, w, c, _lib.PREC_F32, self._st()))
        self.tape.append(("up", x, y))
        return y

    def level_input(self, x29, prev_pred):
        """FISRnet.py:84,113-116,144-147: the level's input, channels padded to a multiple of 16 with zeros."""
        n, h, w, _ = x29.shape
        cin = 29 + (9 if prev_pred is not None else 0)
        t = self.zeros(n, h, w, _pad16(cin))
        t[..., :29].copy_(x29)
        if prev_pred is not None:
            self._ck(self.L.fisr_train_copy_channels(self._p(prev_pred), 9, 0, self._p(t), t.shape[3], 29, 9, n * h * w, 0, self._st()))
            self.tape.append(("cat_pred", prev_pred, t))
        t._fisr_needs_grad = prev_pred is not None       # level 1's input is data only: no data gradient for its first conv
        return t

    # ------------------------------------------------------------------ the network (FISRnet.py:73-173, ops.py:39-76)
    def _res_block(self, x, p, relu_after=False):
        a = self.conv(p + "/conv/0", x, flags=RELU_IN | RELU_OUT)
        return self.conv(p + "/conv/1", a, res=x, flags=RELU_OUT if relu_after else 0)

    def _enc(self, x, p):
        n = self.conv(p + "/conv/0", x)
        n = self._res_block(n, p + "/res_block/0")
        n = self._res_block(n, p + "/res_block/1", relu_after=True)
        return self.maxpool(n), n

    def _dec(self, x, skip, p):
        n = self.conv(p + "/resize", self.up2(x), flags=RELU_OUT)
        n = self.conv(p + "/conv/0", n, x1=skip)
        n = self._res_block(n, p + "/res_block/0")
        return self._res_block(n, p + "/res_block/1", relu_after=True)

    def _level(self, x, lv):
        p = "FISRnet/" + lv
        n, s0 = self._enc(x, p + "/enc/level_0")
        n, s1 = self._enc(n, p + "/enc/level_1")
        n, s2 = self._enc(n, p + "/enc/level_2")
        n = self.conv(p + "/bottleneck/conv/0", n)
        n = self._res_block(n, p + "/bottleneck/res_block/0", relu_after=True)
        n = self._dec(n, s2, p + "/dec/level_2")
        n = self._dec(n, s1, p + "/dec/level_1")
        n = self._dec(n, s0, p + "/dec/level_0")
        b, h, w, _ = n.shape
        pred = self.zeros(b, 2 * h, 2 * w, 9)
        for head, sc in (("FI-SR", (pred, 0, 3, 3)), ("SR", (pred, 3, 1 << 30, 0))):     # [fr1, SR, fr2], FISRnet.py:107-108
            a = self.conv(f"{p}/{head}/conv/0", n)
            a = self._res_block(a, f"{p}/{head}/res_block/0")
            a = self.conv(f"{p}/{head}/conv/1", a, flags=RELU_IN | RELU_OUT | D2S)
            self.conv(f"{p}/{head}/conv/2", a, scatter=sc)
        return pred

    def model(self, x29):
        """x29 [B,H,W,29] device tensor -> (pred_l1, pred_l2, pred_l3), recorded on the tape."""
        p1 = self._level(self.level_input(x29[:, ::4, ::4].contiguous(), None), "level_1")          # FISRnet.py:81
        p2 = self._level(self.level_input(x29[:, ::2, ::2].contiguous(), p1), "level_2")           # :112-113
        p3 = self._level(self.level_input(x29, p2), "level_3")                                     # :144
        return p1, p2, p3

    # ------------------------------------------------------------------ backward
    def _before_write(self, t):
        """An in-place write to `t` on the main stream: a weight-gradient launch on the side stream may still be reading it."""
        ev = getattr(t, "_fisr_side_ev", None)
        if ev is not None:
            self.torch.cuda.current_stream(self.device).wait_event(ev)
            t._fisr_side_ev = None

    def _acc(self, grads, t, g):
        k = t.data_ptr()
        if k in grads:
            self._before_write(grads[k])
            self._ck(self.L.fisr_train_axpy(self._p(g), 1.0, self._p(grads[k]), g.numel(), self._st()))
        else:
            grads[k] = g

    def _wgrad(self, c, x0, c0, x1, c1, relu_in, g, n, h, w):
        """fisr_train_wgrad on the side stream (after everything the main stream has queued so far); the tensors it reads are
        kept from reuse (record_stream) and from in-place writes (_before_write) until it is done."""
        torch, L = self.torch, self.L
        args = (self._p(x0), c0, self._p(x1), c1, relu_in, self._p(g), g.shape[3], self._p(c.gw), self._p(c.gb), c.ci, c.co, n, h, w)
        if not self.overlap_wgrad:
            self._ck(L.fisr_train_wgrad(*args, self._st()))
            return
        main, side = torch.cuda.current_stream(self.device), self._side
        side.wait_stream(main)
        self._ck(L.fisr_train_wgrad(*args, ctypes.c_void_p(side.cuda_stream)))
        ev = torch.cuda.Event()
        ev.record(side)
        for t in (x0, x1, g):
            if t is not None:
                t.record_stream(side)
        g._fisr_side_ev = ev

    def backward(self, pred_grads):
        """pred_grads: {prediction tensor: its gradient}.  Walks the tape in reverse; weight / bias gradients ACCUMULATE
        into conv.gw / conv.gb (four passes share the weights)."""
        torch, L = self.torch, self.L
        grads = {t.data_ptr(): g for t, g in pred_grads}
        for ent in reversed(self.tape):
            kind = ent[0]
            if kind == "conv":
                _, c, x0, x1, res, y, flags, scatter = ent
                n, h, w, c0 = x0.shape
                c1 = x1.shape[3] if x1 is not None else 0
                if scatter is not None:
                    gp = grads.get(y.data_ptr())
                    if gp is None:
                        continue
                    _, coff, split, gap = scatter
                    g = self.zeros(n, h, w, 16)                                    # the head's channels, padded to 16
                    npix = n * h * w
                    if split < c.co:
                        self._ck(L.fisr_train_copy_channels(self._p(gp), 9, coff, self._p(g), 16, 0, split, npix, 0, self._st()))
                        self._ck(L.fisr_train_copy_channels(self._p(gp), 9, coff + split + gap, self._p(g), 16, split, c.co - split, npix, 0, self._st()))
                    else:
                        self._ck(L.fisr_train_copy_channels(self._p(gp), 9, coff, self._p(g), 16, 0, c.co, npix, 0, self._st()))
                else:
                    g = grads.pop(y.data_ptr(), None)
                    if g is None:
                        continue
                    if flags & RELU_OUT:
                        self._before_write(g)
                        self._ck(L.fisr_train_relu_bwd(self._p(g), self._p(y), self._p(g), g.numel(), self._st()))
                    if flags & D2S:
                        g_lr = self.new(n, h, w, c.co)
                        self._ck(L.fisr_train_s2d(self._p(g), self._p(g_lr), n, h, w, c.co // 4, self._st()))
                        g = g_lr
                    if res is not None:
                        self._acc(grads, res, g)
                cg = g.shape[3]
                self._wgrad(c, x0, c0, x1, c1, 1 if flags & RELU_IN else 0, g, n, h, w)                         # weight + bias gradient
                if not getattr(x0, "_fisr_needs_grad", True):
                    continue
                # data gradient: the same conv with rotated taps, input g (cg channels, cg % 16 == 0), output c0 + c1 channels
                dx = self.new(n, h, w, c0 + c1)
                self._ck(L.fisr_train_conv3x3(self._p(g), cg, None, 0, self._p(c.pk_t), self._p(self.zero_bias), c0 + c1,
                                              None, self._p(dx), n, h, w, 0, 0, 0, 0, 0, self._p(c.pkw_t), self._st()))
                if c1:
                    d0, d1 = self.new(n, h, w, c0), self.new(n, h, w, c1)
                    npix = n * h * w
                    self._ck(L.fisr_train_copy_channels(self._p(dx), c0 + c1, 0, self._p(d0), c0, 0, c0, npix, 0, self._st()))
                    self._ck(L.fisr_train_copy_channels(self._p(dx), c0 + c1, c0, self._p(d1), c1, 0, c1, npix, 0, self._st()))
                    parts = ((x0, d0), (x1, d1))
                else:
                    parts = ((x0, dx),)
                for xs, ds in parts:
                    if flags & RELU_IN:
                        self._ck(L.fisr_train_relu_bwd(self._p(ds), self._p(xs), self._p(ds), ds.numel(), self._st()))
                    self._acc(grads, xs, ds)
            elif kind == "pool":
                _, x, y = ent
                g = grads.pop(y.data_ptr(), None)
                if g is None:
                    continue
                n, h, w, c = x.shape
                dx = self.new(n, h, w, c)
                self._ck(L.fisr_train_maxpool2_bwd(self._p(x), self._p(g), self._p(dx), n, h, w, c, self._st()))
                self._acc(grads, x, dx)
            elif kind == "up":
                _, x, y = ent
                g = grads.pop(y.data_ptr(), None)
                if g is None:
                    continue
                n, h, w, c = x.shape
                dx = self.new(n, h, w, c)
                self._ck(L.fisr_train_upsample2_bwd(self._p(g), self._p(dx), n, h, w, c, self._st()))
                self._acc(grads, x, dx)
            elif kind == "cat_pred":
                _, prev, t = ent
                g = grads.pop(t.data_ptr(), None)
                if g is None:
                    continue
                n, h, w, cp = t.shape
                gp = grads.get(prev.data_ptr())
                if gp is None:
                    gp = self.zeros(*prev.shape)
                    grads[prev.data_ptr()] = gp
                self._ck(L.fisr_train_copy_channels(self._p(g), cp, 29, self._p(gp), 9, 0, 9, n * h * w, 1, self._st()))
        torch.cuda.current_stream(self.device).wait_stream(self._side)       # the weight gradients are complete from here on
        self.tape = []

    # ------------------------------------------------------------------ the training step (FISRnet.py:283-491)
    @staticmethod
    def window_input(batch, order):
        """Tensor_slicer_recurrent* + concat (ops.py:92-117, FISRnet.py:287-290)."""
        import torch
        return torch.cat([batch["data15"][..., 3 * order:3 * order + 9], batch["flow16"][..., 4 * order:4 * order + 8],
                          batch["warp24"][..., 6 * order:6 * order + 12]], dim=3).contiguous()

    @staticmethod
    def stride2_input(batch):
        """FISRnet.py:394-401: frames 0, 2, 4 with the stride-2 flows / warps."""
        import torch
        d = batch["data15"]
        return torch.cat([d[..., 0:3], d[..., 6:9], d[..., 12:15], batch["flow_ss2"], batch["warp_ss2"]], dim=3).contiguous()

    def loss_and_grads(self, batch):
        """batch: dict of device NHWC fp32 tensors data15 [B,H,W,15], label21 [B,2H,2W,21], flow16, warp24, flow_ss2 [.,8],
        warp_ss2 [.,12].  Runs the four passes, the loss kernel per level and the whole backward.  Returns (total, terms);
        gradients are left accumulated in the conv states (call zero_grad() first)."""
        torch, L = self.torch, self.L
        self.tape = []
        # The four passes share the weights, so they run as ONE pass over a batch of 4B: windows 0, 1, 2 (stride 1,
        # FISRnet.py:283-310) and the stride-2 window (:403-409) stacked along the batch axis -- a quarter of the kernel
        # launches, four times the work per launch, the same numbers (every sample is convolved independently).
        b0 = batch["data15"].shape[0]
        x_all = torch.cat([self.window_input(batch, k) for k in range(3)] + [self.stride2_input(batch)], dim=0)
        pred_all = self.model(x_all)
        label = batch["label21"]
        gts = (label[:, ::4, ::4].contiguous(), label[:, ::2, ::2].contiguous(), label)   # FISRnet.py:262-263 (see the oracle)
        lam = self.lam
        terms = np.zeros(7)
        pred_grads = []
        for li, lv in enumerate(_weights.LEVELS):
            pa_t = pred_all[li]                                      # [4B, h, w, 9]: chunk k = window k
            _, h, w, _ = pa_t.shape
            npix = b0 * h * w
            chunk = npix * 9 * 4                                     # bytes per window
            n1, n3, s = npix * 3.0, npix * 9.0, LEVEL_SCALE[lv]
            k7 = np.array([lam["recn"] * s * 2 / n3, lam["tm1"] * s * 2 / n1, lam["tmm"] * s * 2 / n1, lam["td"] * s * 2 / n1,
                           lam["ss2"] * lam["recn"] * s * 2 / n3, lam["ss2"] * lam["td"] * s * 2 / n1,
                           lam["ss2"] * lam["tm2"] * s * 2 / n3], dtype=np.float32) * np.float32(self.grad_scale)
            g_all = self.new(4 * b0, h, w, 9)
            sums = self.zeros(8)
            pa = (ctypes.c_void_p * 4)(*[pa_t.data_ptr() + k * chunk for k in range(4)])
            ga = (ctypes.c_void_p * 4)(*[g_all.data_ptr() + k * chunk for k in range(4)])
            self._ck(L.fisr_train_loss(pa, self._p(gts[li]), ga, self._p(sums), npix,
                                       k7.ctypes.data_as(ctypes.POINTER(ctypes.c_float)), self._st()))
            sv = sums.cpu().numpy().astype(np.float64)
            terms += s * np.array([sv[0] / n3, sv[1] / n1, sv[2] / n1, sv[3] / n1, sv[4] / n3, sv[5] / n1, sv[6] / n3])
            pred_grads.append((pa_t, g_all))
        preds = [tuple(p[k * b0:(k + 1) * b0] for p in pred_all) for k in range(4)]     # per window, as the reference names them
        total = (lam["recn"] * terms[0] + lam["tm1"] * terms[1] + lam["tmm"] * terms[2] + lam["td"] * terms[3]
                 + lam["ss2"] * (lam["recn"] * terms[4] + lam["td"] * terms[5] + lam["tm2"] * terms[6]))
        self.last_preds = preds if self.keep_preds else None
        self.backward(pred_grads)
        names = ("recn", "tm", "tmm", "td", "recn_ss2", "td_ss2", "tm_ss2")
        return float(total), dict(zip(names, terms.tolist()))

    def adam_step(self, lr, b1=0.9, b2=0.999, eps=1e-8):
        """tf.train.AdamOptimizer.apply_gradients (TF 1.13): lr_t = lr * sqrt(1 - b2^t) / (1 - b1^t)."""
        self.step_count += 1
        t = self.step_count
        lr_t = lr * math.sqrt(1.0 - b2 ** t) / (1.0 - b1 ** t)
        self._ck(self.L.fisr_train_adam(self._p(self.wflat), self._p(self.gflat), self._p(self.mflat), self._p(self.vflat), self.wflat.numel(),
                                        lr_t, b1, b2, eps, self._st()))                  # every tensor at once (the padding stays 0)
        self.repack()

    def train_step(self, batch, lr, group=None):
        self.zero_grad()
        total, terms = self.loss_and_grads(batch)
        self.allreduce_grads(group)
        self.adam_step(lr)
        return total, terms


def to_device_batch(batch_np, device="cuda:0"):
    import torch
    return {k: torch.from_numpy(np.ascontiguousarray(v, dtype=np.float32)).to(device) for k, v in batch_np.items()}
