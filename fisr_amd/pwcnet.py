"""On-GPU optical flow for `--phase FISR_for_video`: host-side mirror of the reference's flow step.

  PWCNet.compute_flow(frames)  <->  FISR_for_video_Compute_Flow(args)
                                    (FISR_tfoptflow/FISR_for_video_pwcnet_predict_from_img_test.py:84-147)
  PWCNet.load(ckpt_prefix)     <->  ModelBase.load_ckpt of `pwcnet.ckpt-595000` (script :31, model_base.py:142-191)

All arithmetic runs in libfisr_hip.so (pwc_kernels.h behind the `fisr_pwc_*` C-ABI, include/fisr.h); PyTorch only
owns device memory and the stream.  No CPU fallback.  The PWC-Net checkpoint is not part of the reference tree: load
one with `load()` (TF checkpoint-V2 bundle or .npz keyed by the TF variable names) or use `synthetic_weights()`.
"""
from __future__ import annotations

import ctypes
from collections import OrderedDict

import numpy as np

from . import lib as _lib
from .lib import FisrError


def variable_shapes() -> "OrderedDict[str, tuple]":
    """The 182 variables of the inference graph, as the C library enumerates them (TF names and shapes)."""
    L = _lib.lib()
    out = OrderedDict()
    name = ctypes.c_char_p()
    shape = (ctypes.c_int64 * 4)()
    for i in range(L.fisr_pwc_num_variables()):
        rank = L.fisr_pwc_variable(i, ctypes.byref(name), shape)
        out[name.value.decode()] = tuple(int(shape[k]) for k in range(rank))
    return out


def synthetic_weights(seed: int = 595000, flow_gain: float = 1.0) -> "OrderedDict[str, np.ndarray]":
    """Seeded stand-in weights (the reference's `pwcnet-lg-6-2-multisteps-chairsthingsmix` checkpoint is not in
    its tree): He-normal kernels as the reference initialises them (model_pwcnet.py:1085), small biases; the flow
    heads and the up-sampling kernels are scaled down so the synthetic flows stay within a few pixels per level."""
    rng = np.random.default_rng(seed)
    out = OrderedDict()
    for name, shape in variable_shapes().items():
        if name.endswith("/bias"):
            out[name] = (rng.standard_normal(shape) * 0.01).astype(np.float32)
            continue
        fan_in = shape[0] * shape[1] * (shape[3] if "upsample" in name else shape[2])
        std = np.sqrt(2.0 / fan_in)
        if "/flow" in name or (name.endswith("7/kernel") and "dc_conv" in name):
            std *= 0.5 * flow_gain
        if "upsample" in name:
            std = 0.25 * np.sqrt(1.0 / fan_in)
        out[name] = (rng.standard_normal(shape) * std).astype(np.float32)
    return out


class PWCNet:
    """precision "fp32" (float32 tensors and arithmetic; r04: the dense and context layers on the F(4x4,3x3) Winograd kernel where
    the (sub-)image is at least 48 x 64, F(2x2,3x3) below), "fp32w" (the same with F(2x2) everywhere: round 3's fp32 flow) or "fp16"
    (fp16 feature tensors, fp32 accumulation, float32 flows: the 16-bit flow of cfg5)."""

    def __init__(self, device: str = "cuda:0", precision: str = "fp32"):
        import torch
        if precision not in ("fp32", "fp32w", "fp16"):
            raise ValueError("PWCNet precision must be 'fp32', 'fp32w' or 'fp16'")
        self.precision = precision
        self.device = torch.device(device)
        if self.device.type != "cuda" or not torch.cuda.is_available():
            raise FisrError("PWCNet needs a ROCm GPU (cuda device); there is no CPU fallback")
        self._L = _lib.lib()
        self._ctx = ctypes.c_void_p()
        self._check(self._L.fisr_pwc_create(ctypes.byref(self._ctx), self.device.index or 0))
        self._finalized = False
        self._ws = None

    def _check(self, rc):
        if rc < 0:
            raise FisrError(f"libfisr_hip (pwc) error {rc}: {self._L.fisr_pwc_last_error(self._ctx).decode('utf-8', 'replace')}")
        return rc

    def set_weights(self, weights) -> None:
        shapes = variable_shapes()
        for name, shape in shapes.items():
            if name not in weights:
                raise KeyError(f"missing variable {name}")
            if tuple(weights[name].shape) != tuple(shape):
                raise ValueError(f"{name}: shape {tuple(weights[name].shape)} != {shape}")
        for name in shapes:
            a = np.ascontiguousarray(weights[name], np.float32)
            shp = (ctypes.c_int64 * max(a.ndim, 1))(*a.shape)
            self._check(self._L.fisr_pwc_set_weight(self._ctx, name.encode(), a.ctypes.data_as(ctypes.POINTER(ctypes.c_float)), shp, a.ndim))
        self._check(self._L.fisr_pwc_finalize_precision(self._ctx, {"fp16": _lib.PREC_F16, "fp32w": _lib.PREC_F32W}.get(self.precision, _lib.PREC_F32W4)))
        self._finalized = True

    def load(self, path_or_prefix: str) -> None:
        """`.npz` keyed by TF variable names, or a TF checkpoint-V2 bundle prefix (`pwcnet.ckpt-595000`)."""
        if path_or_prefix.endswith(".npz"):
            with np.load(path_or_prefix) as z:
                w = {k: z[k] for k in z.files}
        else:
            from . import tf_bundle
            w = tf_bundle.read_bundle(path_or_prefix, name_filter="pwcnet")
        self.set_weights(w)

    def _workspace(self, need: int):
        import torch
        if need == 0:
            raise FisrError("pwc workspace query failed (weights not finalized or bad shape)")
        if self._ws is None or self._ws.numel() < need:
            self._ws = None
            self._ws = torch.empty(need, dtype=torch.uint8, device=self.device)
        return self._ws

    def _stream(self):
        import torch
        return ctypes.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    def nn(self, im_pair, want_pyramid: bool = False):
        """model_pwcnet.py:1525-1593 on a prepared pair [2,H,W,4] (device float32) -> flow_pred [2,H,W,2]
        (and the refined pyramid flows [[lvl6..lvl2] for a->b, [...] for b->a])."""
        import torch
        im = im_pair.to(device=self.device, dtype=torch.float32).contiguous()
        _, H, W, c = im.shape
        if c != 4 or im.shape[0] != 2:
            raise ValueError("im_pair must be [2,H,W,4]")
        ws = self._workspace(self._L.fisr_pwc_nn_workspace_bytes(self._ctx, H, W))
        out = torch.empty((2, H, W, 2), dtype=torch.float32, device=self.device)
        pyr, ptrs = None, None
        if want_pyramid:
            pyr = [[torch.empty((H >> l, W >> l, 2), dtype=torch.float32, device=self.device) for l in range(6, 1, -1)] for _ in range(2)]
            ptrs = (ctypes.c_void_p * 10)(*[t.data_ptr() for d in pyr for t in d])
        self._check(self._L.fisr_pwc_nn(self._ctx, ctypes.c_void_p(im.data_ptr()), H, W, ctypes.c_void_p(out.data_ptr()), ptrs,
                                        ctypes.c_void_p(ws.data_ptr()), ws.numel(), self._stream()))
        return (out, pyr) if want_pyramid else out

    def flow_pair(self, yuv_a, yuv_b):
        """Two YUV uint8 frames [h,w,3] -> (flow a->b, flow b->a), [h,w,2] float32 LR pixels on the device."""
        import torch
        a = yuv_a.to(device=self.device, dtype=torch.uint8).contiguous()
        b = yuv_b.to(device=self.device, dtype=torch.uint8).contiguous()
        if a.shape != b.shape or a.dim() != 3 or a.shape[2] != 3:
            raise ValueError("frames must be [h,w,3] uint8 of equal size")
        h, w = a.shape[:2]
        ws = self._workspace(self._L.fisr_pwc_flow_workspace_bytes(self._ctx, h, w))
        fab = torch.empty((h, w, 2), dtype=torch.float32, device=self.device)
        fba = torch.empty((h, w, 2), dtype=torch.float32, device=self.device)
        self._check(self._L.fisr_pwc_flow_pair(self._ctx, ctypes.c_void_p(a.data_ptr()), ctypes.c_void_p(b.data_ptr()), h, w,
                                               ctypes.c_void_p(fab.data_ptr()), ctypes.c_void_p(fba.data_ptr()),
                                               ctypes.c_void_p(ws.data_ptr()), ws.numel(), self._stream()))
        return fab, fba

    def flow_stack(self, frames_u8, out=None):
        """One call for a run of frames (list of [h,w,3] uint8 YUV device tensors, 2 .. 9 of them): every frame is pre-processed
        and its feature pyramid extracted once, all 2 (n-1) directions go through the decoder as batches of up to 8 ->
        [n-1, 2, h, w, 2] float32 on the device (pair fr: [0] = fr -> fr+1, [1] = fr+1 -> fr)."""
        import torch
        fr = [f.to(device=self.device, dtype=torch.uint8).contiguous() for f in frames_u8]
        n = len(fr)
        if n < 2 or any(f.shape != fr[0].shape or f.dim() != 3 or f.shape[2] != 3 for f in fr):
            raise ValueError("frames must be two or more [h,w,3] uint8 tensors of equal size")
        h, w = fr[0].shape[:2]
        ws = self._workspace(self._L.fisr_pwc_flow_stack_workspace_bytes(self._ctx, n, h, w))
        if out is None:
            out = torch.empty((n - 1, 2, h, w, 2), dtype=torch.float32, device=self.device)
        ptrs = (ctypes.c_void_p * n)(*[f.data_ptr() for f in fr])
        self._check(self._L.fisr_pwc_flow_stack(self._ctx, ptrs, n, h, w, ctypes.c_void_p(out.data_ptr()),
                                                ctypes.c_void_p(ws.data_ptr()), ws.numel(), self._stream()))
        return out

    def compute_flow(self, frames_u8, chunk: int = 5):
        """script :104-141: frames (list of [h,w,3] uint8 YUV, device or host tensors / arrays) ->
        pred [num_fr-1, 2, h, w, 2] float32 on the device (pair fr: [0] = fr -> fr+1, [1] = fr+1 -> fr).  Runs of `chunk`
        frames per call (neighbouring runs share one frame)."""
        import torch
        fr = [torch.as_tensor(f).to(self.device) for f in frames_u8]
        out = []
        k = 0
        while k < len(fr) - 1:
            run = fr[k:k + chunk]
            out.append(self.flow_stack(run))
            k += len(run) - 1
            print("Processing for computing flows [%5d/%5d]" % (k, len(fr)))
        return torch.cat(out, dim=0)

    def close(self):
        if getattr(self, "_ctx", None) is not None and self._ctx.value:
            self._L.fisr_pwc_destroy(self._ctx)
            self._ctx = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
