"""Host-side mirror of the reference's `FISRnet` class for the inference path.

Same names, argument meaning and error behaviour as the reference seams:

  FISRnet.model(img, sf)            <-> FISRnet.py:73-173   (graph seam, via libfisr_hip.so)
  FISRnet.load(checkpoint_dir)      <-> FISRnet.py:1101-1115 (weight seam)
  FISRnet.test()                    <-> FISRnet.py:746-935
  FISRnet.FISR_for_video(flo, mat)  <-> FISRnet.py:937-1084

PyTorch is used only for device memory, streams and torch.distributed; all arithmetic of the
hot path runs in the hand-written HIP kernels behind the C-ABI (include/fisr.h).  There is no
CPU fallback: without a GPU / the built library every compute call raises FisrError.
"""
from __future__ import annotations

import ctypes
import glob
import math
import os
import time
from types import SimpleNamespace
from typing import Optional, Sequence, Tuple

import numpy as np

from . import lib as _lib
from . import tiling
from . import weights as _weights
from .lib import FisrError

# "fp32" is the fp32 engine as it ships: fp32 tensors, fp32 arithmetic, Winograd minimal filtering for the 132 convs with
# Cout % 64 == 0 (what cuDNN does for the reference's TF 1.13) -- F(4x4,3x3) on the maps where it is the faster kernel (r03:
# conv3x3_wf4.h, "fp32w4" names it explicitly), F(2x2,3x3) elsewhere; "fp32w" is the all-F(2x2) engine of round 2; "fp32d" the
# same engine with the direct (exact fmaf-chain) MFMA kernel for every conv.
_PREC = {"fp32": _lib.PREC_F32W4, "f32": _lib.PREC_F32W4, "float32": _lib.PREC_F32W4, "fp32w": _lib.PREC_F32W,
         "fp32d": _lib.PREC_F32, "fp32w4": _lib.PREC_F32W4,      # F(4x4,3x3) Winograd for the large maps (conv3x3_wf4.h)
         "fp16": _lib.PREC_F16, "f16": _lib.PREC_F16, "float16": _lib.PREC_F16,
         "bf16x3": _lib.PREC_BF16X3, "f16f8": _lib.PREC_F16F8, "mixed": _lib.PREC_MIXED,
         "fp16r": _lib.PREC_F16R, "mixedr": _lib.PREC_MIXEDR,          # round 1's register-staged fp16 kernel (A/B runs)
         "f16f8r": _lib.PREC_F16F8R}                                   # ... and its f16f8 form (r05: "f16f8" runs conv3x3_dma_fs.h)


# ONE default arithmetic for every entry point (FISRnet(), main.py, bench.py): the reference computes in
# fp32 (cfg2 of BASELINE.json), so the default is the fp32 engine; the split-precision modes are opt-in.
DEFAULT_PRECISION = "fp32"
PRECISIONS = ("fp32", "fp32w4", "fp32d", "bf16x3", "f16f8", "mixed", "fp16")     # CLI names: the shipped engines ("fp32d": the exact direct-conv reference)
# A/B engines of the diagnostics build (lib.build(diag=True), loaded through FISR_HIP_SO): superseded kernels everywhere.  The product
# library refuses them in fisr_finalize_weights (FISR_EINVAL); FISRnet(precision=...) still maps the names for those runs.
DIAG_PRECISIONS = ("fp32w", "fp16r", "mixedr", "f16f8r")


def _torch():
    import torch
    return torch


def _ptr(t) -> ctypes.c_void_p:
    return ctypes.c_void_p(t.data_ptr() if t is not None else 0)


def _stream(device) -> ctypes.c_void_p:
    torch = _torch()
    return ctypes.c_void_p(torch.cuda.current_stream(device).cuda_stream)


def default_args(**over) -> SimpleNamespace:
    """The reference's argparse defaults that the inference path reads (main.py:26-103)."""
    a = SimpleNamespace(
        net_type="FISRnet", phase="FISR_for_video", scale_factor=2, exp_num=1,
        test_data_path="./data/test/LR_LFR", test_label_path="./data/test/HR_HFR",
        test_flow_data_path="./data/test/flow/LR_Surfing_SlamDunk_test_ss1.flo",
        test_warped_data_path="./data/test/warped/LR_Surfing_SlamDunk_test_ss1_warp.mat",
        test_img_dir="./test_img_dir", checkpoint_dir="./checkpoint_dir",
        test_patch=(2, 2), test_input_size=(1080, 1920),
        frame_folder_path="./FISR_test_folder/scene1", FISR_input_size=(1080, 1920), frame_num=5,
        FISR_test_patch=(2, 2), precision=DEFAULT_PRECISION, device="cuda:0", batch_tiles=True)
    a.__dict__.update(over)
    return a


def fit_pack_inputs(fr, fl, wp, h: int, w: int):
    """Shape contract of `pack_input`: the kernel indexes all 11 tensors with ONE row stride (that of frame
    0), whereas the reference slices every array independently with [:h,:w] (FISRnet.py:828-843) and raises
    when one is too small.  So: every tensor must be [H,W,C] with the right C and at least the crop size;
    tensors larger than frame 0 are cropped (top-left) to frame 0's size; smaller ones are an error.
    Returns (frames, flows, warps, h0, w0)."""
    h0, w0 = fr[0].shape[:2]
    if h > h0 or w > w0:
        raise ValueError(f"pack_input: crop {h}x{w} exceeds the frame size {h0}x{w0}")

    def fit(t, c, what):
        if t.dim() != 3 or t.shape[2] != c:
            raise ValueError(f"pack_input: {what} must be [h,w,{c}], got {tuple(t.shape)}")
        if t.shape[0] < h or t.shape[1] < w:
            raise ValueError(f"pack_input: {what} is {t.shape[0]}x{t.shape[1]}, smaller than the crop {h}x{w}")
        if t.shape[0] < h0 or t.shape[1] < w0:
            raise ValueError(f"pack_input: {what} is {t.shape[0]}x{t.shape[1]} but frame 0 is {h0}x{w0}: "
                             "crop the inputs to a common size first")
        return t if tuple(t.shape[:2]) == (h0, w0) else t[:h0, :w0].contiguous()

    return ([fit(f, 3, f"frame {i}") for i, f in enumerate(fr)], [fit(f, 2, f"flow {i}") for i, f in enumerate(fl)],
            [fit(f, 3, f"warp {i}") for i, f in enumerate(wp)], h0, w0)


class FISRnet:
    """MI355X-native FISRnet inference engine with the reference class's surface."""

    def __init__(self, args: Optional[SimpleNamespace] = None, device: Optional[str] = None,
                 precision: Optional[str] = None, quiet: bool = True):
        args = args or default_args()
        self.args = args
        self.model_name = getattr(args, "net_type", "FISRnet")      # FISRnet.py:19
        self.exp_num = getattr(args, "exp_num", 1)
        self.scale_factor = int(getattr(args, "scale_factor", 2))   # main.py:29 (float default quirk fixed)
        if self.scale_factor != 2:
            raise ValueError("FISRnet is a x2 network (depth_to_space(.,2), FISRnet.py:99)")
        for k in ("test_data_path", "test_label_path", "test_flow_data_path", "test_warped_data_path",
                  "test_img_dir", "checkpoint_dir", "test_patch", "test_input_size", "frame_folder_path",
                  "FISR_input_size", "frame_num", "FISR_test_patch"):
            setattr(self, k, getattr(args, k, getattr(default_args(), k)))
        self.precision = precision or getattr(args, "precision", DEFAULT_PRECISION)
        self.batch_tiles = bool(getattr(args, "batch_tiles", True))
        if self.precision not in _PREC:
            raise ValueError(f"unknown precision {self.precision!r}")
        torch = _torch()
        self.device = torch.device(device or getattr(args, "device", None) or "cuda:0")
        if self.device.type != "cuda" or not torch.cuda.is_available():
            raise FisrError("FISRnet needs a ROCm GPU (cuda device); there is no CPU fallback")
        self._L = _lib.lib()
        self._ctx = ctypes.c_void_p()
        _lib.check(self._L.fisr_create(ctypes.byref(self._ctx), self.device.index or 0))
        self._finalized = False
        self._ws = None
        self.quiet = quiet
        self.inf_time = []

    # ------------------------------------------------------------------ weights
    @property
    def model_dir(self) -> str:
        return "{}_exp{}".format(self.model_name, self.exp_num)   # FISRnet.py:1086-1089

    def set_weights(self, weights) -> None:
        """Install the 276 variables (dict keyed by TF variable name) and re-pack for the kernels."""
        _weights.check_complete(weights)
        for name, arr in weights.items():
            a = np.ascontiguousarray(arr, np.float32)
            shape = (ctypes.c_int64 * a.ndim)(*a.shape)
            _lib.check(self._L.fisr_set_weight(self._ctx, name.encode(), a.ctypes.data_as(ctypes.POINTER(ctypes.c_float)),
                                               shape, a.ndim), self._ctx)
        _lib.check(self._L.fisr_finalize_weights(self._ctx, _PREC[self.precision]), self._ctx)
        self._finalized = True

    def load(self, checkpoint_dir: str):
        """FISRnet.py:1101-1115: -> (ok, step).  Reads `<checkpoint_dir>/FISRnet_exp<N>/`."""
        print(" [*] Reading checkpoints...")
        path, kind, step = _weights.find_checkpoint(checkpoint_dir, self.model_dir)
        if path is None:
            print(" [*] Failed to find a checkpoint")
            return False, 0
        self.set_weights(_weights.load_weights(path, kind))
        print(" [*] Success to read {}".format(os.path.basename(path)))
        return True, step

    # ------------------------------------------------------------------ graph seam
    def _workspace(self, n: int, h: int, w: int):
        torch = _torch()
        need = self._L.fisr_workspace_bytes(self._ctx, n, h, w)
        if need == 0:
            raise FisrError("fisr_workspace_bytes: invalid shape or weights not finalized")
        if self._ws is None or self._ws.numel() < need:
            self._ws = None
            self._ws = torch.empty(need, dtype=torch.uint8, device=self.device)
        return self._ws

    def model(self, img, sf: int = 2, reuse: bool = False, scope: str = "FISRnet", want_all: bool = True):
        """FISRnet.py:73-173: img [N,H,W,29] float32 on the GPU -> (pred_l1, pred_l2, pred_l3)."""
        torch = _torch()
        if sf != 2:
            raise ValueError("sf must be 2")
        if not self._finalized:
            raise FisrError("weights not loaded: call load() or set_weights() first")
        if img.dim() != 4 or img.shape[3] != 29:
            raise ValueError(f"img must be [N,H,W,29], got {tuple(img.shape)}")
        n, h, w, _ = img.shape
        if h % 32 or w % 32:
            raise ValueError("H and W must be multiples of 32 (FISRnet.py:820-824)")
        img = img.to(device=self.device, dtype=torch.float32).contiguous()
        l3 = torch.empty((n, 2 * h, 2 * w, 9), dtype=torch.float32, device=self.device)
        l2 = torch.empty((n, h, w, 9), dtype=torch.float32, device=self.device) if want_all else None
        l1 = torch.empty((n, h // 2, w // 2, 9), dtype=torch.float32, device=self.device) if want_all else None
        ws = self._workspace(n, h, w)
        _lib.check(self._L.fisr_forward(self._ctx, _ptr(img), n, h, w, _ptr(l3), _ptr(l2), _ptr(l1), _ptr(ws),
                                        ws.numel(), _stream(self.device)), self._ctx)
        return l1, l2, l3

    def capture(self, n: int, h: int, w: int, want_all: bool = True):
        """One forward of a fixed shape captured into a HIP graph: returns (graph, static_input,
        (pred_l1, pred_l2, pred_l3)).  (Measured on ROCm 7.2 / MI355X: replay of the ~300 kernel nodes is
        not faster than the eager launches at 96x96, so nothing in the package depends on it.)
        Copy a new input into `static_input`, `graph.replay()`, read the static outputs.  fisr_forward is
        capture-safe: no allocation, synchronisation or host read-back happens inside it."""
        torch = _torch()
        static_in = torch.zeros((n, h, w, 29), dtype=torch.float32, device=self.device)
        self._workspace(n, h, w)                       # allocated outside the capture, kept by the ctx mirror
        side = torch.cuda.Stream(device=self.device)
        side.wait_stream(torch.cuda.current_stream(self.device))
        with torch.cuda.stream(side):                  # warm-up run (sets kernel attributes) on the side stream
            self.model(static_in, want_all=want_all)
        torch.cuda.current_stream(self.device).wait_stream(side)
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            outs = self.model(static_in, want_all=want_all)
        return graph, static_in, outs

    def engine_description(self) -> str:
        """One line for logs / bench.py: library version and the arithmetic of this engine."""
        return f"{self._L.fisr_version().decode()} precision={self.precision}"

    # ------------------------------------------------------------------ profiling hooks
    def profile(self, on) -> None:
        """on: 0/False off, 1/True per kernel class, 2 per layer."""
        self._L.fisr_profile_enable(self._ctx, int(on))
        self._L.fisr_profile_reset(self._ctx)

    def profile_read(self):
        cap = 1024
        names = (ctypes.c_char_p * cap)()
        ms = (ctypes.c_double * cap)()
        cnt = (ctypes.c_int64 * cap)()
        fl = (ctypes.c_double * cap)()
        by = (ctypes.c_double * cap)()
        k = self._L.fisr_profile_read(self._ctx, cap, names, ms, cnt, fl, by)
        return [dict(name=names[i].decode(), ms=ms[i], launches=cnt[i], flops=fl[i], bytes=by[i]) for i in range(min(k, cap))]

    # ------------------------------------------------------------------ glue kernels
    def warp(self, src_yuv, flow, flow_scale: float = 0.5, quantized: bool = True):
        """warp script :112-129, one direction: src_yuv [h,w,3] (uint8/float, 0..255), flow [h,w,2] px."""
        torch = _torch()
        src = src_yuv.to(device=self.device, dtype=torch.float32).contiguous()
        fl = flow.to(device=self.device, dtype=torch.float32).contiguous()
        h, w = fl.shape[:2]
        if tuple(src.shape) != (h, w, 3) or fl.shape[2] != 2:
            raise ValueError("warp: src must be [h,w,3] and flow [h,w,2]")
        dst = torch.empty((h, w, 3), dtype=torch.float32, device=self.device)
        _lib.check(self._L.fisr_warp(_ptr(src), _ptr(fl), flow_scale, h, w, int(quantized), _ptr(dst), _stream(self.device)))
        return dst

    def pack_input(self, frames_u8: Sequence, flows: Sequence, warps: Sequence, h: int, w: int, out=None):
        """FISRnet.py:828-843: 3 uint8 YUV frames [h0,w0,3], 4 flows [h0,w0,2] (px), 4 warped frames
        [h0,w0,3] (0..255 float32) -> [1,h,w,29] float32 (top-left crop to h x w).  `out`: an existing contiguous
        float32 [1,h,w,29] (or [h,w,29]) device tensor to fill instead of a new one (e.g. one window of a batch)."""
        torch = _torch()
        fr = [f.to(device=self.device, dtype=torch.uint8).contiguous() for f in frames_u8]
        fl = [f.to(device=self.device, dtype=torch.float32).contiguous() for f in flows]
        wp = [f.to(device=self.device, dtype=torch.float32).contiguous() for f in warps]
        if len(fr) != 3 or len(fl) != 4 or len(wp) != 4:
            raise ValueError("pack_input needs 3 frames, 4 flows, 4 warps")
        fr, fl, wp, h0, w0 = fit_pack_inputs(fr, fl, wp, h, w)
        if out is None:
            out = torch.empty((1, h, w, 29), dtype=torch.float32, device=self.device)
        elif (tuple(out.shape[-3:]) != (h, w, 29) or out.numel() != h * w * 29 or out.dtype != torch.float32
              or not out.is_contiguous() or out.device != self.device):
            raise ValueError("pack_input: `out` must be a contiguous float32 [1,h,w,29] tensor on the engine's device")
        a = (ctypes.c_void_p * 3)(*[f.data_ptr() for f in fr])
        b = (ctypes.c_void_p * 4)(*[f.data_ptr() for f in fl])
        c = (ctypes.c_void_p * 4)(*[f.data_ptr() for f in wp])
        _lib.check(self._L.fisr_pack_input(a, b, c, h0, w0, h, w, _ptr(out), _stream(self.device)))
        return out

    def unpack_output(self, pred_hw9, want_rgb: bool = True, out_yuv=None, out_rgb=None):
        """FISRnet.py:883,903-909: -> (yuv_u8 [h,w,9], rgb_u8 [3,h,w,3]).  `out_yuv` / `out_rgb`: existing contiguous uint8 device
        tensors of those shapes to fill instead of new ones (e.g. a frame's slot of a send buffer)."""
        torch = _torch()
        h, w = pred_hw9.shape[:2]
        for t, shp, what in ((out_yuv, (h, w, 9), "out_yuv"), (out_rgb, (3, h, w, 3), "out_rgb")):
            if t is not None and (tuple(t.shape) != shp or t.dtype != torch.uint8 or not t.is_contiguous() or t.device != self.device):
                raise ValueError(f"unpack_output: `{what}` must be a contiguous uint8 {list(shp)} tensor on the engine's device")
        yuv = out_yuv if out_yuv is not None else torch.empty((h, w, 9), dtype=torch.uint8, device=self.device)
        rgb = out_rgb if out_rgb is not None else (torch.empty((3, h, w, 3), dtype=torch.uint8, device=self.device) if want_rgb else None)
        _lib.check(self._L.fisr_unpack_output(_ptr(pred_hw9.contiguous()), h, w, _ptr(yuv), _ptr(rgb), _stream(self.device)))
        return yuv, rgb

    def sse_vs_u8(self, pred, gt_u8) -> float:
        out = ctypes.c_double()
        pred = pred.contiguous()
        gt_u8 = gt_u8.to(self.device).contiguous()
        _lib.check(self._L.fisr_sse_vs_u8(_ptr(pred), _ptr(gt_u8), pred.numel(), ctypes.byref(out), _stream(self.device)))
        return out.value

    def ssim_u8(self, a_u8, b_u8, coff: int = 0) -> float:
        """SSIM_PIL-style SSIM (FISRnet.py:890-891) of the 3 channels starting at `coff` of two
        uint8 device tensors [h,w,C]."""
        a = a_u8.to(self.device).contiguous()
        b = b_u8.to(self.device).contiguous()
        if a.shape != b.shape or a.dim() != 3:
            raise ValueError("ssim_u8: tensors must be [h,w,C] of equal shape")
        out = ctypes.c_double()
        _lib.check(self._L.fisr_ssim_u8(_ptr(a), _ptr(b), a.shape[0], a.shape[1], a.shape[2], coff,
                                        ctypes.byref(out), _stream(self.device)))
        return out.value

    # ------------------------------------------------------------------ tiled forward (FISRnet.py:845-883)
    def forward_tiled(self, inp, num_patch: Tuple[int, int] = (2, 2), tiles: Optional[Sequence[int]] = None,
                      full=None, timed: bool = False, batch_tiles: Optional[bool] = None):
        """inp [B,h,w,29] on the GPU (B windows of equal size) -> full prediction [B,h*2,w*2,9]
        float32 ([h*2,w*2,9] when B == 1; not clipped -- the clip of FISRnet.py:883 is applied by
        unpack_output / sse_vs_u8).  `tiles` restricts the work to a subset of tile indices
        (tile-parallel sharding, see fisr_amd/dist.py).

        The reference runs the tiles one sess.run at a time (FISRnet.py:847-872).  Tiles are
        independent, so equal-shaped tiles (all four in the default 2x2 plan) go through ONE
        batched forward here: bit-identical results, but every conv launch gets 4x (12x for a
        whole 5-frame stack) more workgroups, which is what fills 256 CUs on the deep, small
        layers.  batch_tiles=False restores the one-tile-per-forward schedule (workspace ~5 KB per LR pixel
        of ONE tile instead of all of them); it is also the automatic fall-back when the batched workspace
        does not fit in free HBM (the reference tiles precisely because of limited memory)."""
        torch = _torch()
        if batch_tiles is None:
            batch_tiles = self.batch_tiles
        B, h, w, _ = inp.shape
        sf = self.scale_factor
        plan = [t for t in tiling.plan_tiles(h, w, tuple(num_patch), sf) if tiles is None or t.index in tiles]
        squeeze = full is None and B == 1
        if full is None:
            full = torch.zeros((B, h * sf, w * sf, 9), dtype=torch.float32, device=self.device)
        fullv = full if full.dim() == 4 else full.unsqueeze(0)
        groups = {}
        for t in plan:
            groups.setdefault((t.in_h, t.in_w) if batch_tiles else t.index, []).append(t)
        work = list(groups.values())
        while work:
            grp = work.pop(0)
            simg = torch.cat([inp[:, t.h_lo:t.h_hi, t.w_lo:t.w_hi, :] for t in grp], dim=0).contiguous()
            if timed:
                torch.cuda.synchronize(self.device)
                t0 = time.time()
            try:
                _, _, pred = self.model(simg, sf, want_all=False)
            except getattr(torch, "OutOfMemoryError", torch.cuda.OutOfMemoryError):   # (torch < 2.5 has only the cuda one)
                if len(grp) == 1:
                    raise
                # workspace of the batched group does not fit: run its tiles one by one instead
                del simg
                self._ws = None
                torch.cuda.empty_cache()
                work = [[t] for t in grp] + work
                continue
            if timed:
                torch.cuda.synchronize(self.device)
                self.inf_time.extend([(time.time() - t0) / (len(grp) * B)] * (len(grp) * B))
            for gi, t in enumerate(grp):
                for b in range(B):
                    _lib.check(self._L.fisr_stitch(_ptr(pred[gi * B + b]), t.in_h * sf, t.in_w * sf, t.src_y, t.src_x,
                                                   t.out_h, t.out_w, _ptr(fullv[b]), h * sf, w * sf, t.dst_y, t.dst_x,
                                                   _stream(self.device)))
        return full[0] if squeeze else full

    def forward_tiled_frames(self, windows, h: int, w: int, num_patch: Tuple[int, int] = (2, 2),
                             tiles: Optional[Sequence[int]] = None, full=None, timed: bool = False):
        """forward_tiled(pack_input(...)) without the packed tensor: `windows` is a list of B (frames3, flows4, warps4) tuples as
        pack_input takes them (device tensors of one common frame size), h x w the crop (FISRnet.py:820-824).  Every (tile, window)
        pair becomes one item of fisr_forward_frames, whose level inputs are assembled straight from the source planes
        (FISRnet.py:828-843 + :853-857 + :81,112-113,144 in one kernel per level) -- bit-identical to the two-step path, which stays
        the graph seam.  Returns the stitched prediction [B,2h,2w,9] float32 ([2h,2w,9] when B == 1 and `full` is None).
        `timed` appends every tile's share of its call's wall time to self.inf_time, as forward_tiled does."""
        torch = _torch()
        if not self._finalized:
            raise FisrError("weights not loaded: call load() or set_weights() first")
        B, sf = len(windows), self.scale_factor
        prepared, h0, w0 = [], None, None
        for frames_u8, flows, warps in windows:
            fr = [f.to(device=self.device, dtype=torch.uint8).contiguous() for f in frames_u8]
            fl = [f.to(device=self.device, dtype=torch.float32).contiguous() for f in flows]
            wp = [f.to(device=self.device, dtype=torch.float32).contiguous() for f in warps]
            if len(fr) != 3 or len(fl) != 4 or len(wp) != 4:
                raise ValueError("forward_tiled_frames: every window needs 3 frames, 4 flows, 4 warps")
            fr, fl, wp, a, b = fit_pack_inputs(fr, fl, wp, h, w)
            if h0 is not None and (a, b) != (h0, w0):
                raise ValueError("forward_tiled_frames: all windows must share one frame size")
            h0, w0 = a, b
            prepared.append((fr, fl, wp))
        plan = [t for t in tiling.plan_tiles(h, w, tuple(num_patch), sf) if tiles is None or t.index in tiles]
        squeeze = full is None and B == 1
        if full is None:
            full = torch.zeros((B, h * sf, w * sf, 9), dtype=torch.float32, device=self.device)
        fullv = full if full.dim() == 4 else full.unsqueeze(0)
        groups = {}
        for t in plan:
            groups.setdefault((t.in_h, t.in_w), []).append(t)
        for (th, tw), grp in groups.items():
            if th % 32 or tw % 32:
                raise ValueError("tile sizes must be multiples of 32 (FISRnet.py:820-824)")
            pairs = [(t, b) for t in grp for b in range(B)]
            per_call = _lib.MAX_SRC_ITEMS if self.batch_tiles else 1      # (batch_tiles=False: the reference's one tile per forward)
            work = [pairs[k:k + per_call] for k in range(0, len(pairs), per_call)]
            while work:
                chunk = work.pop(0)
                n = len(chunk)
                items = (_lib.SrcItem * n)()
                for i, (t, b) in enumerate(chunk):
                    fr, fl, wp = prepared[b]
                    for j in range(3):
                        items[i].frames[j] = fr[j].data_ptr()
                    for j in range(4):
                        items[i].flows[j] = fl[j].data_ptr()
                        items[i].warps[j] = wp[j].data_ptr()
                    items[i].y0, items[i].x0 = t.h_lo, t.w_lo
                try:
                    pred = torch.empty((n, th * sf, tw * sf, 9), dtype=torch.float32, device=self.device)
                    ws = self._workspace(n, th, tw)
                except getattr(torch, "OutOfMemoryError", torch.cuda.OutOfMemoryError):
                    if n == 1:
                        raise
                    # the batched workspace does not fit (the reference tiles because memory is limited): one tile per call instead
                    self._ws = None
                    torch.cuda.empty_cache()
                    work = [[c] for c in chunk] + work
                    continue
                if timed:
                    torch.cuda.synchronize(self.device)
                    t0 = time.time()
                _lib.check(self._L.fisr_forward_frames(self._ctx, items, n, h0, w0, th, tw, _ptr(pred), None, None, _ptr(ws),
                                                       ws.numel(), _stream(self.device)), self._ctx)
                if timed:      # (as forward_tiled: the time of one tile's forward, FISRnet.py:868-874 -- here incl. its input assembly)
                    torch.cuda.synchronize(self.device)
                    self.inf_time.extend([(time.time() - t0) / n] * n)
                for i, (t, b) in enumerate(chunk):
                    _lib.check(self._L.fisr_stitch(_ptr(pred[i]), t.in_h * sf, t.in_w * sf, t.src_y, t.src_x, t.out_h, t.out_w,
                                                   _ptr(fullv[b]), h * sf, w * sf, t.dst_y, t.dst_x, _stream(self.device)))
        return full[0] if squeeze else full

    # ------------------------------------------------------------------ harnesses
    def test(self):
        from .harness import run_test
        return run_test(self)

    def FISR_for_video(self, flow_file_name, warp_file_name):
        from .harness import run_fisr_for_video
        return run_fisr_for_video(self, flow_file_name, warp_file_name)

    def close(self):
        if getattr(self, "_ctx", None) is not None and self._ctx.value:
            self._L.fisr_destroy(self._ctx)
            self._ctx = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
