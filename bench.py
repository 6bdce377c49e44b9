#!/usr/bin/env python3
"""Benchmark of the FISRnet hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W [--precision fp32|bf16x3|f16f8|mixed|fp16]
                    [--parallelism frame|tile] [--others bf16x3,f16f8]

A "step" is one pass of the hot path over one 5-frame 1080x1920 LR stack resident in HBM
(cfg2 of BASELINE.json): 3 sliding windows x [input assembly -> 2x2 tiles of 544x992x29 (32-px
halo) through the 138-conv FISRnet forward -> trim/stitch -> clip/quantise/YUV->RGB], i.e. the
inner loops of `FISRnet.test` / `FISR_for_video` (reference FISRnet.py:798-910), producing 9 raw
= 7 unique 2048x3840 frames.  Flows and warped frames are pre-made inputs as in cfg2.

The headline (`value`, `dtype`) is the fp32 engine -- cfg2 says fp32, the reference computes in fp32: fp32 tensors,
fp32 arithmetic, Winograd minimal filtering for the 3x3 convolutions (as cuDNN does under the reference's
TensorFlow): F(4x4,3x3) on every map >= 48x64 and on the 512-channel maps, F(2x2,3x3) on the rest; under
`other_precisions`, `fp32w` is the all-F(2x2) engine and `fp32d` the same engine with the direct exact-fp32 kernel.
The split-precision engines (bf16x3, f16f8: fp32-grade results on the 16-bit / fp8 matrix pipes, far
inside the reference tolerance of +-0.02 dB) are timed in the same run under `other_precisions`, each
with its own roofline, its full-size comparison against the fp32 engine of this run and a check of one
544x992 tile against the committed fp64-oracle grid (tests/golden/model_544x992_sparse.npz).

N > 1 (launched by torch.distributed.run, one rank per GPU, RCCL):
  --parallelism frame (default): every rank processes its own stack (weights replicated, no data-path
      collective in the compute) and the uint8 output frames are gathered to rank 0 over xGMI, as cfg4 of
      BASELINE.json asks -> weak scaling.
  --parallelism tile: the ranks form groups of 4 (one 2x2 tile plan per group, reference seams and
      numerics); a group works on one stack: every rank packs only its core of the input, the 32-px halos
      are all-gathered, each rank runs the forward of ITS tile for the 3 windows, the trimmed uint8 tiles
      are all-gathered (fisr_amd/dist.py).  N = 8 is 2 stacks x 4 tiles.
Timing is barrier + synchronize on both sides, max over ranks.

One JSON line is printed by rank 0.  `roofline` is measured in a second, instrumented pass (HIP events on
the launch stream around every kernel, inside libfisr_hip.so; HIP events on the same stream around the glue
calls).  The counter-derived fields (`traffic`, `pmc_mfma_busy_frac`, `pmc_gb_per_launch`) are NOT measured by this
process: they are read from profiles/pmc_traffic.json, the summary of a `rocprofv3 --pmc` run of this same command
(scripts/gpu_profile.sh), every one carries `source` = that file + the library it was run on, and a field is
dropped (null + `..._dropped`) when that library is not the running one or when the kernel's launches per step
differ between the two runs; `cpu_baseline` times the C oracle (oracle/fisr_oracle.c, OpenMP, fp32) and `cpu_baseline_onednn`
the torch-CPU/oneDNN twin (oracle/torch_cpu.py) on ONE full 544x992 tile on rank 0 at N=1.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np

FLOP_PER_LR_PX = 5288328.0          # SURVEY.md 8d / BASELINE.md section 2 (2 x 2 644 164 MAC)
HBM_PEAK_GBPS = 8000.0              # MI355X_MICROARCH.md (spec; ~6300 achievable with a float4 copy)
# dense MFMA peak of the instruction class each engine issues (MI355X_MICROARCH.md), TFLOP/s
PEAK = {"fp32": 157.3, "fp32d": 157.3, "fp32w": 157.3, "fp32w4": 157.3, "fp16": 2500.0, "bf16x3": 2500.0, "f16f8": 2500.0, "mixed": 2500.0}
# matrix-pipe FLOPs EXECUTED per algorithmic (direct 3x3 convolution) FLOP, per kernel class: Winograd F(2x2,3x3) issues 16
# multiplies per 2x2 outputs instead of 36; split bf16 issues 3 MFMAs per product; fp16 + fp8 remainder one fp16 MFMA per tap
# and one block-scaled fp8 MFMA (twice the fp16 rate per K element, four times the K) per tap pair: 1 + 5/9 * ... = 2.11 units
def executed_per_algorithmic(kernel_name):
    if kernel_name.startswith("conv3x3_wf4"):       # Winograd F(4x4,3x3): 36 multiplies per 4x4 outputs instead of 144
        return 36.0 / 144.0
    if kernel_name.startswith("conv3x3_wino"):
        return 16.0 / 36.0
    if "<bf16x3" in kernel_name:
        return 3.0
    if "<f16f8" in kernel_name:
        return 2.11
    return 1.0
DTYPE = {"fp32": "f32", "fp32d": "f32", "fp32w": "f32", "fp32w4": "f32",
         "fp16": "f16 (f32 accumulate)",
         "bf16x3": "bf16x3 (values as hi+lo bf16 pairs, 3 bf16 MFMA per product, f32 accumulate)",
         "f16f8": "f16f8 (values as fp16 + fp8 remainder, fp16 MFMA + block-scaled fp8 MFMA for the cross terms, f32 accumulate)",
         "mixed": "mixed (f16f8 at the full and half resolution of level 3 -- first two encoder levels, last two decoder levels, the SR head --, f16 elsewhere incl. the FI-SR head, r04; f32 accumulate)"}
UNIQUE_PER_STACK = 7                # 3 windows x 3 frames, overlaps counted once (FISRnet.py:913-920)


def synthetic_stack(seed, H=1080, W=1920):
    """5 LR YUV frames (uint8) and 8 flows (float32 px) of a 5-frame stack."""
    rng = np.random.default_rng(seed)
    coarse = rng.integers(0, 256, (5, H // 8, W // 8, 3)).astype(np.float32)
    frames = np.repeat(np.repeat(coarse, 8, axis=1), 8, axis=2)
    frames = np.clip(frames + rng.normal(0, 6, frames.shape), 0, 255).astype(np.uint8)
    fc = rng.normal(0, 4, (8, H // 32 + 1, W // 32 + 1, 2)).astype(np.float32)
    flows = np.repeat(np.repeat(fc, 32, axis=1), 32, axis=2)[:, :H, :W, :].copy()
    return frames, flows


class Workload:
    """cfg2 on one GPU (or one tile group): device-resident inputs + the step function of one engine."""

    def __init__(self, torch, dev, stack_id, patch, batch, parallelism="frame", topo=None, group=None,
                 gather_group_world=1):
        from fisr_amd import tiling
        self.torch, self.dev, self.patch, self.batch = torch, dev, patch, batch
        self.parallelism, self.topo, self.group = parallelism, topo, group
        self.gather_world = gather_group_world
        H0, W0 = 1080, 1920
        self.h, self.w = tiling.crop_hw(H0, W0, patch)
        frames_np, flows_np = synthetic_stack(100 + stack_id, H0, W0)
        self.frames = [torch.from_numpy(f).to(dev) for f in frames_np]
        self.flows = [torch.from_numpy(f).to(dev) for f in flows_np]
        self.warps = None
        self.tiles = tiling.plan_tiles(self.h, self.w, patch)
        self.full = torch.zeros((3, self.h * 2, self.w * 2, 9), dtype=torch.float32, device=dev)
        self.yuv = torch.zeros((3, self.h * 2, self.w * 2, 9), dtype=torch.uint8, device=dev)
        # cfg4: the uint8 output frames of every rank travel to rank 0 -- asynchronously (side stream, two send buffers, a
        # receive buffer allocated once; fisr_amd/dist.py AsyncGather), so the gather of step k runs under the compute of step k+1
        self.gather = None
        if gather_group_world > 1:
            from fisr_amd import dist as fdist
            self.gather = fdist.AsyncGather(tuple(self.yuv.shape), torch.uint8, dev, dst=0)
        self.step_index = 0
        self.glue_events = None       # instrumented pass: {kernel: [(ev0, ev1, launches)]}
        # "fused": input assembly + tile cut + level inputs in one kernel per level (fisr_forward_frames; bit-identical);
        # "packed": fisr_pack_input -> [3,h,w,29] -> tile slices -> fisr_forward, the reference's sequence of steps
        self.input_path = "fused"

    def premake_warps(self, net):
        # pre-made warps (cfg2): produced once with the warp kernel, outside the timed region
        self.warps = []
        for p in range(4):
            self.warps.append(net.warp(self.frames[p + 1], self.flows[2 * p]))
            self.warps.append(net.warp(self.frames[p], self.flows[2 * p + 1]))

    # -- glue calls, optionally bracketed by HIP events on the stream they launch on (torch's current stream)
    def _timed(self, name, launches, fn):
        if self.glue_events is None:
            return fn()
        torch = self.torch
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        out = fn()
        e1.record()
        self.glue_events.setdefault(name, []).append((e0, e1, launches))
        return out

    def step(self, net):
        return self._step_tile(net) if self.parallelism == "tile" else self._step_frame(net)

    def _step_frame(self, net):
        torch, h, w = self.torch, self.h, self.w
        fr, fl, wp = self.frames, self.flows, self.warps
        pack = lambda s, out=None: self._timed("pack_input", 1, lambda: net.pack_input(fr[s:s + 3], fl[2 * s:2 * s + 4], wp[2 * s:2 * s + 4], h, w, out=out))
        if self.batch == "stack" and self.input_path == "fused":
            # the 3 windows x 4 tiles as the 12 items of one fisr_forward_frames call: the packed tensor is never written
            net.forward_tiled_frames([(fr[s:s + 3], fl[2 * s:2 * s + 4], wp[2 * s:2 * s + 4]) for s in range(3)], h, w, self.patch, full=self.full)
        elif self.batch == "stack":
            # the 3 sliding windows (FISRnet.py:799) x 4 tiles (:847) are independent -> one batched forward; every
            # window is packed straight into its slot of the batch
            if getattr(self, "inp3", None) is None:
                self.inp3 = torch.empty((3, h, w, 29), dtype=torch.float32, device=self.dev)
            for s in range(3):
                pack(s, self.inp3[s:s + 1])
            net.forward_tiled(self.inp3, self.patch, full=self.full)
        else:
            for s in range(3):
                net.forward_tiled(pack(s), self.patch, full=self.full[s:s + 1], batch_tiles=(self.batch == "window"))
        rgb = None
        use_gather = self.gather is not None and self.gather_world > 1
        dst = self.gather.buffer(self.step_index) if use_gather else self.yuv     # (waits on the stream for the gather that last read it)
        if getattr(self, "rgb", None) is None:
            self.rgb = torch.empty((3, self.h * 2, self.w * 2, 3), dtype=torch.uint8, device=self.dev)
        for s in range(3):
            # FISRnet.py:883, 903-909, written straight into the frame's slot of the output / send buffer
            yuv, rgb = self._timed("unpack_output", 1, lambda: net.unpack_output(self.full[s], out_yuv=dst[s], out_rgb=self.rgb))
        if use_gather:
            self.gather.submit(self.step_index)
        self.step_index += 1
        return dst, rgb

    def _step_tile(self, net):
        from fisr_amd import dist as fdist
        torch, h, w = self.torch, self.h, self.w
        fr, fl, wp = self.frames, self.flows, self.warps
        cores = torch.cat([fdist.pack_core(net, fr[s:s + 3], fl[2 * s:2 * s + 4], wp[2 * s:2 * s + 4], h, w,
                                           self.patch, self.topo.tile) for s in range(3)], dim=0)
        yuv, rgb = fdist.tile_parallel_engine_window(net, cores, self.patch, group=self.group, want_rgb=True)
        return yuv, rgb

    @property
    def flop_per_stack(self):
        return 3 * sum(t.in_h * t.in_w for t in self.tiles) * FLOP_PER_LR_PX


PMC_SRC = None          # how the counter-derived fields of this line are labelled (set by _pmc_table)


def _src_of(library):
    """'... src <16 hex digits> ...' -> the hex digits"""
    import re
    m = re.search(r"src ([0-9a-f]{16})", library or "")
    return m.group(1) if m else None


def _pmc_table_checked(running_library):
    """profiles/pmc_traffic.json, or {} when it was measured on another library (its `_meta.library` names the build): a counter
    of another kernel build under this build's name would read as measured."""
    global PMC_SRC
    pmc = _pmc_table()
    meta = pmc.pop("_meta", None) or {}
    have, want = _src_of(meta.get("library")), _src_of(running_library)
    if not pmc:
        PMC_SRC = {"file": "profiles/pmc_traffic.json", "status": "absent"}
        return {}, meta
    if have is None or have != want:
        PMC_SRC = {"file": "profiles/pmc_traffic.json", "status": "dropped: measured on library src %s, running src %s" % (have, want)}
        return {}, meta
    PMC_SRC = {"file": "profiles/pmc_traffic.json", "status": "same library", "library_src": have,
               "how": "rocprofv3 --pmc, separate passes (FETCH_SIZE x 2 + WRITE_SIZE; SQ_VALU_MFMA_BUSY_CYCLES / GRBM_GUI_ACTIVE), scripts/gpu_profile.sh"}
    return pmc, meta


def _pmc_same_population(entry, passes, launches_per_step):
    """the PMC run averaged `dispatches` launches of this kernel NAME; it is this engine's kernel population only if that is `passes` x
    this run's launches per step, `passes` = the whole number of forward passes the PMC run made with this engine, taken from the
    engine's dominant kernel (r03: the two maxpool2 launches the fp32 engine has left vs the 18 of the other engines in the same table)"""
    if not passes or not entry.get("dispatches") or not launches_per_step:
        return False
    return abs(entry["dispatches"] / launches_per_step - passes) <= 0.05


def _pmc_passes(entry, launches_per_step):
    """whole number of forward passes behind a table entry of an engine's dominant kernel, or 0 when it is not a whole number"""
    if not entry or not entry.get("dispatches") or not launches_per_step:
        return 0
    r = entry["dispatches"] / launches_per_step
    return round(r) if r >= 0.95 and abs(r - round(r)) <= 0.05 else 0


def memory_plan(torch, net, dev, patch, batch, parallelism, topo, gather_world, rank, weight_bytes):
    """Bytes this rank's step holds on its device, by item, and whether they fit (cfg2 workload; `gather_world` > 1: rank 0 also
    holds the receive buffer of the frame gather).  The activation arena is what fisr_workspace_bytes says for the largest
    forward batch of the plan -- the library's own figure, not an estimate."""
    from fisr_amd import tiling
    H0, W0 = 1080, 1920
    h, w = tiling.crop_hw(H0, W0, patch)
    tiles = tiling.plan_tiles(h, w, patch)
    if parallelism == "tile":
        t = tiles[topo.tile]
        n_fwd, th, tw = 3, t.in_h, t.in_w                       # the rank's own tile of the 3 windows
    else:
        per = {"stack": 3 * len(tiles), "window": len(tiles), "tile": 1}[batch]
        th, tw = max(t.in_h for t in tiles), max(t.in_w for t in tiles)
        n_fwd = per
    arena = int(net._L.fisr_workspace_bytes(net._ctx, n_fwd, th, tw))
    out_f32 = 3 * (2 * h) * (2 * w) * 9 * 4
    out_u8 = 3 * (2 * h) * (2 * w) * 9
    items = {
        "weights_packed_measured": int(weight_bytes),
        "activation_arena": arena,
        "forward_batch": [n_fwd, th, tw],
        "inputs_frames_flows_warps": 5 * H0 * W0 * 3 + 8 * H0 * W0 * 2 * 4 + 8 * H0 * W0 * 3 * 4,
        "packed_input_3_windows": 3 * h * w * 29 * 4,
        "tile_batch_in_out": n_fwd * th * tw * 29 * 4 + n_fwd * 4 * th * tw * 9 * 4,
        "stitched_output_f32": out_f32, "output_yuv_rgb_u8": 3 * out_u8,
        "gather_send_buffers": 2 * out_u8 if gather_world > 1 else 0,
        "gather_receive_buffer": gather_world * out_u8 if gather_world > 1 and rank == 0 else 0,
    }
    total = sum(v for k, v in items.items() if isinstance(v, int))
    free_b, total_b = torch.cuda.mem_get_info(dev)
    free_b += int(weight_bytes)                               # (the weights are already resident)
    sharing = int(os.environ.get("FISR_BENCH_ONE_DEVICE", "0") == "1") and int(os.environ.get("WORLD_SIZE", "1")) or 1
    budget = total_b // max(sharing, 1)                        # (test mode: all ranks share device 0)
    return {"rank": rank, "device": str(dev), "bytes": items, "total_bytes": total, "device_total_bytes": int(total_b),
            "device_free_bytes_now": int(free_b), "ranks_sharing_device": sharing,
            "fits": bool(total * 1.05 <= budget)}              # 5 % for the allocator's rounding


def _pmc_table():
    """HBM bytes per launch of every kernel, from the rocprofv3 PMC passes of this same command
    (scripts/gpu_profile.sh -> profiles/pmc_traffic.json; FETCH_SIZE x2 gfx950 correction + WRITE_SIZE,
    separate passes as MI355X_MICROARCH.md prescribes)."""
    try:
        with open(os.path.join(ROOT, "profiles", "pmc_traffic.json")) as f:
            return json.load(f)
    except (OSError, ValueError):
        return {}


def _pmc_key(name):
    if name.startswith("conv3x3_dma"):
        return "conv3x3_dma_f16_kernel<false>"
    if name.startswith("conv3x3_wf4"):
        if "res+pool" in name:
            return "conv3x3_wf4_kernel<false, true, true, false, false>"
        if "up2" in name:
            return "conv3x3_wf4_kernel<false, false, false, true, false>"
        # (5th parameter: the GENERAL instantiation of the flow network)
        return "conv3x3_wf4_kernel<%s, %s, false, false, false>" % ("true" if "relu_in" in name else "false", "false" if "nores" in name else "true")
    tname = {"f32": "float", "f32w": "float", "f16": "_Float16", "bf16x3": "bsplit", "f16f8": "fsplit"}[name.split("<")[1].split(",")[0]]
    if name.startswith("conv3x3_wino"):
        return "conv3x3_wino8p_kernel<%s, false, %s>" % ("true" if "relu_in" in name else "false", "false" if "nores" in name else "true")
    nt = name.split("NT")[1][0]
    if tname == "_Float16":        # (rocprofv3 leaves the _Float16 instantiations mangled)
        return f"_ZN4fisr19conv3x3_mfma_kernelIDF16_Li{nt}ELb{1 if 'f32out' in name else 0}E"
    return f"conv3x3_mfma_kernel<{tname}, {nt}, {'true' if 'f32out' in name else 'false'}"


def _pmc_field(pmc, name, field):
    try:
        hits = [v for k, v in pmc.items() if k.startswith(_pmc_key(name)) and field in v]
        if hits:
            return round(max(hits, key=lambda v: v.get("dispatches", 0))[field], 4)
    except (KeyError, IndexError, ValueError):
        pass
    return None


def _pmc_conv_traffic(pmc, name):
    v = _pmc_field(pmc, name, "hbm_bytes_per_launch")
    return None if v is None else round(v / 1e9, 4)


def roofline_pass(net, wl, precision, reps, layer_profile=None):
    """Second, instrumented pass: HIP events around every kernel of the forward (inside the library, on the
    launch stream) and around the glue calls (torch events on the same stream)."""
    torch = wl.torch
    net.profile(1)
    wl.glue_events = {}
    for _ in range(reps):
        wl.step(net)
    torch.cuda.synchronize(wl.dev)
    prof = net.profile_read()
    net.profile(0)
    glue_ev, wl.glue_events = wl.glue_events, None
    if layer_profile:
        net.profile(2)
        wl.step(net)
        torch.cuda.synchronize(wl.dev)
        layers = net.profile_read()
        net.profile(0)
        for p_ in layers:
            p_["tflops"] = round(p_["flops"] / (p_["ms"] * 1e-3) / 1e12, 2) if p_["ms"] > 0 and p_["flops"] else 0.0
            p_["us_per_launch"] = round(p_["ms"] * 1e3 / max(1, p_["launches"]), 1)
        os.makedirs(os.path.dirname(os.path.abspath(layer_profile)), exist_ok=True)
        with open(layer_profile, "w") as f:
            json.dump(sorted(layers, key=lambda q: -q["ms"]), f, indent=1)
    convs = [p for p in prof if p["name"].startswith("conv3x3") and p["launches"]]
    if not convs:
        return None
    from fisr_amd import lib as _fl
    pmc, pmc_meta = _pmc_table_checked(_fl.lib().fisr_version().decode())
    lps = {p["name"]: p["launches"] / max(reps, 1) for p in prof}          # launches per step of this run, per kernel
    dropped = []

    def pmc_raw(key):
        hits = [v for k, v in pmc.items() if k.startswith(key)]
        return max(hits, key=lambda v: v.get("dispatches", 0)) if hits else None

    # forward passes the PMC run made with THIS engine: from the engine's dominant conv kernel
    dom_name = max(convs, key=lambda p: p["ms"])["name"]
    try:
        pmc_passes = _pmc_passes(pmc_raw(_pmc_key(dom_name)), lps.get(dom_name, 0))
    except (KeyError, IndexError, ValueError):
        pmc_passes = 0

    def pmc_entry(key, name):
        """the PMC table's entry for kernel `name` (table key prefix `key`), or None (and a note) when its launch population differs"""
        e = pmc_raw(key)
        if e is None:
            return None
        if not _pmc_same_population(e, pmc_passes, lps.get(name, 0)):
            dropped.append("%s: %s dispatches in the PMC run, %.1f launches per step here, %s passes of this engine" % (name, e.get("dispatches", 0), lps.get(name, 0), pmc_passes or "no whole number of"))
            return None
        return e

    def conv_field(name, field, scale=1.0):
        try:
            e = pmc_entry(_pmc_key(name), name)
        except (KeyError, IndexError, ValueError):
            e = None
        return None if e is None or field not in e else round(e[field] * scale, 4)

    dom = max(convs, key=lambda p: p["ms"])
    ach = dom["flops"] / (dom["ms"] * 1e-3) / 1e12
    peak = PEAK[precision]
    # ---- HBM-bound kernels: GB/s against the 8 TB/s peak, from algorithmic bytes (and the PMC bytes where
    # profiles/pmc_traffic.json has the kernel) over the HIP-event time of this run
    hbm = {}

    def add_hbm(name, ms, launches, alg_bytes, pmc_key):
        if not launches or ms <= 0:
            return
        rec = {"avg_launch_us": round(ms * 1e3 / launches, 2), "launches": int(launches),
               "algorithmic_gb_per_launch": round(alg_bytes / launches / 1e9, 4),
               "algorithmic_gbps": round(alg_bytes / (ms * 1e-3) / 1e9, 1)}
        rec["frac_of_hbm_peak"] = round(rec["algorithmic_gbps"] / HBM_PEAK_GBPS, 4)
        e = pmc_entry(pmc_key, name) if name in lps else ([v for k, v in pmc.items() if k.startswith(pmc_key)] or [None])[0]
        if e and "hbm_bytes_per_launch" in e:
            rec["pmc_gb_per_launch"] = round(e["hbm_bytes_per_launch"] / 1e9, 4)
            rec["pmc_over_algorithmic"] = round(e["hbm_bytes_per_launch"] / max(alg_bytes / launches, 1.0), 3)
        hbm[name] = rec

    for p in prof:
        if p["launches"] and (not p["name"].startswith("conv3x3") or "NT0" in p["name"]):
            add_hbm(p["name"], p["ms"], p["launches"], p["bytes"], p["name"].split("<")[0] + "_kernel")
    px_lr, px_hr = wl.h * wl.w, 4 * wl.h * wl.w
    glue_bytes = {"pack_input": px_lr * (9 + 32 + 48 + 116.0),            # 3 u8 frames + 4 flows + 4 warps in, 29 f32 out
                  "unpack_output": px_hr * (36 + 9 + 9.0),                # 9 f32 in, 9 u8 yuv + 9 u8 rgb out
                  "stitch": None}
    for name, evs in glue_ev.items():
        ms = sum(e0.elapsed_time(e1) for e0, e1, _ in evs)
        launches = sum(k for _, _, k in evs)
        if glue_bytes.get(name):
            add_hbm(name, ms, launches, glue_bytes[name] * launches, name + "_kernel")
    tot_ms = sum(p["ms"] for p in prof)
    # `achieved` / `frac`: what the matrix pipe EXECUTED (a roofline fraction, <= 1 by construction): the algorithmic
    # direct-convolution FLOPs of the launch x the kernel's executed-per-algorithmic factor / its average duration.  The
    # algorithmic rate itself (which exceeds the fp32 peak for the Winograd kernel: it skips 20 of 36 multiplies) is kept
    # as `algorithmic_tflops` / `algorithmic_over_peak`.
    epa = executed_per_algorithmic(dom["name"])
    busy = conv_field(dom["name"], "mfma_busy_frac")
    rl = {"bound": "mfma", "kernel": dom["name"], "achieved": round(epa * ach, 2), "peak": peak,
          "unit": "TFLOP/s", "frac": round(epa * ach / peak, 4),
          "achieved_is": "matrix-pipe FLOPs executed per second = algorithmic FLOPs x executed_per_algorithmic / duration",
          "executed_per_algorithmic": round(epa, 4),
          "algorithmic_tflops": round(ach, 2), "algorithmic_over_peak": round(ach / peak, 4),
          "pmc_mfma_busy_frac": busy,
          "traffic": conv_field(dom["name"], "hbm_bytes_per_launch", 1e-9),
          "traffic_unit": "GB of HBM per launch (rocprofv3 PMC: FETCH_SIZE x 2 + WRITE_SIZE in separate passes); null = not measured for this build / launch population",
          "counter_fields_source": PMC_SRC,
          "avg_launch_us": round(dom["ms"] * 1e3 / dom["launches"], 2), "launches": int(dom["launches"]),
          "algorithmic_gflop_per_launch": round(dom["flops"] / dom["launches"] / 1e9, 2),
          "algorithmic_gbyte_per_launch": round(dom["bytes"] / dom["launches"] / 1e9, 4),
          "share_of_gpu_time": round(dom["ms"] / tot_ms, 4),
          "all_conv_algorithmic_tflops": round(sum(p["flops"] for p in convs) / (sum(p["ms"] for p in convs) * 1e-3) / 1e12, 2),
          "all_conv_executed_frac": round(sum(p["flops"] * executed_per_algorithmic(p["name"]) for p in convs) /
                                          (sum(p["ms"] for p in convs) * 1e-3) / 1e12 / peak, 4),
          "kernels": {p["name"]: {"ms": round(p["ms"], 3), "launches": int(p["launches"]),
                                  **({"executed_frac": round(executed_per_algorithmic(p["name"]) * p["flops"] / (p["ms"] * 1e-3) / 1e12 / peak, 4),
                                      "traffic_gb": conv_field(p["name"], "hbm_bytes_per_launch", 1e-9)} if p["name"].startswith("conv3x3") and p["ms"] > 0 else {})}
                      for p in prof},
          "hbm_bound_kernels": hbm}
    if dropped:
        rl["counter_fields_dropped"] = sorted(set(dropped))
    return rl


def time_warp(net, wl, reps=8):
    """The frame warp (separate HIP gather kernel, pre-made in cfg2 so outside `value`): HIP events on its
    launch stream around `reps` launches at 1080x1920; algorithmic bytes 44 B/px."""
    torch = wl.torch
    net.warp(wl.frames[1], wl.flows[0])
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    src = wl.frames[1].float().contiguous()
    e0.record()
    for _ in range(reps):
        net.warp(src, wl.flows[0])
    e1.record()
    torch.cuda.synchronize(wl.dev)
    us = e0.elapsed_time(e1) * 1e3 / reps
    px = wl.flows[0].shape[0] * wl.flows[0].shape[1]
    rec = {"avg_launch_us": round(us, 2), "algorithmic_gb_per_launch": round(px * 44 / 1e9, 4),
           "algorithmic_gbps": round(px * 44 / us / 1e3, 1)}
    rec["frac_of_hbm_peak"] = round(rec["algorithmic_gbps"] / HBM_PEAK_GBPS, 4)
    from fisr_amd import lib as _fl
    hit = _pmc_table_checked(_fl.lib().fisr_version().decode())[0].get("warp_kernel")
    if hit:             # (one launch shape only: 1080 x 1920, no population to compare)
        rec["pmc_gb_per_launch"] = round(hit["hbm_bytes_per_launch"] / 1e9, 4)
        rec["pmc_source"] = PMC_SRC
    return rec


def _psnr_shift(torch, out, ref, seed=7):
    """PSNR protocol of SURVEY.md 8c-ii on the GPU: pseudo ground truth = ref + Gaussian noise at the reference's published PSNRs
    (37.86 dB on the FI-SR channels, 48.07 dB on the SR channels, README.md:97); returns max over the channel groups of
    |PSNR(out, gt) - PSNR(ref, gt)| in dB.  out, ref: [..., 9] float tensors, clipped to [0, 1] here."""
    g = torch.Generator(device=out.device).manual_seed(seed)
    a, b = out.clamp(0, 1).double(), ref.clamp(0, 1).double()
    worst = 0.0
    for ch, db in ((slice(0, 3), 37.86), (slice(3, 6), 48.07), (slice(6, 9), 37.86)):
        gt = b[..., ch] + torch.randn(b[..., ch].shape, generator=g, device=out.device, dtype=torch.float64) * 10 ** (-db / 20)
        ps = lambda t: 10 * torch.log10(1.0 / ((t - gt) ** 2).mean())
        worst = max(worst, abs(float(ps(a[..., ch]) - ps(b[..., ch]))))
    return worst


def cfg5_pipeline(torch, dev, W, wl, net32, local_rank, reps=3):
    """cfg5 of BASELINE.json -- "FISR_for_video end-to-end: on-GPU PWC-Net flow + warp + FISRnet fused pipeline, bf16, 1xMI355X" -- on
    the same 5-frame 1080p stack with NOTHING pre-made: PWC-Net-large for the 8 directions in one fisr_pwc_flow_stack call (one
    pyramid per frame, the directions batched), the 8 frame warps, then the FISRnet step.  16-bit arithmetic as the config
    reads: flow engine FISR_PREC_F16 (fp16 features, fp32 accumulation, fp32 flows), network engine `mixed` (fp16 / fp16 + fp8
    remainder).  Timed serially and as a two-stream pipeline (flow + warps of stack k+1 on a side stream under the network of
    stack k, two sets of flow / warp buffers).  Accuracy: the 16-bit pipeline's output frames against the all-fp32 pipeline's
    (fp32 flow, fp32 network) with the PSNR protocol, the fp16 flow against the fp32 flow, and the fp16 flow of the committed
    1080p golden pair against the float64 oracle (tests/golden/pwc_flow_1080p_sparse.npz).  Seeded stand-in weights (neither
    checkpoint is in the reference tree)."""
    from fisr_amd import pwcnet
    from fisr_amd.fisrnet import FISRnet
    Wp = pwcnet.synthetic_weights(595000)
    out = {"what": "5-frame 1080p stack, nothing pre-made: flow of 8 directions (PWC-Net-large on the x2 up-scaled frames, one pyramid per "
                   "frame, directions batched) + 8 warps + the FISRnet step -> 7 unique 4K frames",
           "unit": "frames/s", "weights": "synthetic seeded (neither checkpoint is in the reference tree)"}
    eng = FISRnet(device=f"cuda:{local_rank}", precision="mixed")
    eng.set_weights(W)
    keep = (wl.flows, wl.warps)
    try:
        res = {}
        for tag, fprec, net in (("f32", "fp32", net32), ("16bit", "fp16", eng)):
            pwc = pwcnet.PWCNet(f"cuda:{local_rank}", precision=fprec)
            pwc.set_weights(Wp)
            bufs = [torch.empty((4, 2, 1080, 1920, 2), dtype=torch.float32, device=dev) for _ in range(2)]

            def produce(k):
                fl = pwc.flow_stack(wl.frames, out=bufs[k & 1])
                flows = [fl[p, d] for p in range(4) for d in range(2)]
                warps = []
                for p in range(4):
                    warps.append(net.warp(wl.frames[p + 1], flows[2 * p]))
                    warps.append(net.warp(wl.frames[p], flows[2 * p + 1]))
                return flows, warps

            # serial, three sections timed with events on the one stream
            wl.flows, wl.warps = produce(0)
            wl.step(net)
            torch.cuda.synchronize(dev)
            e = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
            e[0].record()
            for k in range(reps):
                fl = pwc.flow_stack(wl.frames, out=bufs[0])
            e[1].record()
            flows = [fl[p, d] for p in range(4) for d in range(2)]
            for k in range(reps):
                wl.flows = flows
                wl.premake_warps(net)
            e[2].record()
            for k in range(reps):
                wl.step(net)
            e[3].record()
            torch.cuda.synchronize(dev)
            t_flow, t_warp, t_net = (e[i].elapsed_time(e[i + 1]) / reps for i in range(3))
            rec = {"flow_dtype": "f32" if fprec == "fp32" else "f16 features, f32 accumulate, f32 flows", "network_engine": net.engine_description(),
                   "flow_ms": round(t_flow, 2), "warp_ms": round(t_warp, 3), "fisrnet_ms": round(t_net, 2),
                   "serial_fps": round(UNIQUE_PER_STACK / ((t_flow + t_warp + t_net) * 1e-3), 3)}
            res[tag] = {"full": wl.full.clone(), "flows": torch.stack(flows).clone()}
            # two-stream pipeline: the side stream prepares stack k+1 while the main stream runs the network on stack k
            side = torch.cuda.Stream(device=dev)
            main = torch.cuda.current_stream(dev)
            ready = [torch.cuda.Event() for _ in range(2)]
            freed = [torch.cuda.Event() for _ in range(2)]
            sets = [None, None]

            def run_pipeline(n):
                with torch.cuda.stream(side):
                    sets[0] = produce(0)
                    ready[0].record(side)
                for k in range(n):
                    if k + 1 < n:
                        with torch.cuda.stream(side):
                            if k >= 1:
                                side.wait_event(freed[(k + 1) & 1])          # its buffers were the network's input two stacks ago
                            sets[(k + 1) & 1] = produce(k + 1)
                            ready[(k + 1) & 1].record(side)
                    main.wait_event(ready[k & 1])
                    wl.flows, wl.warps = sets[k & 1]
                    wl.step(net)
                    freed[k & 1].record(main)

            run_pipeline(2)
            torch.cuda.synchronize(dev)
            t0 = time.perf_counter()
            n = max(4, 2 * reps)
            run_pipeline(n)
            torch.cuda.synchronize(dev)
            dt = (time.perf_counter() - t0) / n
            rec["pipelined_ms_per_stack"] = round(dt * 1e3, 2)
            rec["pipelined_fps"] = round(UNIQUE_PER_STACK / dt, 3)
            out["all_f32" if tag == "f32" else "sixteen_bit"] = rec
            pwc.close()
        sx = out["sixteen_bit"]
        out["value"] = max(sx["serial_fps"], sx["pipelined_fps"])
        out["dtype"] = "16-bit: fp16 flow features + the mixed FISRnet engine (fp32 accumulation everywhere, fp32 flows)"
        d = (res["16bit"]["flows"] - res["f32"]["flows"]).double()
        out["flow_16bit_vs_f32_px"] = {"max_abs": float(d.abs().max()), "rms": float((d * d).mean().sqrt()),
                                       "flow_abs_max": float(res["f32"]["flows"].abs().max())}
        a, b = res["16bit"]["full"], res["f32"]["full"]
        dd = (a.clamp(0, 1) - b.clamp(0, 1)).double()
        out["frames_16bit_vs_all_f32"] = {"what": "all 3x2048x3840x9 output values of the stack, 16-bit pipeline vs fp32 flow + fp32 network",
                                          "max_abs": float(dd.abs().max()), "rms": float((dd * dd).mean().sqrt()),
                                          "psnr_shift_db": round(_psnr_shift(torch, a, b), 5)}
        out["frames_16bit_vs_all_f32"]["within_0p02_db"] = bool(out["frames_16bit_vs_all_f32"]["psnr_shift_db"] <= 0.02)
        del res
        # the fp16 flow engine against the float64 oracle on the committed full-size golden
        gpath = os.path.join(ROOT, "tests", "golden", "pwc_flow_1080p_sparse.npz")
        if os.path.isfile(gpath):
            sys.path.insert(0, os.path.join(ROOT, "tests"))
            from tests_support import make_flow_frames
            g = np.load(gpath)
            fa, fb = make_flow_frames(int(g["seed"]), 1080, 1920)
            pg = pwcnet.PWCNet(f"cuda:{local_rank}", precision="fp16")
            pg.set_weights(pwcnet.synthetic_weights(595000, flow_gain=float(g["flow_gain"])))
            fl = pg.flow_stack([torch.from_numpy(fa).to(dev), torch.from_numpy(fb).to(dev)])
            st = int(g["stride"])
            dg = fl[0, :, ::st, ::st].double().cpu().numpy() - g["flow_sparse"]
            out["flow_fp16_vs_oracle_golden_px"] = {"what": "one 1080p pair, both directions, every 8th LR pixel, vs the float64 oracle",
                                                    "max_abs": float(np.abs(dg).max()), "rms": float(np.sqrt((dg ** 2).mean()))}
            pg.close()
    finally:
        wl.flows, wl.warps = keep
        eng.close()
        torch.cuda.empty_cache()
    return out


def time_training_step(W, torch, local_rank, steps=5):
    """Row f4 (FISRnet.py:175-497): one training step of the reference's configuration -- batch 8 of 96x96 LR patches of 5
    frames (main.py:74), four weight-sharing passes forward and backward, seven loss terms, Adam -- on fisr_amd/train.py.
    Not part of `value`."""
    from fisr_amd import train
    r = np.random.default_rng(0)
    b, p, f32 = 8, 96, np.float32
    batch = train.to_device_batch(dict(
        data15=r.random((b, p, p, 15), dtype=f32), label21=r.random((b, 2 * p, 2 * p, 21), dtype=f32),
        flow16=(r.standard_normal((b, p, p, 16)) * 0.02).astype(f32), warp24=r.random((b, p, p, 24), dtype=f32),
        flow_ss2=(r.standard_normal((b, p, p, 8)) * 0.04).astype(f32), warp_ss2=r.random((b, p, p, 12), dtype=f32)),
        f"cuda:{local_rank}")
    net = train.TrainNet(W, device=f"cuda:{local_rank}")
    for _ in range(2):
        net.train_step(batch, 1e-4)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        loss, _ = net.train_step(batch, 1e-4)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    flop = 4 * b * FLOP_PER_LR_PX * p * p * 3      # forward + data gradient + weight gradient, four passes
    del net
    torch.cuda.empty_cache()
    return {"what": "one training step: batch 8 x 96x96 LR patches of 5 frames, 4 weight-sharing passes forward + backward, "
                    "7 loss terms, Adam (fp32)", "ms_per_step": round(dt * 1e3, 1), "samples_per_s": round(b / dt, 2),
            "tflops_3x_forward": round(flop / dt / 1e12, 1), "loss_after": round(float(loss), 4)}


def oracle_tile_check(net, torch):
    """One 544x992 reference tile through this engine against the fp64-oracle values committed on a sparse
    grid (tests/golden/model_544x992_sparse.npz, made by oracle/make_golden_fullsize.py), with the PSNR
    protocol of SURVEY.md 8c-ii: pseudo ground truth = oracle + Gaussian noise at the reference's published
    PSNRs (37.86 dB FI-SR channels, 48.07 dB SR channels, README.md:97)."""
    path = os.path.join(ROOT, "tests", "golden", "model_544x992_sparse.npz")
    if not os.path.isfile(path):
        return None
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from tests_support import make_full_size_input
    g = np.load(path)
    x = make_full_size_input(int(g["seed"]), 544, 992)
    _, _, l3 = net.model(torch.from_numpy(x).to(net.device), want_all=False)
    st = int(g["stride"])
    got = np.clip(l3[0, ::st, ::st, :].cpu().numpy().astype(np.float64), 0, 1)
    exp = np.clip(g["l3_sparse"].astype(np.float64), 0, 1)
    rng = np.random.default_rng(7)
    out = {"what": "one 544x992x29 tile vs the fp64 oracle on every 16th HR pixel, clipped to [0,1]",
           "max_abs": float(np.abs(got - exp).max()), "rms": float(np.sqrt(((got - exp) ** 2).mean()))}
    shifts = []
    for ch, db in ((slice(0, 3), 37.86), (slice(3, 6), 48.07), (slice(6, 9), 37.86)):
        sigma = 10 ** (-db / 20)
        gt = exp[..., ch] + rng.normal(0, sigma, exp[..., ch].shape)
        psnr = lambda a: 10 * np.log10(1.0 / ((a - gt) ** 2).mean())
        shifts.append(abs(psnr(got[..., ch]) - psnr(exp[..., ch])))
    out["psnr_shift_db_vs_oracle"] = float(max(shifts))
    out["within_0p02_db"] = bool(max(shifts) <= 0.02)
    return out


def _timeit(fn):
    t0 = time.perf_counter()
    fn()
    return time.perf_counter() - t0


def cpu_baselines(W, wl):
    """Bounded CPU sample on the host cores: ONE full 544x992 reference tile (2.85 TFLOP, 1/12 of a step's
    tiles) through (a) the C oracle, the parity checker itself ("port", naive OpenMP loops) and (b) the
    torch-CPU/oneDNN twin of the same graph (channels-last, all cores: the closest stand-in for the
    reference's TF-1.13 Eigen/MKL-DNN CPU path, which cannot be installed here)."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    # what the process may actually use: scheduler affinity and the cgroup CPU quota (a 128-core host with a quota of 16 cores
    # explains a "1.5 % of peak" oneDNN figure better than oneDNN does); threads are pinned to cores, one per core
    host = {"logical_cpus": os.cpu_count(), "affinity_cpus": len(os.sched_getaffinity(0)), "cgroup_cpu_quota_cores": None}
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            with open(path) as f:
                parts = f.read().split()
            if path.endswith("cpu.max"):
                if parts[0] != "max":
                    host["cgroup_cpu_quota_cores"] = round(int(parts[0]) / int(parts[1]), 2)
            else:
                q = int(parts[0])
                if q > 0:
                    with open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as f2:
                        host["cgroup_cpu_quota_cores"] = round(q / int(f2.read()), 2)
            break
        except (OSError, ValueError, IndexError):
            continue
    try:
        with open("/proc/cpuinfo") as f:
            txt = f.read()
        cores = {(b.split("physical id")[1].split("\n")[0], b.split("core id")[1].split("\n")[0]) for b in txt.split("\n\n") if "core id" in b}
        host["physical_cores"] = len(cores) or None
    except (OSError, IndexError):
        host["physical_cores"] = None
    usable = host["affinity_cpus"]
    if host["physical_cores"]:
        usable = min(usable, host["physical_cores"])
    if host["cgroup_cpu_quota_cores"]:
        usable = max(1, min(usable, int(host["cgroup_cpu_quota_cores"])))
    host["threads_used"] = usable
    os.environ.setdefault("OMP_NUM_THREADS", str(usable))
    os.environ.setdefault("OMP_PROC_BIND", "close")
    os.environ.setdefault("OMP_PLACES", "cores")
    import c_oracle
    from tests_support import make_full_size_input
    ch, cw = 544, 992
    x = make_full_size_input(4242, ch, cw)
    tile_flop = ch * cw * FLOP_PER_LR_PX
    tiles_per_stack = wl.flop_per_stack / tile_flop
    cpu_model = "unknown"
    try:
        with open("/proc/cpuinfo") as f:
            for ln in f:
                if ln.startswith("model name"):
                    cpu_model = ln.split(":", 1)[1].strip()
                    break
    except OSError:
        pass
    blob = c_oracle.pack_blob(W)
    # (explicit thread count: the OpenMP runtime was initialised when torch was imported, it no longer reads the environment)
    c_oracle.forward(x[:, :32, :32], blob, double=False, threads=usable)           # warm up threads
    t0 = time.perf_counter()
    c_oracle.forward(x, blob, double=False, threads=usable)
    dt = time.perf_counter() - t0
    cores = int(c_oracle.lib().fisr_oracle_num_threads())
    port = {"value": round(UNIQUE_PER_STACK / (dt * tiles_per_stack), 5), "unit": "frames/s", "cores": cores,
            "kind": "port", "cpu_model": cpu_model, "host": host,
            "sample": f"1x one full {ch}x{cw}x29 tile through oracle/fisr_oracle.c (fp32, OpenMP, {dt:.2f} s, "
                      f"{tile_flop / dt / 1e9:.1f} GFLOP/s), x{tiles_per_stack:.0f} tiles per 1080p stack"}
    onednn = None
    try:
        import torch
        import torch_cpu
        # one thread per usable physical core (affinity / cgroup quota respected); the SMT thread count oversubscribes the box
        # and is several times slower
        torch.set_num_threads(usable)
        # what these cores deliver on a dense fp32 GEMM (4096^3, best of 3): the yardstick for the conv figures below
        ga, gb = torch.randn(4096, 4096), torch.randn(4096, 4096)
        torch.mm(ga, gb)
        tg = min(_timeit(lambda: torch.mm(ga, gb)) for _ in range(3))
        host["sgemm_4096_gflops"] = round(2 * 4096 ** 3 / tg / 1e9, 1)
        del ga, gb
        Wt = torch_cpu.prepare_weights(W)
        torch_cpu.forward(x[:, :96, :96], Wt)                       # warm-up (primitive creation)
        qh, qw = 256, 480                                          # ~a quarter of the tile, multiples of 32
        t0 = time.perf_counter()
        torch_cpu.forward(x[:, :qh, :qw], Wt)                       # quarter tile: estimate before committing
        dq = time.perf_counter() - t0
        if dq * 4 <= 20.0:
            sh, sw, ts = ch, cw, []
            t_all = time.perf_counter()
            while len(ts) < 3 and (not ts or time.perf_counter() - t_all + ts[-1] < 25.0):
                t0 = time.perf_counter()
                torch_cpu.forward(x, Wt)
                ts.append(time.perf_counter() - t0)
            dt = float(np.median(ts))
            what = f"median of {len(ts)}x one full {ch}x{cw}x29 tile"
        else:
            sh, sw, dt = qh, qw, dq
            what = f"1x one {qh}x{qw}x29 sample (a full tile would exceed the bench's CPU budget)"
        s_flop = sh * sw * FLOP_PER_LR_PX
        onednn = {"value": round(UNIQUE_PER_STACK / (wl.flop_per_stack / (s_flop / dt)), 5), "unit": "frames/s",
                  "cores": int(torch.get_num_threads()), "kind": "port", "cpu_model": cpu_model, "host": host,
                  "sample": f"{what} through oracle/torch_cpu.py (torch {torch.__version__} CPU, oneDNN, fp32, "
                            f"channels-last, {dt:.2f} s, {s_flop / dt / 1e9:.1f} GFLOP/s), extrapolated by FLOPs to the "
                            f"{tiles_per_stack:.0f} tiles of a 1080p stack"}
    except Exception as e:                                          # noqa: BLE001 -- a baseline must not kill the bench line
        onednn = {"error": repr(e)}
    return port, onednn


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--precision", default="fp32", choices=sorted(PEAK))
    ap.add_argument("--others", default="fp32w,fp32d,bf16x3,f16f8,mixed",
                    help="further engines timed on rank 0 at N=1 under other_precisions ('' = none)")
    ap.add_argument("--parallelism", default="frame", choices=["frame", "tile"])
    ap.add_argument("--patch", default="2,2", help="tiles per frame, reference default (2,2)")
    ap.add_argument("--batch", default="stack", choices=["stack", "window", "tile"],
                    help="how many independent tiles go through one forward: the whole 5-frame stack "
                         "(3 windows x 4 tiles), one window (4 tiles) or one tile (the reference's schedule)")
    ap.add_argument("--input-path", default="fused", choices=["fused", "packed"],
                    help="fused: level inputs assembled straight from the frames / flows / warps (fisr_forward_frames, with --batch stack); "
                         "packed: fisr_pack_input -> tile slices -> fisr_forward")
    ap.add_argument("--layer-profile", default=None, help="write a per-layer timing table (JSON) to this path")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-parity", action="store_true", help="skip the oracle-tile checks (profiling runs: keeps the "
                    "per-kernel averages of rocprofv3 free of the small one-tile launches)")
    ap.add_argument("--no-gather", action="store_true", help="frame-parallel: skip the RCCL gather of the output frames")
    ap.add_argument("--no-flow", action="store_true", help="skip the cfg5 measurement (on-GPU PWC-Net flow + warp + FISRnet)")
    ap.add_argument("--no-train", action="store_true", help="skip the training-step measurement (row f4)")
    ap.add_argument("--dry-run", action="store_true", help="plan only: every rank reports the bytes it would hold on its device "
                    "(weights, activation arena of its forward batch, inputs, outputs, gather buffers) against the device's memory; "
                    "no step is run; exit status 3 if a rank's plan does not fit")
    args = ap.parse_args()
    patch = tuple(int(v) for v in args.patch.strip("()").split(","))

    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world == 1 and args.gpus > 1:
        raise SystemExit("launch with: python -m torch.distributed.run --nproc-per-node N bench.py --gpus N ...")
    # test hooks (a 1-GPU box cannot host two RCCL ranks): FISR_BENCH_BACKEND=gloo and FISR_BENCH_ONE_DEVICE=1
    # run the N > 1 control flow -- rendezvous, collectives, barriers, max-over-ranks -- with every rank on cuda:0
    backend = os.environ.get("FISR_BENCH_BACKEND", "nccl")
    if os.environ.get("FISR_BENCH_ONE_DEVICE"):
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)   # RCCL over xGMI
        else:
            dist.init_process_group(backend)

    from fisr_amd import dist as fdist
    from fisr_amd import weights
    from fisr_amd.fisrnet import FISRnet

    topo = group = None
    parallelism = args.parallelism
    n_tiles = patch[0] * patch[1]
    if parallelism == "tile" and (world == 1 or world % n_tiles):
        if rank == 0 and world > 1:
            print(f"# --parallelism tile needs a multiple of {n_tiles} ranks; world={world} runs frame-parallel", file=sys.stderr)
        parallelism = "frame"
    if parallelism == "tile":
        topo = fdist.TileTopology(patch, world, rank)
        group = topo.make_group()
    stacks = topo.n_groups if topo else world            # independent 5-frame stacks per step over the job
    stack_id = topo.group_index if topo else rank

    from fisr_amd import lib as _flib
    flib_version = _flib.lib().fisr_version().decode()       # carries a hash of csrc/: ties the line to the binary's sources
    W = weights.synthetic_weights(2020)
    free_before_weights = torch.cuda.mem_get_info(dev)[0]
    net = FISRnet(device=f"cuda:{local_rank}", precision=args.precision)
    net.set_weights(W)
    torch.cuda.synchronize(dev)
    weight_bytes = max(0, int(free_before_weights - torch.cuda.mem_get_info(dev)[0]))      # what the engine's packed weights took
    # ---- memory plan of this rank ("be right the first time a node shows up"): what the step will hold on the device,
    #      against what the device has.  A plan that does not fit is refused HERE, with numbers, on every rank -- not by an
    #      out-of-memory error inside the third collective of the first step.
    plan = memory_plan(torch, net, dev, patch, args.batch, parallelism, topo, world if (parallelism == "frame" and not args.no_gather) else 1, rank,
                       weight_bytes)
    plans = [plan]
    if world > 1:
        plans = [None] * world
        dist.all_gather_object(plans, plan)
    bad = [q for q in plans if not q["fits"]]
    if args.dry_run or bad:
        if rank == 0:
            print(json.dumps({"dry_run": True, "n_gpus": world, "parallelism": parallelism, "precision": args.precision, "batch": args.batch,
                              "fits": not bad, "per_rank": plans}), flush=True)
        if world > 1:
            dist.barrier()
            dist.destroy_process_group()
        sys.exit(3 if bad else 0)
    wl = Workload(torch, dev, stack_id, patch, args.batch, parallelism, topo, group,
                  gather_group_world=world if (parallelism == "frame" and not args.no_gather) else 1)
    wl.premake_warps(net)
    wl.input_path = args.input_path

    def sync():
        torch.cuda.synchronize(dev)
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize(dev)

    def drain():
        if wl.gather is not None:
            wl.gather.wait()                       # the timed region ends when the last gathered frames have ARRIVED on rank 0

    for _ in range(args.warmup):
        wl.step(net)
    # (the process's FIRST timing event costs tens of milliseconds -- the runtime sets up its profiling signals; r04: it sat inside
    #  the timed region and added ~45 ms to it, 9 ms per step at --steps 5.  Untimed work, like the warmup steps.)
    ev_w0, ev_w1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ev_w0.record(); ev_w1.record()
    drain()
    sync()
    _ = ev_w0.elapsed_time(ev_w1)
    if wl.gather is not None:
        wl.gather.gather_ms = 0.0
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    ev0.record()
    for _ in range(args.steps):
        wl.step(net)
    ev1.record()                                   # this rank's own compute stream is done here ...
    drain()                                        # ... the gathers that were still in flight here
    t_local = time.perf_counter() - t0
    sync()
    elapsed = time.perf_counter() - t0
    per_rank = None
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        # per-rank picture, so that a scaling curve explains itself: each rank's own step time (HIP events on its compute stream),
        # its wall time up to "my last gather has arrived", and the host time it spent inside the collective calls
        mine = {"rank": rank, "compute_ms_per_step": round(ev0.elapsed_time(ev1) / args.steps, 3),
                "wall_ms_per_step_incl_gather_drain": round(t_local / args.steps * 1e3, 3),
                "host_ms_in_gather_calls_per_step": round(wl.gather.gather_ms / args.steps, 3) if wl.gather is not None else None}
        per_rank = [None] * world
        dist.all_gather_object(per_rank, mine)
    value = stacks * UNIQUE_PER_STACK * args.steps / elapsed

    solo = rank == 0 and world == 1
    roofline = None
    if not args.no_roofline and parallelism == "frame":
        gw, wl.gather_world = wl.gather_world, 1          # the instrumented pass times kernels, not the collective
        roofline = roofline_pass(net, wl, args.precision, max(1, min(args.steps, 2)),
                                 args.layer_profile if rank == 0 else None)
        wl.gather_world = gw
        if roofline is not None and solo:
            roofline["hbm_bound_kernels"]["warp"] = time_warp(net, wl)

    parity_oracle = None
    other = {}
    cfg5 = None
    if solo and not args.no_flow:
        try:
            cfg5 = cfg5_pipeline(torch, dev, W, wl, net, local_rank)
        except Exception as e:                                      # noqa: BLE001 -- must not kill the headline line
            cfg5 = {"error": repr(e)}
    if solo:
        parity_oracle = None if args.no_parity else oracle_tile_check(net, torch)
        wl.step(net)
        torch.cuda.synchronize(dev)
        out_main = wl.full.clone()

        def compare(a, b, what):
            a, b = a.clamp(0, 1), b.clamp(0, 1)
            d = (a - b).double()
            mse = float((d * d).mean())
            return {"what": what + ", all 3x%dx%dx9 values of the same step, clipped to [0,1]" % (wl.h * 2, wl.w * 2),
                    "max_abs": float(d.abs().max()), "rms": mse ** 0.5,
                    "psnr_db": round(10 * np.log10(1.0 / mse), 2) if mse > 0 else None,
                    "uint8_mismatch_frac": float(((a * 255).to(torch.uint8) != (b * 255).to(torch.uint8)).double().mean())}

        for alt in [a for a in args.others.split(",") if a and a != "none" and a != args.precision]:
            eng = FISRnet(device=f"cuda:{local_rank}", precision=alt)
            eng.set_weights(W)
            wl.step(eng)                                      # warm-up + this engine's output
            torch.cuda.synchronize(dev)
            out_alt = wl.full.clone()
            nrep = max(2, min(args.steps, 5))
            t0 = time.perf_counter()
            for _ in range(nrep):
                wl.step(eng)
            torch.cuda.synchronize(dev)
            dt = (time.perf_counter() - t0) / nrep
            rl = None if args.no_roofline else roofline_pass(eng, wl, alt, 1)
            rec = {"value": round(UNIQUE_PER_STACK / dt, 3), "unit": "frames/s", "ms_per_step": round(dt * 1e3, 2),
                   "steps": nrep, "dtype": DTYPE[alt],
                   "roofline": {k: rl[k] for k in ("bound", "kernel", "achieved", "peak", "unit", "frac", "executed_per_algorithmic",
                                                   "algorithmic_tflops", "pmc_mfma_busy_frac", "traffic", "avg_launch_us", "launches",
                                                   "share_of_gpu_time", "all_conv_algorithmic_tflops", "all_conv_executed_frac",
                                                   "kernels") if k in rl} if rl else None,
                   "parity_vs_" + args.precision: compare(out_alt, out_main, f"{alt} vs the {args.precision} engine"),
                   "parity_vs_oracle": None if args.no_parity else oracle_tile_check(eng, torch)}
            other[alt] = rec
            eng.close()
            del eng, out_alt
            torch.cuda.empty_cache()

    training = None
    if solo and not args.no_train:
        try:
            training = time_training_step(W, torch, local_rank)
        except Exception as e:  # noqa: BLE001  (the headline must survive a failure of this extra)
            training = {"error": repr(e)}

    cpu_port = cpu_onednn = None
    if solo and not args.no_cpu_baseline:
        cpu_port, cpu_onednn = cpu_baselines(W, wl)

    if rank == 0:
        t0_ = wl.tiles[0]
        if parallelism == "tile":
            par = (f"tile-parallel: {stacks} stack(s) x {n_tiles} tiles, one tile per GPU, RCCL all-gather of 32-px "
                   "input halos and of uint8 output tiles inside each group of " + str(n_tiles))
            scaling = "strong" if stacks == 1 else "weak"
        elif world > 1:
            par = f"frame-parallel x{world} (one stack per GPU" + ("" if args.no_gather else ", RCCL gather of the uint8 output frames to rank 0") + ")"
            scaling = "weak"
        else:
            par, scaling = "single GPU", "weak"
        line = {
            "metric": "2K->4K FISR output frames/sec per node (unique frames of 5-frame 1080p stacks)",
            "value": round(value, 3), "unit": "frames/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 3),
            # (the same steps by HIP events on this rank's compute stream: what the GPU spent, beside the wall clock `value` is built from)
            "ms_per_step_hip_events": round(ev0.elapsed_time(ev1) / args.steps, 3),
            "host_enqueue_ms_per_step": round(t_local / args.steps * 1e3, 3),     # (host time to enqueue the steps: far below ms_per_step unless the host is the bottleneck)
            "higher_is_better": True, "scaling": scaling, "vs_baseline": None,
            "dtype": DTYPE[args.precision], "data": "synthetic",
            "config": {"workload": f"cfg2: 5-frame 1080x1920 stack -> 3 windows x {patch[0]}x{patch[1]} tiles of "
                                   f"{t0_.in_h}x{t0_.in_w}x29 (32-px halo) -> 7 unique {wl.h * 2}x{wl.w * 2} frames; "
                                   "pre-made flow+warp resident in HBM; synthetic seeded weights",
                       "engine": net.engine_description(),
                       "parallelism": par,
                       "input_path": ("fused: frames/flows/warps -> level inputs in one kernel per level (fisr_forward_frames)"
                                      if (args.input_path == "fused" and args.batch == "stack" and parallelism != "tile")
                                      else "packed: fisr_pack_input -> tile slices -> fisr_forward"),
                       "tiles_per_forward": 3 if parallelism == "tile" else {"stack": 3 * len(wl.tiles), "window": len(wl.tiles), "tile": 1}[args.batch],
                       "tflop_per_step": round(wl.flop_per_stack / 1e12, 3),
                       "raw_fps": round(stacks * 9 * args.steps / elapsed, 3),
                       "forwards_per_s": round(stacks * 3 * args.steps / elapsed, 3),
                       "algorithmic_tflops_whole_step": round(stacks * wl.flop_per_stack * args.steps / elapsed / 1e12, 2)},
            "per_rank": per_rank,
            "collective": (None if world == 1 else
                           {"what": "frame-parallel: asynchronous gather of every rank's uint8 output frames to rank 0 (side stream, double-buffered)"
                                    if parallelism == "frame" and not args.no_gather else
                                    ("tile-parallel: all-gather of the 32-px halo rings and of the uint8 output tiles inside each tile group"
                                     if parallelism == "tile" else "none"),
                            "bytes_per_rank_per_step": int(wl.yuv.numel()) if parallelism == "frame" and not args.no_gather else None,
                            "backend": backend}),
            "library": flib_version,
            "roofline": roofline, "cpu_baseline": cpu_port, "cpu_baseline_onednn": cpu_onednn,
            "parity_vs_oracle": parity_oracle, "cfg5": cfg5, "training_step": training,
            "other_precisions": other or None,
        }
        print(json.dumps(line), flush=True)
    if world > 1:
        # RCCL keeps an init banner in its C stdio buffer and flushes it at process exit, i.e. AFTER the
        # JSON line; leave without running the C-level exit handlers so the JSON line stays the last line.
        dist.barrier()
        sys.stdout.flush(); sys.stderr.flush()
        os._exit(0)


if __name__ == "__main__":
    if os.environ.get("FISR_BENCH_CPROFILE"):
        import cProfile, pstats
        pr = cProfile.Profile()
        try:
            pr.runcall(main)
        finally:
            pstats.Stats(pr, stream=sys.stderr).sort_stats("tottime").print_stats(18)
    else:
        main()
