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
`other_precisions`, `fp32d` is the same engine with the direct exact-fp32 kernel (the exact reference; `fp32w`, the all-F(2x2) engine of
round 2, is an A/B engine of the diagnostics build since r06).
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

from bench_report import (DTYPE, FLOP_PER_LR_PX, HBM_PEAK_GBPS, PEAK, UNIQUE_PER_STACK, cfg5_pipeline, cpu_baselines,
                          executed_per_algorithmic, expected_step_ms, memory_plan, oracle_tile_check, roofline_pass, time_training_step, time_warp,
                          _pmc_file, _pmc_key, _pmc_passes, _pmc_same_population)      # noqa: F401 -- (the last four: tests/test_host.py reaches them through this module)


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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--precision", default="fp32", choices=sorted(PEAK))
    ap.add_argument("--others", default="fp32d,bf16x3,f16f8,mixed",
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
    gather_on = parallelism == "frame" and not args.no_gather
    plan = memory_plan(torch, net, dev, patch, args.batch, parallelism, topo, world if gather_on else 1, rank, weight_bytes, args.input_path)
    # ... and what one step puts on the xGMI links, per collective, with the time the busiest link needs for it (fisr_amd/dist.py
    # collective_plan: pure arithmetic) -- printed next to the compute time of a step so that the first run on a node has an
    # expectation to be held against
    from fisr_amd import tiling as _tiling
    ph, pw = _tiling.crop_hw(1080, 1920, patch)
    plan["collectives_per_step"] = fdist.collective_plan(parallelism, patch, world, rank, ph, pw, gather=gather_on)
    plans = [plan]
    if world > 1:
        plans = [None] * world
        dist.all_gather_object(plans, plan)
    # --dry-run judges the plan against the device's size; a real run is refused only when the plan exceeds what is FREE on the
    # device right now (and merely warned about a plan that is large for the device)
    bad = [q for q in plans if not (q["fits"] if args.dry_run else q["fits_free_now"])]
    if not args.dry_run and rank == 0:
        for q in plans:
            if not q["fits"] and q["fits_free_now"]:
                print("# memory plan of rank %d is %.1f GB, more than its share of the device, but fits what is free now: running" % (q["rank"], q["total_bytes"] / 1e9), file=sys.stderr)
    if args.dry_run or bad:
        if rank == 0:
            print(json.dumps({"dry_run": True, "n_gpus": world, "parallelism": parallelism, "precision": args.precision, "batch": args.batch,
                              "fits": not bad, "compute_ms_per_step_expected": expected_step_ms(args.precision, parallelism, patch),
                              "link_model": {"xgmi_link_gbps_both_directions": fdist.XGMI_LINK_GBPS_BIDIR, "links_per_gpu": fdist.XGMI_LINKS_PER_GPU,
                                             "note": "link_ms_direct / link_ms_ring in collectives_per_step: one direction of one link = half of that"},
                              "per_rank": plans}), flush=True)
        if world > 1:
            dist.barrier()
            dist.destroy_process_group()
        sys.exit(3 if bad else 0)
    wl = Workload(torch, dev, stack_id, patch, args.batch, parallelism, topo, group, gather_group_world=world if gather_on else 1)
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
    # ================= the timed region: EXACTLY args.steps steps, barrier + synchronize on both sides (sync() above and below) =================
    t0 = time.perf_counter()
    ev0.record()
    for _ in range(args.steps):
        wl.step(net)
    ev1.record()                                   # this rank's own compute stream is done here ...
    drain()                                        # ... the gathers that were still in flight here
    t_local = time.perf_counter() - t0
    sync()
    elapsed = time.perf_counter() - t0
    # ================= end of the timed region; everything below (bench_report.py) is reporting on other passes =================
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
    extras_failed = []          # an extra that raised: named at the top level of the line, and the process exits with status 5 AFTER printing it
    if solo and not args.no_flow:
        try:
            cfg5 = cfg5_pipeline(torch, dev, W, wl, net, local_rank)
        except Exception as e:                                      # noqa: BLE001 -- must not kill the headline line
            cfg5 = {"error": repr(e)}
            extras_failed.append("cfg5: " + repr(e))
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
            try:
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
                       "roofline": {k: rl[k] for k in ("bound", "kernel", "achieved", "peak", "unit", "frac", "frac_algorithmic", "executed_per_algorithmic",
                                                       "algorithmic_tflops", "pmc_mfma_busy_frac", "traffic", "avg_launch_us", "launches",
                                                       "share_of_gpu_time", "all_conv_algorithmic_tflops", "all_conv_executed_frac",
                                                       "kernels") if k in rl} if rl else None,
                       "parity_vs_" + args.precision: compare(out_alt, out_main, f"{alt} vs the {args.precision} engine"),
                       "parity_vs_oracle": None if args.no_parity else oracle_tile_check(eng, torch)}
                other[alt] = rec
                eng.close()
                del eng, out_alt
                torch.cuda.empty_cache()
            except Exception as e:                                  # noqa: BLE001 -- the headline must survive; the failure is named in extras_failed
                other[alt] = {"error": repr(e)}
                extras_failed.append("other_precisions.%s: %r" % (alt, e))
                torch.cuda.empty_cache()

    training = None
    if solo and not args.no_train:
        try:
            training = time_training_step(W, torch, local_rank)
        except Exception as e:  # noqa: BLE001  (the headline must survive a failure of this extra)
            training = {"error": repr(e)}
            extras_failed.append("training_step: " + repr(e))

    cpu_port = cpu_onednn = None
    if solo and not args.no_cpu_baseline:
        cpu_port, cpu_onednn = cpu_baselines(W, wl)
        if isinstance(cpu_onednn, dict) and "error" in cpu_onednn:
            extras_failed.append("cpu_baseline_onednn: " + cpu_onednn["error"])

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
            # where roofline.traffic / pmc_mfma_busy_frac come from: a committed rocprofv3 PMC table of THIS library (replayed, not
            # measured in this process) or dropped -- also here at top level so that a consumer that keeps only known roofline keys sees it
            "counter_fields_source": (roofline or {}).get("counter_fields_source"),
            "parity_vs_oracle": parity_oracle, "cfg5": cfg5, "training_step": training,
            "other_precisions": other or None,
            "extras_failed": extras_failed,
        }
        print(json.dumps(line), flush=True)
    if world > 1:
        # RCCL keeps an init banner in its C stdio buffer and flushes it at process exit, i.e. AFTER the
        # JSON line; leave without running the C-level exit handlers so the JSON line stays the last line.
        dist.barrier()
        sys.stdout.flush(); sys.stderr.flush()
        os._exit(5 if extras_failed else 0)
    if extras_failed:
        sys.exit(5)


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
