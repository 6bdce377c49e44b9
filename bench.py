#!/usr/bin/env python3
"""Benchmark of the FISRnet hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W [--precision bf16x3|fp32|fp16]

A "step" is one pass of the hot path over one 5-frame 1080x1920 LR stack resident in HBM
(cfg2 of BASELINE.json): 3 sliding windows x [input assembly -> 2x2 tiles of 544x992x29 (32-px
halo) through the 138-conv FISRnet forward -> trim/stitch -> clip/quantise/YUV->RGB], i.e. the
inner loops of `FISRnet.test` / `FISR_for_video` (reference FISRnet.py:798-910), producing 9 raw
= 7 unique 2048x3840 frames.  Flows and warped frames are pre-made inputs as in cfg2.

Default arithmetic is `bf16x3` (every value a hi+lo bf16 pair, 3 bf16 MFMAs per product, fp32
accumulate: fp32-grade results, far inside the reference tolerance of +-0.02 dB); the exact-fp32
MFMA path is timed next to it (`fp32_exact`) and the two outputs are compared at full size
(`parity_vs_fp32`); `other_precisions` adds the fp16+fp8 split mode (`f16f8`, ~2^-15 per product,
the fastest mode inside the reference tolerance).  `--precision fp32|f16f8` changes the headline.

N > 1 (launched by torch.distributed.run, one rank per GPU): every rank processes its own stack
(frame-parallel, weights replicated, no data-path collective) -> weak scaling; timing is
barrier + synchronize on both sides, max over ranks.

One JSON line is printed by rank 0.  `roofline` is measured in a second, instrumented pass
(HIP events on the launch stream around every kernel, inside libfisr_hip.so); `cpu_baseline`
times the C oracle (oracle/fisr_oracle.c, OpenMP, fp32) on a bounded sample on rank 0 at N=1.
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
PEAK = {"fp32": 157.3, "fp16": 2500.0, "bf16x3": 2500.0, "f16f8": 2500.0}   # dense MFMA TFLOP/s (MI355X_MICROARCH.md)
MFMA_PER_PRODUCT = {"fp32": 1, "fp16": 1, "bf16x3": 3, "f16f8": 2.11}
DTYPE = {"fp32": "f32", "fp16": "f16 (f32 accumulate)",
         "bf16x3": "bf16x3 (values as hi+lo bf16 pairs, 3 bf16 MFMA per product, f32 accumulate)",
         "f16f8": "f16f8 (values as fp16 + fp8 remainder, fp16 MFMA + block-scaled fp8 MFMA for the cross terms, f32 accumulate)"}
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
    """cfg2 on one GPU: device-resident inputs + the step function of one FISRnet engine."""

    def __init__(self, torch, dev, rank, patch, batch):
        from fisr_amd import tiling
        self.torch, self.dev, self.patch, self.batch = torch, dev, patch, batch
        H0, W0 = 1080, 1920
        self.h, self.w = tiling.crop_hw(H0, W0, patch)
        frames_np, flows_np = synthetic_stack(100 + rank, H0, W0)
        self.frames = [torch.from_numpy(f).to(dev) for f in frames_np]
        self.flows = [torch.from_numpy(f).to(dev) for f in flows_np]
        self.warps = None
        self.tiles = tiling.plan_tiles(self.h, self.w, patch)
        self.full = torch.zeros((3, self.h * 2, self.w * 2, 9), dtype=torch.float32, device=dev)

    def premake_warps(self, net):
        # pre-made warps (cfg2): produced once with the warp kernel, outside the timed region
        self.warps = []
        for p in range(4):
            self.warps.append(net.warp(self.frames[p + 1], self.flows[2 * p]))
            self.warps.append(net.warp(self.frames[p], self.flows[2 * p + 1]))

    def step(self, net):
        torch, h, w = self.torch, self.h, self.w
        fr, fl, wp = self.frames, self.flows, self.warps
        outs = None
        if self.batch == "stack":
            # the 3 sliding windows (FISRnet.py:799) x 4 tiles (:847) are independent -> one batched forward
            inp = torch.cat([net.pack_input(fr[s:s + 3], fl[2 * s:2 * s + 4], wp[2 * s:2 * s + 4], h, w)
                             for s in range(3)], dim=0)
            net.forward_tiled(inp, self.patch, full=self.full)
            for s in range(3):
                outs = net.unpack_output(self.full[s])                 # FISRnet.py:883, 903-909
        else:
            for s in range(3):
                inp = net.pack_input(fr[s:s + 3], fl[2 * s:2 * s + 4], wp[2 * s:2 * s + 4], h, w)
                net.forward_tiled(inp, self.patch, full=self.full[s:s + 1], batch_tiles=(self.batch == "window"))
                outs = net.unpack_output(self.full[s])
        return outs

    @property
    def flop_per_stack(self):
        return 3 * sum(t.in_h * t.in_w for t in self.tiles) * FLOP_PER_LR_PX


def shader_clock_under_load(precision):
    """MHz the chip actually runs at while the dominant conv kernel executes: s_memtime ticks per 100 MHz
    s_memrealtime tick, one sample per workgroup of a 64->64 launch at the bench's batch (diagnostic entry
    fisr_bench_conv with FISR_TRACE_FILE).  The fast modes are power-throttled far below the nominal 2.4 GHz
    that the datasheet MFMA peak assumes."""
    import ctypes
    import tempfile
    from fisr_amd import lib as flib
    pid = {"fp32": 0, "fp16": 1, "bf16x3": 2, "f16f8": 3}[precision]
    with tempfile.TemporaryDirectory() as d:
        path = os.path.join(d, "trace.bin")
        os.environ["FISR_TRACE_FILE"] = path
        try:
            us = ctypes.c_double()
            rc = flib.lib().fisr_bench_conv(pid, 12, 544, 992, 64, 64, 3, 1, 3, ctypes.byref(us))
        finally:
            os.environ.pop("FISR_TRACE_FILE", None)
        if rc != 0 or not os.path.isfile(path):
            return None
        a = np.fromfile(path, dtype=np.uint64).reshape(-1, 8).astype(np.int64)
    a = a[(a[:, 2] > a[:, 0]) & (a[:, 6] > a[:, 5])]
    if not len(a):
        return None
    return float(np.median((a[:, 2] - a[:, 0]) / (a[:, 6] - a[:, 5]) * 100.0))


def roofline_pass(net, wl, precision, reps, layer_profile=None):
    torch = wl.torch
    net.profile(1)
    for _ in range(reps):
        wl.step(net)
    torch.cuda.synchronize(wl.dev)
    prof = net.profile_read()
    net.profile(0)
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
    dom = max(convs, key=lambda p: p["ms"])
    tot_ms = sum(p["ms"] for p in prof)
    ach = dom["flops"] / (dom["ms"] * 1e-3) / 1e12
    # HBM bytes per launch from the rocprofv3 PMC passes of this same command (scripts/gpu_profile.sh ->
    # profiles/pmc_traffic.json; FETCH_SIZE x2 gfx950 correction + WRITE_SIZE, separate passes)
    traffic = None
    try:
        with open(os.path.join(ROOT, "profiles", "pmc_traffic.json")) as f:
            pmc = json.load(f)
        tname = {"f32": "float", "f16": "_Float16", "bf16x3": "bsplit", "f16f8": "fsplit"}[dom["name"].split("<")[1].split(",")[0]]
        nt = dom["name"].split("NT")[1][0]
        key = f"conv3x3_mfma_kernel<{tname}, {nt}, {'true' if 'f32out' in dom['name'] else 'false'}"
        hits = [v for k, v in pmc.items() if k.startswith(key)]
        if hits:
            traffic = round(max(hits, key=lambda v: v.get("dispatches", 0))["hbm_bytes_per_launch"] / 1e9, 4)
    except (OSError, KeyError, IndexError, ValueError):
        traffic = None
    return {"bound": "mfma", "kernel": dom["name"], "achieved": round(ach, 2), "peak": PEAK[precision],
            "unit": "TFLOP/s", "frac": round(ach / PEAK[precision], 4), "traffic": traffic,
            "traffic_unit": "GB of HBM per launch (rocprofv3 PMC, profiles/pmc_traffic.json)",
            "mfma_issue_frac": round(MFMA_PER_PRODUCT[precision] * ach / PEAK[precision], 4),
            "avg_launch_us": round(dom["ms"] * 1e3 / dom["launches"], 2), "launches": int(dom["launches"]),
            "algorithmic_gflop_per_launch": round(dom["flops"] / dom["launches"] / 1e9, 2),
            "algorithmic_gbyte_per_launch": round(dom["bytes"] / dom["launches"] / 1e9, 4),
            "share_of_gpu_time": round(dom["ms"] / tot_ms, 4),
            "all_conv_tflops": round(sum(p["flops"] for p in convs) / (sum(p["ms"] for p in convs) * 1e-3) / 1e12, 2),
            "kernels": {p["name"]: {"ms": round(p["ms"], 3), "launches": int(p["launches"])} for p in prof}}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--precision", default="bf16x3", choices=["bf16x3", "f16f8", "fp32", "fp16"])
    ap.add_argument("--patch", default="2,2", help="tiles per frame, reference default (2,2)")
    ap.add_argument("--batch", default="stack", choices=["stack", "window", "tile"],
                    help="how many independent tiles go through one forward: the whole 5-frame stack "
                         "(3 windows x 4 tiles), one window (4 tiles) or one tile (the reference's schedule)")
    ap.add_argument("--layer-profile", default=None, help="write a per-layer timing table (JSON) to this path")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-fp32-ref", action="store_true", help="skip the exact-fp32 comparison run")
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
    # run the N > 1 control flow -- rendezvous, barriers, max-over-ranks -- with every rank on cuda:0
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

    from fisr_amd import weights
    from fisr_amd.fisrnet import FISRnet

    W = weights.synthetic_weights(2020)
    net = FISRnet(device=f"cuda:{local_rank}", precision=args.precision)
    net.set_weights(W)
    wl = Workload(torch, dev, rank, patch, args.batch)
    wl.premake_warps(net)

    def sync():
        torch.cuda.synchronize(dev)
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize(dev)

    for _ in range(args.warmup):
        wl.step(net)
    sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        wl.step(net)
    sync()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    value = world * UNIQUE_PER_STACK * args.steps / elapsed

    roofline = None
    if not args.no_roofline:
        roofline = roofline_pass(net, wl, args.precision, max(1, min(args.steps, 2)),
                                 args.layer_profile if rank == 0 else None)
        if roofline is not None and rank == 0:
            mhz = shader_clock_under_load(args.precision)
            if mhz:
                roofline["shader_clock_mhz_under_load"] = round(mhz)
                roofline["mfma_issue_frac_at_that_clock"] = round(roofline["mfma_issue_frac"] * 2400.0 / mhz, 4)

    fp32_exact = parity = None
    other = {}
    if rank == 0 and world == 1 and not args.no_fp32_ref:
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

        out_fp32 = out_main if args.precision == "fp32" else None
        for alt in ("fp32", "f16f8", "bf16x3"):
            if alt == args.precision:
                continue
            eng = FISRnet(device=f"cuda:{local_rank}", precision=alt)
            eng.set_weights(W)
            wl.step(eng)                                      # warm-up + this precision's output
            torch.cuda.synchronize(dev)
            out_alt = wl.full.clone()
            t0 = time.perf_counter()
            for _ in range(2):
                wl.step(eng)
            torch.cuda.synchronize(dev)
            dt = (time.perf_counter() - t0) / 2
            rl = roofline_pass(eng, wl, alt, 1)
            rec = {"value": round(UNIQUE_PER_STACK / dt, 3), "unit": "frames/s", "ms_per_step": round(dt * 1e3, 2),
                   "dtype": DTYPE[alt],
                   "roofline": {k: rl[k] for k in ("kernel", "achieved", "peak", "frac", "mfma_issue_frac", "avg_launch_us")} if rl else None}
            if alt == "fp32":
                out_fp32 = out_alt
                fp32_exact = rec
                parity = compare(out_main, out_fp32, f"{args.precision} vs exact fp32")
            else:
                if out_fp32 is not None:
                    rec["parity_vs_fp32"] = compare(out_alt, out_fp32, f"{alt} vs exact fp32")
                other[alt] = rec
            eng.close()
            del eng, out_alt
            torch.cuda.empty_cache()

    cpu_baseline = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        sys.path.insert(0, os.path.join(ROOT, "oracle"))
        import c_oracle
        ch, cw = 256, 384
        x = np.random.default_rng(3).random((1, ch, cw, 29)).astype(np.float32)
        blob = c_oracle.pack_blob(W)
        c_oracle.forward(x[:, :32, :32], blob, double=False)           # warm up threads
        t0 = time.perf_counter()
        reps = 0
        while reps < 1 or (time.perf_counter() - t0 < 10.0 and reps < 50):
            c_oracle.forward(x, blob, double=False)
            reps += 1
        dt = (time.perf_counter() - t0) / reps
        cpu_flops = ch * cw * FLOP_PER_LR_PX / dt
        cores = c_oracle.lib().fisr_oracle_num_threads()
        cpu_baseline = {"value": round(UNIQUE_PER_STACK / (wl.flop_per_stack / cpu_flops), 5), "unit": "frames/s",
                        "cores": int(cores), "kind": "port",
                        "sample": f"{reps}x one {ch}x{cw}x29 forward through oracle/fisr_oracle.c (fp32, OpenMP, "
                                  f"{cpu_flops / 1e9:.1f} GFLOP/s), extrapolated by FLOPs to the tiled 1080p stack"}

    if rank == 0:
        t0_ = wl.tiles[0]
        line = {
            "metric": "2K->4K FISR output frames/sec per node (unique frames of 5-frame 1080p stacks)",
            "value": round(value, 3), "unit": "frames/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": DTYPE[args.precision], "data": "synthetic",
            "config": {"workload": f"cfg2: 5-frame 1080x1920 stack -> 3 windows x {patch[0]}x{patch[1]} tiles of "
                                   f"{t0_.in_h}x{t0_.in_w}x29 (32-px halo) -> 7 unique {wl.h * 2}x{wl.w * 2} frames; "
                                   "pre-made flow+warp resident in HBM; synthetic seeded weights",
                       "parallelism": f"frame-parallel x{world}" if world > 1 else "single GPU",
                       "tiles_per_forward": {"stack": 3 * len(wl.tiles), "window": len(wl.tiles), "tile": 1}[args.batch],
                       "tflop_per_step": round(wl.flop_per_stack / 1e12, 3),
                       "raw_fps": round(world * 9 * args.steps / elapsed, 3),
                       "forwards_per_s": round(world * 3 * args.steps / elapsed, 3),
                       "achieved_tflops_whole_step": round(world * wl.flop_per_stack * args.steps / elapsed / 1e12, 2)},
            "roofline": roofline, "cpu_baseline": cpu_baseline, "fp32_exact": fp32_exact, "parity_vs_fp32": parity,
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
    main()
