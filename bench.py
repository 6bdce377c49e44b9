#!/usr/bin/env python3
"""Benchmark of the FISRnet hot path on MI355X (see DESIGN.md "Measurement").

    python bench.py --gpus N --steps K --warmup W [--precision fp32|fp16]

A "step" is one pass of the hot path over one 5-frame 1080x1920 LR stack resident in HBM
(cfg2 of BASELINE.json): 3 sliding windows x [input assembly -> 2x2 tiles of 544x992x29 (32-px
halo) through the 138-conv FISRnet forward -> trim/stitch -> clip/quantise/YUV->RGB], i.e. the
inner loops of `FISRnet.test` / `FISR_for_video` (reference FISRnet.py:798-910), producing 9 raw
= 7 unique 2048x3840 frames.  Flows and warped frames are pre-made inputs as in cfg2.

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
PEAK = {"fp32": 157.3, "fp16": 2500.0, "bf16x3": 2500.0}   # dense MFMA TFLOP/s, MI355X_MICROARCH.md chip table
UNIQUE_PER_STACK = 7                # 3 windows x 3 frames, overlaps counted once (FISRnet.py:913-920)


def synthetic_stack(seed, H=1080, W=1920):
    """5 LR YUV frames (uint8), 8 flows (float32 px), 8 warped frames (float32 0..255)."""
    rng = np.random.default_rng(seed)
    coarse = rng.integers(0, 256, (5, H // 8, W // 8, 3)).astype(np.float32)
    frames = np.repeat(np.repeat(coarse, 8, axis=1), 8, axis=2)
    frames = np.clip(frames + rng.normal(0, 6, frames.shape), 0, 255).astype(np.uint8)
    fc = rng.normal(0, 4, (8, H // 32 + 1, W // 32 + 1, 2)).astype(np.float32)
    flows = np.repeat(np.repeat(fc, 32, axis=1), 32, axis=2)[:, :H, :W, :].copy()
    return frames, flows


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--precision", default="fp32", choices=["fp32", "fp16", "bf16x3"])
    ap.add_argument("--patch", default="2,2", help="tiles per frame, reference default (2,2)")
    ap.add_argument("--batch", default="stack", choices=["stack", "window", "tile"],
                    help="how many independent tiles go through one forward: the whole 5-frame stack "
                         "(3 windows x 4 tiles), one window (4 tiles) or one tile (the reference's schedule)")
    ap.add_argument("--layer-profile", default=None, help="write a per-layer timing table (JSON) to this path")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    args = ap.parse_args()
    patch = tuple(int(v) for v in args.patch.strip("()").split(","))

    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with: python -m torch.distributed.run --nproc-per-node N bench.py --gpus N ...")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=dev)   # RCCL over xGMI

    from fisr_amd import tiling, weights
    from fisr_amd.fisrnet import FISRnet

    net = FISRnet(device=f"cuda:{local_rank}", precision=args.precision)
    net.set_weights(weights.synthetic_weights(2020))

    H0, W0 = 1080, 1920
    h, w = tiling.crop_hw(H0, W0, patch)
    frames_np, flows_np = synthetic_stack(100 + rank, H0, W0)
    frames = [torch.from_numpy(f).to(dev) for f in frames_np]
    flows = [torch.from_numpy(f).to(dev) for f in flows_np]
    # pre-made warps (cfg2): produced once with the warp kernel, outside the timed region
    warps = []
    for p in range(4):
        warps.append(net.warp(frames[p + 1], flows[2 * p]))
        warps.append(net.warp(frames[p], flows[2 * p + 1]))
    full = torch.zeros((3, h * 2, w * 2, 9), dtype=torch.float32, device=dev)

    def step():
        outs = None
        if args.batch == "stack":
            # the 3 sliding windows (FISRnet.py:799) x 4 tiles (:847) are independent -> one batched forward
            inp = torch.cat([net.pack_input(frames[s:s + 3], flows[2 * s:2 * s + 4], warps[2 * s:2 * s + 4], h, w)
                             for s in range(3)], dim=0)
            net.forward_tiled(inp, patch, full=full)
            for s in range(3):
                outs = net.unpack_output(full[s])            # FISRnet.py:883, 903-909
        else:
            for s in range(3):
                inp = net.pack_input(frames[s:s + 3], flows[2 * s:2 * s + 4], warps[2 * s:2 * s + 4], h, w)
                net.forward_tiled(inp, patch, full=full[s:s + 1], batch_tiles=(args.batch == "window"))
                outs = net.unpack_output(full[s])
        return outs

    def sync():
        torch.cuda.synchronize(dev)
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize(dev)

    for _ in range(args.warmup):
        step()
    sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    sync()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    tiles = tiling.plan_tiles(h, w, patch)
    lr_px_per_stack = 3 * sum(t.in_h * t.in_w for t in tiles)
    flop_per_stack = lr_px_per_stack * FLOP_PER_LR_PX
    value = world * UNIQUE_PER_STACK * args.steps / elapsed

    roofline = None
    if not args.no_roofline:
        net.profile(True)
        for _ in range(max(1, min(args.steps, 2))):
            step()
        torch.cuda.synchronize(dev)
        prof = net.profile_read()
        net.profile(False)
        if args.layer_profile and rank == 0:
            net.profile(2)
            step()
            torch.cuda.synchronize(dev)
            layers = net.profile_read()
            net.profile(False)
            for p_ in layers:
                p_["tflops"] = round(p_["flops"] / (p_["ms"] * 1e-3) / 1e12, 2) if p_["ms"] > 0 and p_["flops"] else 0.0
                p_["us_per_launch"] = round(p_["ms"] * 1e3 / max(1, p_["launches"]), 1)
            os.makedirs(os.path.dirname(os.path.abspath(args.layer_profile)), exist_ok=True)
            with open(args.layer_profile, "w") as f:
                json.dump(sorted(layers, key=lambda q: -q["ms"]), f, indent=1)
        convs = [p for p in prof if p["name"].startswith("conv3x3") and p["launches"]]
        if convs:
            dom = max(convs, key=lambda p: p["ms"])
            tot_ms = sum(p["ms"] for p in prof)
            ach = dom["flops"] / (dom["ms"] * 1e-3) / 1e12
            roofline = {"bound": "mfma", "kernel": dom["name"], "achieved": round(ach, 2),
                        "peak": PEAK[args.precision], "unit": "TFLOP/s",
                        "frac": round(ach / PEAK[args.precision], 4), "traffic": None,
                        "avg_launch_us": round(dom["ms"] * 1e3 / dom["launches"], 2),
                        "launches": int(dom["launches"]),
                        "mfma_issue_frac": round((3 if args.precision == "bf16x3" else 1) * ach / PEAK[args.precision], 4),
                        "share_of_gpu_time": round(dom["ms"] / tot_ms, 4),
                        "all_conv_tflops": round(sum(p["flops"] for p in convs) / (sum(p["ms"] for p in convs) * 1e-3) / 1e12, 2),
                        "kernels": {p["name"]: {"ms": round(p["ms"], 3), "launches": int(p["launches"])} for p in prof}}

    cpu_baseline = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        sys.path.insert(0, os.path.join(ROOT, "oracle"))
        import c_oracle
        ch, cw = 128, 192
        x = np.random.default_rng(3).random((1, ch, cw, 29)).astype(np.float32)
        blob = c_oracle.pack_blob(weights.synthetic_weights(2020))
        c_oracle.forward(x[:, :32, :32], blob, double=False)           # warm up threads
        t0 = time.perf_counter()
        reps = 0
        while reps < 1 or (time.perf_counter() - t0 < 10.0 and reps < 50):
            c_oracle.forward(x, blob, double=False)
            reps += 1
        dt = (time.perf_counter() - t0) / reps
        cpu_flops = ch * cw * FLOP_PER_LR_PX / dt
        cores = c_oracle.lib().fisr_oracle_num_threads()
        cpu_baseline = {"value": round(UNIQUE_PER_STACK / (flop_per_stack / cpu_flops), 5), "unit": "frames/s",
                        "cores": int(cores), "kind": "port",
                        "sample": f"{reps}x one {ch}x{cw}x29 forward through oracle/fisr_oracle.c (fp32, OpenMP, "
                                  f"{cpu_flops / 1e9:.1f} GFLOP/s), extrapolated by FLOPs to the tiled 1080p stack"}

    if rank == 0:
        line = {
            "metric": "2K->4K FISR output frames/sec per node (unique frames of 5-frame 1080p stacks)",
            "value": round(value, 3), "unit": "frames/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": {"fp32": "f32", "fp16": "f16 (f32 accumulate)",
                      "bf16x3": "bf16x3 (values as hi+lo bf16 pairs, 3 bf16 MFMA per product, f32 accumulate)"}[args.precision], "data": "synthetic",
            "config": {"workload": f"cfg2: 5-frame 1080x1920 stack -> 3 windows x {patch[0]}x{patch[1]} tiles of "
                                   f"{tiles[0].in_h}x{tiles[0].in_w}x29 (32-px halo) -> 7 unique 2048x3840 frames; "
                                   "pre-made flow+warp resident in HBM; synthetic seeded weights",
                       "parallelism": f"frame-parallel x{world}" if world > 1 else "single GPU",
                       "tiles_per_forward": {"stack": 3 * len(tiles), "window": len(tiles), "tile": 1}[args.batch],
                       "tflop_per_step": round(flop_per_stack / 1e12, 3),
                       "raw_fps": round(world * 9 * args.steps / elapsed, 3),
                       "forwards_per_s": round(world * 3 * args.steps / elapsed, 3),
                       "achieved_tflops_whole_step": round(world * flop_per_stack * args.steps / elapsed / 1e12, 2)},
            "roofline": roofline, "cpu_baseline": cpu_baseline,
        }
        print(json.dumps(line))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
