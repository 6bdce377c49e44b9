"""Reporters of bench.py: everything the bench line carries BESIDE the timed headline -- the instrumented roofline pass and the
counter table it joins, the memory / collective plan of `--dry-run`, cfg5 (flow + warp + network), the training step, the
oracle-tile parity check and the two CPU baselines.  None of it runs inside bench.py's timed region (bench.py: `Workload`, `main`);
split out of bench.py in round 5 so that the timed step reads in one screen.  See DESIGN.md section 5."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))

FLOP_PER_LR_PX = 5288328.0          # SURVEY.md 8d / BASELINE.md section 2 (2 x 2 644 164 MAC)
HBM_PEAK_GBPS = 8000.0              # MI355X_MICROARCH.md (spec; ~6300 achievable with a float4 copy)
# dense MFMA peak of the instruction class each engine issues (MI355X_MICROARCH.md), TFLOP/s
PEAK = {"fp32": 157.3, "fp32d": 157.3, "fp32w": 157.3, "fp32w4": 157.3, "fp16": 2500.0, "bf16x3": 2500.0, "f16f8": 2500.0, "mixed": 2500.0,
        "f16f8r": 2500.0, "mixedr": 2500.0}      # (the two A/B engines on round 1's register-staged kernels)
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
DTYPE["f16f8r"] = DTYPE["f16f8"] + " [register-staged kernel, A/B]"
DTYPE["mixedr"] = DTYPE["mixed"] + " [register-staged kernels, A/B]"
UNIQUE_PER_STACK = 7                # 3 windows x 3 frames, overlaps counted once (FISRnet.py:913-920)


PMC_SRC = None          # how the counter-derived fields of this line are labelled (set by _pmc_table)


def _src_of(library):
    """'... src <16 hex digits> ...' -> the hex digits"""
    import re
    m = re.search(r"src ([0-9a-f]{16})", library or "")
    return m.group(1) if m else None


def _pmc_table_checked(running_library, engine=None):
    """profiles/pmc_traffic[_<engine>].json, or {} when it was measured on another library (its `_meta.library` names the build): a
    counter of another kernel build under this build's name would read as measured."""
    global PMC_SRC
    pmc = _pmc_table(engine)
    fname = os.path.relpath(_pmc_file(engine), ROOT)
    meta = pmc.pop("_meta", None) or {}
    have, want = _src_of(meta.get("library")), _src_of(running_library)
    if not pmc:
        PMC_SRC = {"file": fname, "status": "absent"}
        return {}, meta
    if have is None or have != want:
        PMC_SRC = {"file": fname, "status": "dropped: measured on library src %s, running src %s" % (have, want)}
        return {}, meta
    PMC_SRC = {"file": fname, "status": "same library", "library_src": have,
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


def memory_plan(torch, net, dev, patch, batch, parallelism, topo, gather_world, rank, weight_bytes, input_path="fused"):
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
    fused = input_path == "fused" and batch == "stack" and parallelism != "tile"
    out_f32 = 3 * (2 * h) * (2 * w) * 9 * 4
    out_u8 = 3 * (2 * h) * (2 * w) * 9
    items = {
        "weights_packed_measured": int(weight_bytes),
        "activation_arena": arena,
        "forward_batch": [n_fwd, th, tw],
        "inputs_frames_flows_warps": 5 * H0 * W0 * 3 + 8 * H0 * W0 * 2 * 4 + 8 * H0 * W0 * 3 * 4,
        # (the fused input path -- fisr_forward_frames, the default with --batch stack -- never allocates the packed tensor or the
        #  tile slices: the level inputs are cut straight out of the source planes)
        "packed_input_3_windows": 0 if fused else 3 * h * w * 29 * 4,
        "tile_batch_in_out": (0 if fused else n_fwd * th * tw * 29 * 4) + n_fwd * 4 * th * tw * 9 * 4,
        "stitched_output_f32": out_f32, "output_yuv_rgb_u8": 3 * out_u8,
        "gather_send_buffers": 2 * out_u8 if gather_world > 1 else 0,
        "gather_receive_buffer": gather_world * out_u8 if gather_world > 1 and rank == 0 else 0,
    }
    total = sum(v for k, v in items.items() if isinstance(v, int))
    free_b, total_b = torch.cuda.mem_get_info(dev)
    free_b += int(weight_bytes)                               # (the weights are already resident)
    sharing = int(os.environ.get("FISR_BENCH_ONE_DEVICE", "0") == "1") and int(os.environ.get("WORLD_SIZE", "1")) or 1
    budget = total_b // max(sharing, 1)                        # (test mode: all ranks share device 0)
    # `fits`: the plan against the device's size (what --dry-run answers); `fits_free_now`: against what is free on the device at
    # this moment, weights counted back in -- the test a real run refuses on (a plan near an artificial per-rank share of a shared
    # device, or beside another process's memory, is decided by what is actually free)
    return {"rank": rank, "device": str(dev), "bytes": items, "total_bytes": total, "device_total_bytes": int(total_b),
            "device_free_bytes_now": int(free_b), "ranks_sharing_device": sharing,
            "fits": bool(total * 1.05 <= budget),              # 5 % for the allocator's rounding
            "fits_free_now": bool(total * 1.05 <= free_b // max(sharing, 1))}


def expected_step_ms(precision, parallelism, patch):
    """What one step of this engine took on one MI355X in the latest committed bench line (profiles/r??_bench_default.json), for
    `--dry-run` to print beside the link times: frame-parallel ranks run the whole step, a tile-parallel rank its tile of the 3
    windows (1 / tiles of the step's convolutions).  None when no committed line has the engine."""
    import glob
    best = None
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r??_bench_default.json"))):
        try:
            with open(path) as f:
                d = json.load(f)
        except (OSError, ValueError):
            continue
        ms = d.get("ms_per_step") if precision == "fp32" else ((d.get("other_precisions") or {}).get(precision) or {}).get("ms_per_step")
        if ms:
            best = {"ms": float(ms), "source": os.path.relpath(path, ROOT)}
    if best is None:
        return None
    if parallelism == "tile":
        best["ms"] = round(best["ms"] / (patch[0] * patch[1]), 2)
        best["note"] = "a rank's tile of the 3 windows: the single-GPU step over the tiles of the plan"
    return best


def _pmc_file(engine=None):
    """profiles/pmc_traffic_<engine>.json when that engine was profiled ALONE (r05: `mixed` shares kernel names with f16f8 / fp16, so a
    table of all engines in one run can never match its launch population), else the table of the all-engines run"""
    own = os.path.join(ROOT, "profiles", "pmc_traffic_%s.json" % engine) if engine else None
    return own if own and os.path.isfile(own) else os.path.join(ROOT, "profiles", "pmc_traffic.json")


def _pmc_table(engine=None):
    """HBM bytes per launch of every kernel, from the rocprofv3 PMC passes of this same command
    (scripts/gpu_profile.sh -> profiles/pmc_traffic[_<engine>].json; FETCH_SIZE x2 gfx950 correction + WRITE_SIZE,
    separate passes as MI355X_MICROARCH.md prescribes)."""
    try:
        with open(_pmc_file(engine)) as f:
            return json.load(f)
    except (OSError, ValueError):
        return {}


def _pmc_key(name):
    if name.startswith("conv3x3_dma_fs"):       # "conv3x3_dma_fs<f16f8,tw32,relu_in,nores>" -> conv3x3_dma_fs_kernel<32, true, false, false>
        f = name.split("<")[1].rstrip(">").split(",")
        return "conv3x3_dma_fs_kernel<%s, %s, %s, %s>" % (f[1][2:], "true" if f[2] == "relu_in" else "false",
                                                         "false" if f[3] == "nores" else "true", "true" if f[3] == "res+pool" else "false")
    # (keys are PREFIXES of the demangled names: no closing '>' -- the kernels grew template parameters behind these, r05: SHARE, NT)
    if name.startswith("conv3x3_dma"):
        return "conv3x3_dma_f16_kernel<false,"
    if name.startswith("conv3x3_wf4"):
        if "res+pool" in name:
            return "conv3x3_wf4_kernel<false, true, true, false, false"
        if "up2" in name:
            return "conv3x3_wf4_kernel<false, false, false, true, false"
        # (5th parameter: the GENERAL instantiation of the flow network)
        return "conv3x3_wf4_kernel<%s, %s, false, false, false" % ("true" if "relu_in" in name else "false", "false" if "nores" in name else "true")
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
    pmc, pmc_meta = _pmc_table_checked(_fl.lib().fisr_version().decode(), precision)
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
        hits = [v for k, v in pmc.items() if k.startswith(key)]
        if len(hits) > 1 and not _pmc_same_population(e, pmc_passes, lps.get(name, 0)):
            # one profile class, several instantiations (the fp32 heads: head_conv_f32_kernel<2> and <3>): the class's population is
            # their sum, its per-launch figures their dispatch-weighted means
            tot = sum(v.get("dispatches", 0) for v in hits)
            if tot:
                m = {"dispatches": tot}
                for f in set().union(*[set(v) for v in hits]) - {"dispatches"}:
                    vals = [(v[f], v.get("dispatches", 0)) for v in hits if isinstance(v.get(f), (int, float))]
                    if len(vals) == len(hits):
                        m[f] = sum(a * b for a, b in vals) / tot
                e = m
        if not _pmc_same_population(e, pmc_passes, lps.get(name, 0)):
            dropped.append("%s: %s dispatches in the PMC run, %.1f launches per step here, %s passes of this engine" % (name, e.get("dispatches", 0), lps.get(name, 0), pmc_passes or "no whole number of"))
            return None
        return e

    def conv_field(name, field, scale=1.0):
        try:
            e = pmc_entry(_pmc_key(name), name)
        except (KeyError, IndexError, ValueError):
            e = None
        return None if e is None or e.get(field) is None else round(e[field] * scale, 4)      # (None: a row summarize_prof.py nulled)

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
        if e and e.get("valu_issue_frac") is not None:      # (the heads: vector-ALU kernels -- their floor is this, not the bytes; profiles/r05_heads_pmc.txt)
            rec["pmc_valu_issue_frac"] = round(e["valu_issue_frac"], 3)
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
          # the other reading of the same launch: ALGORITHMIC (direct-convolution) FLOPs per second over the peak -- above 1 for the
          # Winograd kernels, which execute a quarter (F(4x4)) / 4/9 (F(2x2)) of the direct algorithm's multiplies
          "frac_algorithmic": round(ach / peak, 4),
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
