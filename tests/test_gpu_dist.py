"""GPU tests of the tile-parallel path with the REAL engine (run with `-m gpu` on the MI355X box).

A 1-GPU box cannot host several RCCL ranks, so the ranks are gloo processes that share cuda:0 (the
collectives of fisr_amd/dist.py stage through the host under gloo and go device-to-device under RCCL;
everything else -- core packing, halo assembly, fisr_forward on the halo'd tile, trim, on-GPU quantise,
tile gather -- is the code an 8-GPU node runs).  Reference unit of work: FISRnet.py:798-799, 847-880,
utils.py:118-159.
"""
import os
import socket

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")
import torch.distributed as dist  # noqa: E402
import torch.multiprocessing as mp  # noqa: E402


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


# (first in the file on purpose: a box with two GPUs reaches the only test that moves bytes BETWEEN devices over RCCL before the long ones)
def _comm_worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from fisr_amd import dist as fdist
        torch.cuda.set_device(rank)
        comm = fdist.FisrComm.from_torch_group(rank)
        mine = torch.full((4, 1000), rank + 1, dtype=torch.uint8, device=f"cuda:{rank}")
        got = comm.allgather(mine)
        peer = comm.sendrecv(mine, (rank + 1) % world)
        torch.cuda.synchronize()
        ok = got.shape == (world, 4, 1000) and got[:, 0, 0].tolist() == [r + 1 for r in range(world)] and int(peer[0, 0]) == (rank + 1) % world + 1
        comm.close()
        q.put((rank, bool(ok)))
    finally:
        dist.destroy_process_group()


def test_fisr_comm_two_ranks_over_rccl():
    """The C-ABI's own RCCL communicator (fisr_comm_*: unique id, init, all-gather, send/recv) between two GPUs.  Needs two
    devices: skipped on the one-GPU box, runs wherever a node is available."""
    if not torch.cuda.is_available() or torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs")
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_comm_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=300) for _ in range(2)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(ok for _, ok in res)


def _inputs(seed, H, W):
    rng = np.random.default_rng(seed)
    frames = [torch.from_numpy(rng.integers(0, 256, (H, W, 3), dtype=np.uint8)) for _ in range(3)]
    flows = [torch.from_numpy(rng.normal(0, 20, (H, W, 2)).astype(np.float32)) for _ in range(4)]
    warps = [torch.from_numpy((rng.random((H, W, 3)) * 255).astype(np.float32)) for _ in range(4)]
    return frames, flows, warps


def _worker(rank, world, port, num_patch, precision, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from fisr_amd import dist as fdist
        from fisr_amd import weights
        from fisr_amd.fisrnet import FISRnet
        torch.cuda.set_device(0)
        net = FISRnet(device="cuda:0", precision=precision)
        net.set_weights(weights.synthetic_weights(2020))
        topo = fdist.TileTopology(num_patch, world, rank)
        grp = topo.make_group()
        H, W = 64 * num_patch[0], 96 * num_patch[1]
        B = 2                                               # two windows through one pair of collectives
        wins = [_inputs(40 + 10 * topo.group_index + b, H, W) for b in range(B)]
        dev = lambda ts: [t.cuda() for t in ts]
        cores = torch.cat([fdist.pack_core(net, dev(fr), dev(fl), dev(wp), H, W, num_patch, topo.tile)
                           for fr, fl, wp in wins], dim=0)
        got = fdist.tile_parallel_engine_window(net, cores, num_patch, group=grp)      # [B, 2H, 2W, 9] uint8
        # single-process reference on the same engine: pack the whole frame, tile loop, quantise
        ok = True
        for b, (fr, fl, wp) in enumerate(wins):
            inp = net.pack_input(dev(fr), dev(fl), dev(wp), H, W)
            full = net.forward_tiled(inp, num_patch)
            exp, _ = net.unpack_output(full, want_rgb=False)
            ok = ok and bool(torch.equal(got[b], exp))
        torch.cuda.synchronize()
        q.put((rank, ok, tuple(got.shape)))
        net.close()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("num_patch,groups,precision", [((2, 2), 1, "fp32"), ((2, 2), 2, "bf16x3"), ((2, 4), 1, "fp32d")])
def test_tile_parallel_real_engine_bit_exact(num_patch, groups, precision):
    """world = tiles x groups gloo ranks on cuda:0: the sharded path equals forward_tiled bit for bit
    (tiles are independent given the halo, so sharding must not change a single byte)."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    world = num_patch[0] * num_patch[1] * groups
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, num_patch, precision, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=600) for _ in range(world)]
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    H, W = 64 * num_patch[0], 96 * num_patch[1]
    for r in sorted(res):
        assert r[1], r
        assert r[2] == (2, 2 * H, 2 * W, 9)


def _flow_worker(rank, world, port, folder, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import types
        from fisr_amd import harness
        torch.cuda.set_device(0)
        args = types.SimpleNamespace(frame_folder_path=folder, FISR_input_size=(96, 96), frame_num=5, flow_precision="fp32",
                                     synthetic_weights=3, pwc_ckpt=None)
        net = types.SimpleNamespace(device=torch.device("cuda:0"))
        name, pred = harness.compute_flow(net, args, rank, world, return_array=True)
        q.put((rank, name, pred.shape, float(np.abs(pred).sum()), pred if rank == 1 else None))
    finally:
        dist.destroy_process_group()


def test_flow_of_a_clip_is_sharded_over_the_ranks(tmp_path):
    """`FISR_for_video` under several ranks used to let rank 0 compute the whole clip's flow while the others sat in a
    barrier, then re-read the file by name.  Now the frame pairs are sharded (pair p on rank p % world), every rank gathers all
    flows in memory and rank 0 writes the reference's .flo: two gloo ranks on cuda:0 against the single-process result."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import types
    from fisr_amd import harness
    from fisr_amd import io as fio
    rng = np.random.default_rng(5)
    folder = tmp_path / "clip"
    folder.mkdir()
    base = rng.integers(0, 256, (12, 12, 3)).astype(np.float32)
    for k in range(5):
        fr = np.kron(np.roll(base, k, axis=1), np.ones((8, 8, 1), np.float32)) + rng.normal(0, 3, (96, 96, 3))
        fio.write_png(str(folder / f"fr_{k:02d}.png"), np.clip(fr, 0, 255).astype(np.uint8))
    args = types.SimpleNamespace(frame_folder_path=str(folder), FISR_input_size=(96, 96), frame_num=5, flow_precision="fp32",
                                 synthetic_weights=3, pwc_ckpt=None)
    name1, solo = harness.compute_flow(types.SimpleNamespace(device=torch.device("cuda:0")), args, return_array=True)
    assert solo.shape == (4, 2, 96, 96, 2)
    os.remove(name1)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_flow_worker, args=(r, 2, port, str(folder), q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=600) for _ in range(2)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert res[0][2] == res[1][2] == (4, 2, 96, 96, 2) and res[0][3] == res[1][3]          # both ranks hold the same, whole flow
    # (a pair run alone and inside a 5-frame run share the pyramid code path: same kernels, same inputs -> the same flow)
    assert np.abs(res[1][4] - solo).max() < 1e-4, float(np.abs(res[1][4] - solo).max())
    assert os.path.isfile(name1) and np.array_equal(fio.read_flo_file_5dim(name1), res[1][4])   # rank 0 wrote the reference's file


def _run_bench(world, extra, port):
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, FISR_BENCH_BACKEND="gloo", FISR_BENCH_ONE_DEVICE="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(root, "bench.py"), "--gpus", str(world), "--steps", "2", "--warmup", "1",
           "--precision", "fp16", "--batch", "tile", "--no-roofline"] + extra
    p = subprocess.run(cmd, cwd=root, env=env, capture_output=True, text=True, timeout=1500)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]                 # exactly ONE JSON line, printed by rank 0
    return json.loads(lines[0])


@pytest.mark.parametrize("parallelism", ["frame", "tile"])
def test_bench_eight_ranks_end_to_end_on_one_device(parallelism):
    """`bench.py --gpus 8` as the driver launches it (torch.distributed.run, one process per rank), both parallelism modes, with
    the ranks sharing cuda:0 over gloo (a 1-GPU box cannot host RCCL ranks): rendezvous, 8 engines, the asynchronous gather of
    the output frames (frame) / halo + tile all-gathers in 2 groups of 2x2 (tile), barriers, max-over-ranks timing, per-rank
    report, one JSON line.  The arithmetic is the single-GPU engine's (bit-exactness of the sharding: the tests above)."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    line = _run_bench(8, ["--parallelism", parallelism], _free_port())
    assert line["n_gpus"] == 8 and line["steps"] == 2 and line["value"] > 0 and line["scaling"] == "weak"
    assert ("frame-parallel x8" in line["config"]["parallelism"]) if parallelism == "frame" else ("2 stack(s) x 4 tiles" in line["config"]["parallelism"])
    assert len(line["per_rank"]) == 8 and sorted(r["rank"] for r in line["per_rank"]) == list(range(8))
    assert all(r["compute_ms_per_step"] > 0 for r in line["per_rank"])
    # value = units of all ranks / max-over-ranks time: 8 stacks (frame) or 2 stacks (tile) of 7 unique frames per step
    stacks = 8 if parallelism == "frame" else 2
    assert abs(line["value"] - stacks * 7 * 2 / (line["ms_per_step"] * 2e-3)) < 0.02 * line["value"]
    if parallelism == "frame":
        assert line["collective"]["bytes_per_rank_per_step"] == 3 * 2048 * 3840 * 9
        # the gather's share of a rank's step, from the per-rank report: wall time up to "my last gather has arrived" against the
        # compute stream's own time, and the host time inside the collective calls (asynchronous: a fraction of the wall time)
        for r in line["per_rank"]:
            wall, comp, host = r["wall_ms_per_step_incl_gather_drain"], r["compute_ms_per_step"], r["host_ms_in_gather_calls_per_step"]
            share = max(0.0, wall - comp) / wall
            assert 0.0 <= share < 1.0 and host is not None and 0.0 <= host <= wall, r
        print("gather share of the step per rank (gloo, 8 ranks on one device):",
              [round(max(0.0, r["wall_ms_per_step_incl_gather_drain"] - r["compute_ms_per_step"]) / r["wall_ms_per_step_incl_gather_drain"], 3) for r in line["per_rank"]])


def test_bench_dry_run_reports_the_memory_plan_of_every_rank():
    """`bench.py --gpus 8 --dry-run`: no step is run; every rank reports what it would hold on its device --
    packed weights (measured), the activation arena the LIBRARY asks for its forward batch, inputs, outputs, gather buffers --
    against the device, and the launcher's exit status says whether every plan fits (here: 8 default fp32 plans of ~35 GB on ONE
    288-GB device, the test hook's sharing rule: 1/8 of the device each)."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, FISR_BENCH_BACKEND="gloo", FISR_BENCH_ONE_DEVICE="1")
    for extra, prec in ((["--parallelism", "frame"], "fp32"), (["--parallelism", "tile"], "fp32")):
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=8", "--master-addr", "127.0.0.1",
               "--master-port", str(_free_port()), os.path.join(root, "bench.py"), "--gpus", "8", "--precision", prec, "--dry-run"] + extra
        p = subprocess.run(cmd, cwd=root, env=env, capture_output=True, text=True, timeout=900)
        lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
        assert len(lines) == 1, (p.stdout[-2000:], p.stderr[-2000:])
        plan = json.loads(lines[0])
        assert plan["dry_run"] and plan["n_gpus"] == 8 and len(plan["per_rank"]) == 8
        # (every rank leaves with status 3 when a plan does not fit; torch.distributed.run reports its children's failure as 1)
        assert (p.returncode == 0) == plan["fits"], (p.returncode, plan["fits"])
        for r in plan["per_rank"]:
            b = r["bytes"]
            assert b["activation_arena"] > 0 and b["weights_packed_measured"] > 100e6 and r["total_bytes"] > b["activation_arena"]
            assert r["device_total_bytes"] > 200e9 and r["ranks_sharing_device"] == 8
        # what a step puts on the links, per rank and collective, with the link-time estimate beside the expected compute time
        assert plan["link_model"]["xgmi_link_gbps_both_directions"] == 153.0 and plan["link_model"]["links_per_gpu"] == 7
        frame_bytes = 3 * 2048 * 3840 * 9
        for r in plan["per_rank"]:
            cs = r["collectives_per_step"]
            if extra[1] == "frame":
                assert len(cs) == 1 and cs[0]["group_size"] == 8
                assert cs[0]["send_bytes"] == (0 if r["rank"] == 0 else frame_bytes) and cs[0]["recv_bytes"] == (7 * frame_bytes if r["rank"] == 0 else 0)
                assert abs(cs[0]["link_ms_direct"] - frame_bytes / 76.5e9 * 1e3) < 0.01 and abs(cs[0]["link_ms_ring"] - 7 * cs[0]["link_ms_direct"]) < 0.05
            else:
                assert [c["group_size"] for c in cs] == [4, 4]
                assert cs[0]["send_bytes"] == 3 * (2 * 32 * 960 + 2 * 512 * 32) * 29 * 4 and cs[0]["recv_bytes"] == 3 * cs[0]["send_bytes"]
                assert cs[1]["send_bytes"] == 3 * 1024 * 1920 * 18 and cs[1]["recv_bytes"] == 3 * cs[1]["send_bytes"]
        exp = plan["compute_ms_per_step_expected"]
        assert exp is None or (exp["ms"] > 5 and exp["source"].startswith("profiles/"))
        if exp is not None:      # the collectives are a few per cent of a step: the plan says so before a node has been seen
            worst = max(c["link_ms_ring"] for r in plan["per_rank"] for c in r["collectives_per_step"])
            print(f"dry run {extra[1]}: slowest collective {worst:.2f} ms on a ring against {exp['ms']:.1f} ms of compute per step")
        r0 = plan["per_rank"][0]["bytes"]
        if extra[1] == "frame":        # 12 tiles of 544 x 992 per forward; rank 0 holds the gather's receive buffer
            assert r0["forward_batch"] == [12, 544, 992] and r0["gather_receive_buffer"] == 8 * 3 * 2048 * 3840 * 9
            assert all(q["bytes"]["gather_receive_buffer"] == 0 for q in plan["per_rank"][1:])
        else:                          # the rank's own tile of the 3 windows
            assert r0["forward_batch"] == [3, 544, 992] and r0["gather_send_buffers"] == 0
        print(f"dry run {extra[1]}: fits {plan['fits']}, per-rank total {plan['per_rank'][0]['total_bytes'] / 1e9:.1f} GB "
              f"(arena {r0['activation_arena'] / 1e9:.1f} GB, weights {r0['weights_packed_measured'] / 1e9:.2f} GB)")


def _rccl_single_worker(port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        from fisr_amd import dist as fdist
        ok = True
        # the asynchronous output gather exactly as bench.py drives it, on the real backend: side stream, events, views of
        # one receive buffer as the gather list, alternating send buffers, several steps in flight
        ag = fdist.AsyncGather((3, 64, 96, 9), torch.uint8, "cuda:0", dst=0)
        for step in range(4):
            buf = ag.buffer(step)
            buf.fill_(step + 1)
            ag.submit(step)
        r = ag.wait()
        torch.cuda.synchronize()
        ok = ok and tuple(r.shape) == (1, 3, 64, 96, 9) and int(r.max()) == 4 and int(r.min()) == 4
        # ... and it must not BLOCK: under gloo the "asynchronous" gather holds the host for the whole step (322.8 of 324.4 ms in
        # a 2-rank log of round 4); under RCCL submit() only enqueues.  (a) host time of submit() < 1 ms (best of 5, warm);
        # (b) the compute stream does not wait for a gather in flight -- the side stream is held up by a long sleep kernel, work
        # enqueued on the compute stream behind submit() finishes while the gather's done-event is still pending; (c) re-using the
        # send buffer DOES wait for it (buffer() makes the compute stream wait for that event).
        import time as _time
        big = fdist.AsyncGather((3, 512, 960, 9), torch.uint8, "cuda:0", dst=0)
        host_ms = []
        for step in range(5):
            big.buffer(step).fill_(7)
            t0 = _time.perf_counter()
            big.submit(step)
            host_ms.append((_time.perf_counter() - t0) * 1e3)
        big.wait()
        torch.cuda.synchronize()
        assert min(host_ms) < 1.0, f"AsyncGather.submit blocks the host under RCCL: {host_ms} ms"
        with torch.cuda.stream(big.side):
            torch.cuda._sleep(int(1.0e9))                  # ~0.4 s of the side stream (cycles of the shader clock)
        buf = big.buffer(6)                                # slot 0: its last gather finished above -> no wait on the compute stream
        buf.fill_(9)
        big.submit(6)
        probe = torch.zeros(1 << 20, device="cuda:0").add_(1)
        after = torch.cuda.Event()
        after.record()
        after.synchronize()                                # the compute stream's work behind submit() is DONE ...
        assert not big.done[0].query(), "the compute stream waited for a gather in flight (or the side stream was not held up)"
        again = big.buffer(8)                              # slot 0 again: now the compute stream must wait for that gather
        again.fill_(1)
        reuse = torch.cuda.Event()
        reuse.record()
        assert not reuse.query(), "re-using a send buffer did not wait for the gather that reads it"
        big.wait()
        reuse.synchronize()
        assert big.done[0].query() and int(probe[0]) == 1 and int(big.recv.max()) == 9
        # halo all-gather + tile gather (a 1x1 "tile group"), prefetched on the side stream
        core = torch.rand((2, 64, 96, 29), device="cuda:0")
        pf = fdist.HaloPrefetcher((1, 1), "cuda:0")
        tile_in = pf.finish(pf.start(core))
        ok = ok and torch.equal(tile_in, core)
        full = fdist.gather_tiles(torch.ones((2, 128, 192, 9), dtype=torch.uint8, device="cuda:0"), (1, 1))
        ok = ok and tuple(full.shape) == (2, 128, 192, 9)
        # the reductions / object gathers of bench.py's report
        t = torch.tensor([1.5], dtype=torch.float64, device="cuda:0")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        objs = [None]
        dist.all_gather_object(objs, {"rank": 0})
        dist.barrier()
        ok = ok and float(t.item()) == 1.5 and objs[0] == {"rank": 0}
        # the C-ABI's own communicator on the same device
        comm = fdist.FisrComm.from_torch_group(0)
        got = comm.allgather(torch.arange(1000, dtype=torch.int32, device="cuda:0"))
        torch.cuda.synchronize()
        ok = ok and tuple(got.shape) == (1, 1000) and int(got[0, 999]) == 999
        comm.close()
        q.put(bool(ok))
    finally:
        dist.destroy_process_group()


def test_rccl_backend_single_rank_collectives():
    """Every collective call this code makes -- the asynchronous gather on its side stream, the halo / tile all-gathers, the
    reductions and object gathers of bench.py, fisr_comm_* -- once through REAL RCCL (`backend="nccl"`): a world of one rank is
    all a one-GPU box can host, so the data path is a self-copy, but the API use, stream / event ordering and librccl loading are
    the ones an 8-GPU node runs."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_rccl_single_worker, args=(_free_port(), q))
    p.start()
    ok = q.get(timeout=600)
    p.join(timeout=120)
    assert p.exitcode == 0 and ok
