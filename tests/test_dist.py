"""World-size-2 (and 4) multi-process CPU tests of the N>1 sharding logic over gloo."""
import os
import socket

import numpy as np
import pytest

torch = pytest.importorskip("torch")
import torch.distributed as dist  # noqa: E402
import torch.multiprocessing as mp  # noqa: E402

from fisr_amd import dist as fdist  # noqa: E402
from fisr_amd import tiling  # noqa: E402


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _fake_forward(x):
    """Deterministic stand-in for the FISRnet forward with a small receptive field:
    [1,H,W,29] -> [1,2H,2W,9]; x2 nearest up-sample of a 3x3 box filter of 9 channels."""
    x = x[..., :9].permute(0, 3, 1, 2)
    y = torch.nn.functional.avg_pool2d(torch.nn.functional.pad(x, (1, 1, 1, 1)), 3, stride=1)
    y = torch.nn.functional.interpolate(y, scale_factor=2, mode="nearest")
    return y.permute(0, 2, 3, 1).contiguous()


def _reference_frame(inp, num_patch):
    """Single-process tile loop of FISRnet.py:847-880 with the same forward."""
    _, h, w, _ = inp.shape
    full = torch.zeros((h * 2, w * 2, 9))
    for t in tiling.plan_tiles(h, w, num_patch):
        pred = _fake_forward(inp[:, t.h_lo:t.h_hi, t.w_lo:t.w_hi, :])[0]
        full[t.dst_y:t.dst_y + t.out_h, t.dst_x:t.dst_x + t.out_w] = pred[t.src_y:t.src_y + t.out_h, t.src_x:t.src_x + t.out_w]
    return full


def _worker(rank, world, port, num_patch, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        # world may hold several tile groups (8 ranks = 2 windows x (2,2) tiles): every group works on its
        # own window, the collectives stay inside the group
        topo = fdist.TileTopology(num_patch, world, rank)
        grp = topo.make_group()
        g = torch.Generator().manual_seed(7 + topo.group_index)
        h, w = 64 * num_patch[0], 96 * num_patch[1]
        inp = torch.rand((1, h, w, 29), generator=g)
        y0, y1, x0, x1 = fdist.core_region(h, w, num_patch, topo.tile)
        core = inp[0, y0:y1, x0:x1].contiguous()
        # 1. halo exchange reproduces the reference's input slice
        tile_in = fdist.exchange_halos(core, num_patch, grp)
        t = tiling.plan_tiles(h, w, num_patch)[topo.tile]
        ok_halo = torch.equal(tile_in, inp[0, t.h_lo:t.h_hi, t.w_lo:t.w_hi])
        # 2. the whole tile-parallel window equals the single-process tile loop
        frame = fdist.tile_parallel_window(core, num_patch, _fake_forward, group=grp)
        ok_frame = torch.equal(frame, _reference_frame(inp, num_patch))
        # 2b. a batch of windows (a 5-frame stack = 3 windows) through ONE pair of collectives
        inp3 = torch.rand((3, h, w, 29), generator=g)
        frames3 = fdist.tile_parallel_window(inp3[:, y0:y1, x0:x1].contiguous(), num_patch, _fake_forward, group=grp)
        ok_frame = ok_frame and all(torch.equal(frames3[b], _reference_frame(inp3[b:b + 1], num_patch)) for b in range(3))
        # 3. frame-parallel sharding covers every unit exactly once
        units = fdist.shard_units(7, world, rank)
        gathered = [None] * world
        dist.all_gather_object(gathered, units)
        ok_units = sorted(sum(gathered, [])) == list(range(7))
        # 3b. frame-parallel output collection (cfg4: gather of the uint8 frames to rank 0)
        mine = torch.full((2, 3), rank, dtype=torch.uint8)
        got = fdist.gather_to(mine, dst=0)
        ok_units = ok_units and ((got is None) if rank else (got.shape == (world, 2, 3) and got[:, 0, 0].tolist() == list(range(world))))
        # 3c. the asynchronous form bench.py uses: alternating send buffers, a receive buffer allocated once, several steps
        ag = fdist.AsyncGather((2, 3), torch.uint8, "cpu", dst=0)
        recv_id = id(ag.recv)
        for step in range(5):
            ag.buffer(step).fill_(10 * step + rank)
            ag.submit(step)
            r = ag.wait()
            ok_units = ok_units and ((r is None) if rank else (id(r) == recv_id and r[:, 1, 2].tolist() == [10 * step + k for k in range(world)]))
        # 3d. the halo exchange started ahead (tile-parallel: under the previous window's forward) gives the same tile input
        pf = fdist.HaloPrefetcher(num_patch, "cpu", group=grp)
        ok_halo = ok_halo and torch.equal(pf.finish(pf.start(core)), tile_in)
        frame_pf = fdist.tile_parallel_window(core, num_patch, _fake_forward, group=grp, tile_in=pf.finish(pf.start(core)))
        ok_frame = ok_frame and torch.equal(frame_pf, frame)
        # 4. max-over-ranks timing reduction used by bench.py
        tt = torch.tensor([float(rank + 1)], dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        q.put((rank, ok_halo, ok_frame, ok_units, float(tt.item()) == world))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("num_patch,groups", [((1, 2), 1), ((2, 2), 1), ((2, 2), 2), ((2, 4), 1)])
def test_tile_parallel_gloo(num_patch, groups):
    """(2,2) x 2 groups and (2,4) x 1 group are the two 8-GPU plans of one node."""
    world = num_patch[0] * num_patch[1] * groups
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, num_patch, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for r in sorted(res):
        assert all(r[1:]), r


def test_shard_units_and_errors():
    assert fdist.shard_units(8, 8, 3) == [3]
    assert fdist.shard_units(3, 2, 1) == [1]
    assert fdist.shard_units(2, 4, 3) == []
    with pytest.raises(ValueError):
        fdist.shard_units(4, 2, 2)
    with pytest.raises(ValueError):
        fdist.tile_of_rank((2, 2), 2, 0)
    assert fdist.tile_of_rank((2, 2), 8, 6) == 2
    t = fdist.TileTopology((2, 2), 8, 6)
    assert (t.n_groups, t.group_index, t.tile, t.pH, t.pW, t.group_ranks) == (2, 1, 2, 1, 0, [4, 5, 6, 7])
    assert fdist.core_region(1024, 1920, (2, 2), 3) == (512, 1024, 960, 1920)
    assert fdist.core_region(1024, 1920, (2, 4), 5) == (512, 1024, 480, 960)
