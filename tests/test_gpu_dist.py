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
