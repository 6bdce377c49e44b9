"""Multi-GPU sharding of the FISRnet hot path: one process per GPU, torch.distributed
(backend "nccl" == RCCL over xGMI on ROCm; "gloo" in the CPU tests).

The reference is single-GPU (main.py:19-20).  Its work units are independent:

  * frame-parallel ("window"/"stack"): every 3-frame window (FISRnet.py:798-799) -- or whole
    5-frame stack -- is an independent forward; weights are replicated (193 MB).  Units are
    dealt round-robin; NO data-path collective.  This is what bench.py scales (weak scaling).
  * tile-parallel: the num_patch[0] x num_patch[1] tiles of one window (FISRnet.py:847-880) go
    to different ranks (one tile per rank; the ranks form world / tiles independent tile groups,
    e.g. 8 GPUs = 2 windows x (2x2) tiles with the reference's default seams) for latency.  Two real
    exchange steps, both all-gathers over xGMI inside a tile group:
      1. input halos -- each rank assembles only its own core region of the 29-ch input; the
         32-px border ring of every core is all-gathered and each rank cuts the strips its tile
         needs out of its neighbours' rings (7 MB/rank at 1080p 2x2);
      2. output tiles -- each rank's trimmed, quantised (uint8) prediction is all-gathered so
         every rank (or just rank 0) holds the full frame (8.8-35 MB/rank).
    xGMI is a fully connected mesh (7 links x ~153 GB/s per GPU): an all-gather of this size
    is one hop and latency-dominated (~0.1-0.3 ms) next to a multi-ms forward.

Everything here is index plumbing on torch tensors; it works on any device/backend, so the
N>1 logic is covered by world_size-2 gloo tests on CPU (tests/test_dist.py) with an injected
forward function.
"""
from __future__ import annotations

from typing import Callable, List, Sequence, Tuple

from . import tiling

PB = tiling.PATCH_BOUNDARY


# xGMI on an MI355X node: a fully connected mesh, 7 links per GPU, ~153 GB/s per link counting both directions (the figure this
# code base plans with; no link counter has been read on hardware yet) -> ~76 GB/s into one GPU from one peer.
XGMI_LINK_GBPS_BIDIR = 153.0
XGMI_LINKS_PER_GPU = 7


def collective_plan(parallelism: str, num_patch: Tuple[int, int], world: int, rank: int, h: int, w: int, windows: int = 3,
                    sf: int = 2, gather: bool = True, want_rgb: bool = True, pb: int = PB):
    """What ONE step (one 5-frame stack: `windows` windows of the cropped h x w LR frame) puts on the links, for `rank`: a list of
    {"collective", "group_size", "send_bytes", "recv_bytes", "link_ms_direct", "link_ms_ring"} -- bytes this rank sends / receives
    per step and the time the slowest link is busy with them, for the two schedules RCCL may pick: every peer over its own
    link at once ("direct": bytes of the largest single peer-to-peer piece / one direction of a link) and a ring
    ((group_size - 1) hops of one piece each).  Pure arithmetic: `bench.py --dry-run` prints it next to the compute time, so the
    first run on a node has a number to be compared with (FISRnet.py:798-799, 847-880 are the units being moved)."""
    one_way = XGMI_LINK_GBPS_BIDIR / 2 * 1e9
    out = []
    if world <= 1:
        return out
    if parallelism == "frame":
        if gather:
            piece = windows * (h * sf) * (w * sf) * 9                   # the uint8 YUV frames of the rank's stack
            out.append({"collective": "gather of the uint8 output frames to rank 0 (AsyncGather, side stream)", "group_size": world,
                        "send_bytes": 0 if rank == 0 else piece, "recv_bytes": (world - 1) * piece if rank == 0 else 0,
                        # every sender has its own link into rank 0: the links work side by side
                        "link_ms_direct": round(piece / one_way * 1e3, 3), "link_ms_ring": round((world - 1) * piece / one_way * 1e3, 3),
                        "note": "rank 0 takes %d x %.1f MB per step over %d of its %d links" % (world - 1, piece / 1e6, min(world - 1, XGMI_LINKS_PER_GPU), XGMI_LINKS_PER_GPU)})
        return out
    if parallelism != "tile":
        raise ValueError("parallelism must be 'frame' or 'tile'")
    T = num_patch[0] * num_patch[1]
    sH, sW = h // num_patch[0], w // num_patch[1]
    ring = windows * (2 * pb * sW + 2 * sH * pb) * 29 * 4               # border_ring() of the rank's cores, float32
    tile = windows * (sH * sf) * (sW * sf) * (18 if want_rgb else 9)    # trimmed uint8 tile(s): YUV (+ RGB)
    for name, piece in (("all-gather of the 32-px input halo rings inside the tile group (HaloPrefetcher, under the previous forward)", ring),
                        ("all-gather of the trimmed uint8 output tiles inside the tile group", tile)):
        out.append({"collective": name, "group_size": T, "send_bytes": piece, "recv_bytes": (T - 1) * piece,
                    "link_ms_direct": round(piece / one_way * 1e3, 3), "link_ms_ring": round((T - 1) * piece / one_way * 1e3, 3)})
    return out


def shard_units(n_units: int, world: int, rank: int) -> List[int]:
    """Round-robin assignment of independent units (windows / stacks) to ranks."""
    if not 0 <= rank < world:
        raise ValueError("rank out of range")
    return list(range(rank, n_units, world))


def tile_of_rank(num_patch: Tuple[int, int], world: int, rank: int) -> int:
    """Tile index of `rank` inside its tile group (see TileTopology); world must be a multiple of the tiles."""
    return TileTopology(num_patch, world, rank).tile


class TileTopology:
    """How `world` ranks share the tiles of the reference's test_patch plan.

    T = num_patch[0] * num_patch[1] tiles per window (FISRnet.py:847).  world must be a multiple of T: the
    ranks form world / T *tile groups* of T consecutive ranks; a group works on one window (or stack of
    windows) at a time, rank `g*T + t` owning tile t, so 8 GPUs run 2 windows x (2x2) tiles with the
    reference's default seams and numerics, or 1 window x (2x4) tiles (`--test_patch (2,4)`).  Groups are
    independent of each other (frame-parallel across groups); the two collectives of the path (halo rings
    in, uint8 tiles out) stay inside a group."""

    def __init__(self, num_patch: Tuple[int, int], world: int, rank: int):
        nh, nw = int(num_patch[0]), int(num_patch[1])
        T = nh * nw
        if T < 1 or world < 1 or world % T:
            raise ValueError(f"tile-parallel needs world ({world}) to be a multiple of the number of tiles {tuple(num_patch)}")
        if not 0 <= rank < world:
            raise ValueError("rank out of range")
        self.num_patch, self.world, self.rank, self.tiles = (nh, nw), world, rank, T
        self.n_groups = world // T
        self.group_index, self.tile = divmod(rank, T)
        self.group_ranks = list(range(self.group_index * T, (self.group_index + 1) * T))
        self.pH, self.pW = divmod(self.tile, nw)

    def all_group_ranks(self) -> List[List[int]]:
        return [list(range(g * self.tiles, (g + 1) * self.tiles)) for g in range(self.n_groups)]

    def make_group(self):
        """torch.distributed subgroup of this rank's tile group (every rank must call this: new_group is
        collective over the default group, all groups are created in the same order on all ranks)."""
        import torch.distributed as dist
        if self.n_groups == 1:
            return None                      # the default group IS the tile group
        mine = None
        for ranks in self.all_group_ranks():
            g = dist.new_group(ranks)
            if self.rank in ranks:
                mine = g
        return mine


def _group_rank_size(group):
    import torch.distributed as dist
    return dist.get_rank(group), dist.get_world_size(group)


def _all_gather(t, group=None):
    """all_gather of one contiguous tensor per rank -> list of tensors.  RCCL ("nccl") moves device
    tensors directly over xGMI; gloo (CPU tests, and the one-box GPU tests where several ranks share a
    device) only carries host tensors for this collective, so device tensors are staged through the host."""
    import torch
    import torch.distributed as dist
    _, world = _group_rank_size(group)
    backend = dist.get_backend(group)
    if t.is_cuda and backend != "nccl":
        host = t.cpu()
        parts = [torch.empty_like(host) for _ in range(world)]
        dist.all_gather(parts, host, group=group)
        return [p.to(t.device) for p in parts]
    parts = [torch.empty_like(t) for _ in range(world)]
    dist.all_gather(parts, t, group=group)
    return parts


def gather_to(t, dst: int = 0, group=None):
    """Gather one contiguous tensor per rank to rank `dst` (frame-parallel output collection, cfg4 of
    BASELINE.json: "RCCL gather of outputs"): returns [world, *t.shape] on dst, None elsewhere.  Same
    host staging rule as _all_gather for non-RCCL backends."""
    import torch
    import torch.distributed as dist
    rank, world = _group_rank_size(group)
    gdst = dst if group is None else dist.get_global_rank(group, dst)
    staged = t.is_cuda and dist.get_backend(group) != "nccl"
    src = (t.cpu() if staged else t).contiguous()
    parts = [torch.empty_like(src) for _ in range(world)] if rank == dst else None
    dist.gather(src, parts, dst=gdst, group=group)
    if rank != dst:
        return None
    out = torch.stack(parts)
    return out.to(t.device) if staged else out


class AsyncGather:
    """Frame-parallel output collection (cfg4 of BASELINE.json: "RCCL gather of outputs") that does not stall the compute stream.

    Every rank hands its uint8 output frames to `submit()`; the gather to rank `dst` runs on a SIDE stream into a receive buffer
    [world, *shape] allocated once (no per-step allocation, no torch.stack copy), while the compute stream goes on with the next
    stack.  The caller alternates between `n_buffers` send buffers (`buffer(i)`): buffer i may be overwritten again only after the
    gather that read it has finished, which `submit()` of the same slot / `wait()` enforce with events -- nothing blocks the host.
    Under RCCL ("nccl") the collective is device-to-device over xGMI; other backends (gloo in the one-box tests) stage through the
    host, synchronously (they exist to test the control flow, not to be fast)."""

    def __init__(self, shape, dtype, device, dst: int = 0, group=None, n_buffers: int = 2):
        import torch
        import torch.distributed as dist
        self.torch, self.dist = torch, dist
        self.group, self.dst = group, dst
        self.rank, self.world = _group_rank_size(group)
        self.gdst = dst if group is None else dist.get_global_rank(group, dst)
        self.device = torch.device(device)
        self.staged = self.device.type == "cuda" and dist.get_backend(group) != "nccl"
        self.send = [torch.zeros(tuple(shape), dtype=dtype, device=self.device) for _ in range(n_buffers)]
        rdev = "cpu" if self.staged else self.device
        self.recv = torch.zeros((self.world,) + tuple(shape), dtype=dtype, device=rdev) if self.rank == dst else None
        self.side = torch.cuda.Stream(device=self.device) if self.device.type == "cuda" else None
        self.done = [None] * n_buffers             # event per send buffer: its last gather has read it
        self.submitted = 0
        self.gather_ms = 0.0                       # host time spent inside the collective calls (diagnostic)

    def buffer(self, i: int):
        """Send buffer i, safe to overwrite on the compute stream: waits (on the stream, not the host) for the gather that last read it."""
        k = i % len(self.send)
        if self.done[k] is not None and self.side is not None:
            self.torch.cuda.current_stream(self.device).wait_event(self.done[k])
        return self.send[k]

    def submit(self, i: int):
        """Gather send buffer i (already written on the current stream) to rank dst.  Returns immediately under RCCL."""
        import time
        torch, dist = self.torch, self.dist
        k = i % len(self.send)
        t0 = time.perf_counter()
        parts = [self.recv[r] for r in range(self.world)] if self.rank == self.dst else None
        if self.side is None:                                   # CPU tensors (gloo tests)
            dist.gather(self.send[k], parts, dst=self.gdst, group=self.group)
        elif self.staged:                                       # device tensors over a host-only backend
            torch.cuda.current_stream(self.device).synchronize()
            dist.gather(self.send[k].cpu(), parts, dst=self.gdst, group=self.group)
        else:
            ready = torch.cuda.Event()
            ready.record(torch.cuda.current_stream(self.device))
            with torch.cuda.stream(self.side):
                self.side.wait_event(ready)
                dist.gather(self.send[k], parts, dst=self.gdst, group=self.group)     # enqueued behind `ready` on RCCL's stream
                ev = torch.cuda.Event()
                ev.record(self.side)
            self.done[k] = ev
        self.submitted += 1
        self.gather_ms += (time.perf_counter() - t0) * 1e3

    def wait(self):
        """Everything submitted so far has arrived (call before reading `recv` or before a timed region ends)."""
        if self.side is not None and not self.staged:
            self.side.synchronize()
        return self.recv


class HaloPrefetcher:
    """Tile-parallel: the halo all-gather of window k+1 under the forward of window k.  `start(core)` launches border-ring
    extraction + all-gather + tile assembly on a side stream and returns a handle; `finish(handle)` makes the compute stream wait
    for it and returns the halo'd tile input.  (Host-only backends run it synchronously.)"""

    def __init__(self, num_patch: Tuple[int, int], device, group=None, pb: int = PB):
        import torch
        import torch.distributed as dist
        self.torch = torch
        self.num_patch, self.group, self.pb = num_patch, group, pb
        self.device = torch.device(device)
        self.sync = self.device.type != "cuda" or dist.get_backend(group) != "nccl"
        self.side = None if self.sync else torch.cuda.Stream(device=self.device)

    def start(self, core):
        torch = self.torch
        if self.sync:
            return (exchange_halos(core, self.num_patch, self.group, self.pb), None)
        ready = torch.cuda.Event()
        ready.record(torch.cuda.current_stream(self.device))
        with torch.cuda.stream(self.side):
            self.side.wait_event(ready)
            tile_in = exchange_halos(core, self.num_patch, self.group, self.pb)
            core.record_stream(self.side)
            ev = torch.cuda.Event()
            ev.record(self.side)
        return (tile_in, ev)

    def finish(self, handle):
        tile_in, ev = handle
        if ev is not None:
            cur = self.torch.cuda.current_stream(self.device)
            cur.wait_event(ev)
            tile_in.record_stream(cur)
        return tile_in


# ----------------------------------------------------------------------------- halo exchange
def border_ring(core, pb: int = PB):
    """core [..., sH, sW, C] -> flat buffer [top pb rows | bottom pb rows | left pb cols | right pb cols]
    (leading batch dims kept: one ring per window when a stack of windows is exchanged at once)."""
    import torch
    lead = core.shape[:-3]
    f = lambda x: x.reshape(lead + (-1,))
    return torch.cat([f(core[..., :pb, :, :]), f(core[..., -pb:, :, :]),
                      f(core[..., :, :pb, :]), f(core[..., :, -pb:, :])], dim=-1)


def _ring_views(ring, sH: int, sW: int, c: int, pb: int):
    lead = ring.shape[:-1]
    o = 0
    top = ring[..., o:o + pb * sW * c].reshape(lead + (pb, sW, c)); o += pb * sW * c
    bot = ring[..., o:o + pb * sW * c].reshape(lead + (pb, sW, c)); o += pb * sW * c
    left = ring[..., o:o + sH * pb * c].reshape(lead + (sH, pb, c)); o += sH * pb * c
    right = ring[..., o:o + sH * pb * c].reshape(lead + (sH, pb, c))
    return top, bot, left, right


def assemble_tile_input(core, rings: Sequence, num_patch: Tuple[int, int], rank: int, pb: int = PB):
    """Build tile `rank`'s halo'd input (what FISRnet.py:865 slices out of the full frame) from its own
    core [..., sH, sW, C] and the all-gathered border rings of all tiles of the group."""
    import torch
    nh, nw = num_patch
    pH, pW = rank // nw, rank % nw
    sH, sW, c = core.shape[-3:]
    lead = core.shape[:-3]
    up, down, lft, rgt = pH > 0, pH < nh - 1, pW > 0, pW < nw - 1
    H = sH + pb * (up + down)
    W = sW + pb * (lft + rgt)
    out = torch.zeros(lead + (H, W, c), dtype=core.dtype, device=core.device)
    y0, x0 = pb * up, pb * lft
    out[..., y0:y0 + sH, x0:x0 + sW, :] = core

    def ring(r):
        return _ring_views(rings[r], sH, sW, c, pb)

    if up:
        out[..., :pb, x0:x0 + sW, :] = ring(rank - nw)[1]                     # neighbour's bottom rows
    if down:
        out[..., y0 + sH:, x0:x0 + sW, :] = ring(rank + nw)[0]                # neighbour's top rows
    if lft:
        out[..., y0:y0 + sH, :pb, :] = ring(rank - 1)[3]                      # neighbour's right cols
    if rgt:
        out[..., y0:y0 + sH, x0 + sW:, :] = ring(rank + 1)[2]                 # neighbour's left cols
    # corners come from the diagonal neighbour's top/bottom rows
    if up and lft:
        out[..., :pb, :pb, :] = ring(rank - nw - 1)[1][..., :, -pb:, :]
    if up and rgt:
        out[..., :pb, x0 + sW:, :] = ring(rank - nw + 1)[1][..., :, :pb, :]
    if down and lft:
        out[..., y0 + sH:, :pb, :] = ring(rank + nw - 1)[0][..., :, -pb:, :]
    if down and rgt:
        out[..., y0 + sH:, x0 + sW:, :] = ring(rank + nw + 1)[0][..., :, :pb, :]
    return out


def exchange_halos(core, num_patch: Tuple[int, int], group=None, pb: int = PB):
    """All-gather the 32-px border rings inside the tile group and return this rank's halo'd tile input.
    core: [sH,sW,C] or [B,sH,sW,C] (B windows exchanged in one collective)."""
    rank, world = _group_rank_size(group)
    if world != num_patch[0] * num_patch[1]:
        raise ValueError(f"tile group of {world} ranks cannot hold the tiles of {tuple(num_patch)}")
    rings = _all_gather(border_ring(core, pb).contiguous(), group)
    return assemble_tile_input(core, rings, num_patch, rank, pb)


# ----------------------------------------------------------------------------- output gather
def gather_tiles(my_tile, num_patch: Tuple[int, int], group=None):
    """my_tile [..., sH*sf, sW*sf, C] (already trimmed) -> full frame [..., h*sf, w*sf, C] on every rank
    of the tile group."""
    import torch
    rank, world = _group_rank_size(group)
    if world != num_patch[0] * num_patch[1]:
        raise ValueError(f"tile group of {world} ranks cannot hold the tiles of {tuple(num_patch)}")
    parts = _all_gather(my_tile.contiguous(), group)
    nh, nw = num_patch
    rows = [torch.cat(parts[r * nw:(r + 1) * nw], dim=-2) for r in range(nh)]
    return torch.cat(rows, dim=-3)


def tile_parallel_window(core_input, num_patch: Tuple[int, int], forward: Callable, sf: int = 2, group=None,
                         postprocess: Callable | None = None, pb: int = PB, tile_in=None):
    """One window (or a batch of B windows), one tile per rank of the tile group.  core_input [sH,sW,29] /
    [B,sH,sW,29] is this rank's core region of the packed input; `forward(tile_in [B,H,W,29]) ->
    [B,H*sf,W*sf,9]` is the FISRnet forward (net.model(...)[2] on the GPU path).  Returns the full frame(s)
    on every rank of the group.  Reference unit of work: FISRnet.py:847-880."""
    rank, _ = _group_rank_size(group)
    nh, nw = num_patch
    pH, pW = rank // nw, rank % nw
    batched = core_input.dim() == 4
    if tile_in is None:                  # (else: already exchanged, e.g. by a HaloPrefetcher under the previous window's forward)
        tile_in = exchange_halos(core_input, num_patch, group, pb)
    pred = forward(tile_in if batched else tile_in.unsqueeze(0))
    if not batched:
        pred = pred[0]
    # trim_patch_boundary (utils.py:138-159): drop pb*sf HR pixels on the sides that carry a halo
    y0 = pb * sf if pH > 0 else 0
    x0 = pb * sf if pW > 0 else 0
    sH, sW = core_input.shape[-3], core_input.shape[-2]
    pred = pred[..., y0:y0 + sH * sf, x0:x0 + sW * sf, :]
    if postprocess is not None:
        pred = postprocess(pred)
    return gather_tiles(pred, num_patch, group)


def core_region(h: int, w: int, num_patch: Tuple[int, int], tile: int):
    """(y0, y1, x0, x1) of tile `tile`'s core (no halo) in the cropped h x w frame."""
    nh, nw = num_patch
    sH, sW = h // nh, w // nw
    pH, pW = divmod(tile, nw)
    return pH * sH, (pH + 1) * sH, pW * sW, (pW + 1) * sW


def pack_core(net, frames_u8, flows, warps, h: int, w: int, num_patch: Tuple[int, int], tile: int):
    """Input assembly (FISRnet.py:828-843) of ONLY this rank's core region: the 11 source tensors are cut
    to the core rectangle and packed by the HIP kernel -> [1, sH, sW, 29] float32 on the GPU."""
    y0, y1, x0, x1 = core_region(h, w, num_patch, tile)
    cut = lambda t: t[y0:y1, x0:x1].contiguous()
    return net.pack_input([cut(f) for f in frames_u8], [cut(f) for f in flows], [cut(f) for f in warps],
                          y1 - y0, x1 - x0)


def tile_parallel_engine_window(net, cores, num_patch: Tuple[int, int], group=None, want_rgb: bool = False, tile_in=None):
    """The tile-parallel path with the real engine: cores [B,sH,sW,29] (this rank's core of B windows) ->
    halo all-gather -> fisr_forward on the halo'd tile (batch B) -> trim -> on-GPU clip/quantise/colour
    (fisr_unpack_output, per pixel, so per tile is exact) -> all-gather of the uint8 tiles.  Returns the
    full uint8 YUV frames [B, 2h, 2w, 9] on every rank of the tile group (71 MB per 4K window instead of
    283 MB in fp32), or (yuv, rgb [B, 3, 2h, 2w, 3]) with want_rgb."""
    import torch

    def fwd(tile_in):
        return net.model(tile_in, want_all=False)[2]

    def quantise(pred):                                            # [B, sH*2, sW*2, 9] float32 -> uint8
        outs = []
        for p in pred:
            yuv, rgb = net.unpack_output(p.contiguous(), want_rgb=want_rgb)
            if want_rgb:                                           # [3,h,w,3] -> 9 more channels, one gather
                yuv = torch.cat([yuv, rgb.permute(1, 2, 0, 3).reshape(yuv.shape)], dim=-1)
            outs.append(yuv)
        return torch.stack(outs)

    out = tile_parallel_window(cores, num_patch, fwd, sf=net.scale_factor, group=group, postprocess=quantise, tile_in=tile_in)
    if not want_rgb:
        return out
    B, H2, W2, _ = out.shape
    return out[..., :9].contiguous(), out[..., 9:].reshape(B, H2, W2, 3, 3).permute(0, 3, 1, 2, 4).contiguous()


class FisrComm:
    """The C-ABI's own RCCL communicator (include/fisr.h `fisr_comm_*`): the same two collectives for hosts
    that do not run under torch.distributed.  `unique_id()` on rank 0, ship the 128 bytes to the other ranks
    by any side channel (here: a torch.distributed broadcast if a group exists, else single rank), `init`."""

    def __init__(self, comm_id: bytes, nranks: int, rank: int, device_index: int):
        import ctypes
        from . import lib as _lib
        self._lib, self._L = _lib, _lib.lib()
        self._h = ctypes.c_void_p()
        buf = ctypes.create_string_buffer(bytes(comm_id), _lib.COMM_ID_BYTES)
        _lib.check(self._L.fisr_comm_init(ctypes.byref(self._h), buf, nranks, rank, device_index))
        self.nranks, self.rank = nranks, rank

    @staticmethod
    def unique_id() -> bytes:
        import ctypes
        from . import lib as _lib
        buf = ctypes.create_string_buffer(_lib.COMM_ID_BYTES)
        _lib.check(_lib.lib().fisr_comm_unique_id(buf))
        return buf.raw

    @classmethod
    def from_torch_group(cls, device_index: int, group=None):
        """Bootstrap over an existing torch.distributed group of any backend (the id travels as a CPU tensor)."""
        import torch
        import torch.distributed as dist
        if not dist.is_initialized():
            return cls(cls.unique_id(), 1, 0, device_index)
        rank, world = dist.get_rank(group), dist.get_world_size(group)
        obj = [cls.unique_id() if rank == 0 else None]
        dist.broadcast_object_list(obj, src=0, group=group)
        return cls(obj[0], world, rank, device_index)

    def allgather(self, send, recv=None):
        """send: contiguous device tensor; returns [nranks, *send.shape] (same dtype), async on the current stream."""
        import ctypes
        import torch
        send = send.contiguous()
        if recv is None:
            recv = torch.empty((self.nranks,) + tuple(send.shape), dtype=send.dtype, device=send.device)
        st = torch.cuda.current_stream(send.device).cuda_stream
        self._lib.check(self._L.fisr_comm_allgather(self._h, ctypes.c_void_p(send.data_ptr()), ctypes.c_void_p(recv.data_ptr()),
                                                    send.numel() * send.element_size(), ctypes.c_void_p(st)))
        return recv

    def sendrecv(self, send, peer: int):
        import ctypes
        import torch
        send = send.contiguous()
        recv = torch.empty_like(send)
        st = torch.cuda.current_stream(send.device).cuda_stream
        self._lib.check(self._L.fisr_comm_sendrecv(self._h, ctypes.c_void_p(send.data_ptr()), ctypes.c_void_p(recv.data_ptr()),
                                                   send.numel() * send.element_size(), peer, ctypes.c_void_p(st)))
        return recv

    def close(self):
        if self._h:
            self._L.fisr_comm_destroy(self._h)
            self._h = None
