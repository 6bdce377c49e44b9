"""Multi-GPU sharding of the FISRnet hot path: one process per GPU, torch.distributed
(backend "nccl" == RCCL over xGMI on ROCm; "gloo" in the CPU tests).

The reference is single-GPU (main.py:19-20).  Its work units are independent:

  * frame-parallel ("window"/"stack"): every 3-frame window (FISRnet.py:798-799) -- or whole
    5-frame stack -- is an independent forward; weights are replicated (193 MB).  Units are
    dealt round-robin; NO data-path collective.  This is what bench.py scales (weak scaling).
  * tile-parallel: the num_patch[0] x num_patch[1] tiles of one window (FISRnet.py:847-880) go
    to different ranks (one tile per rank, world == number of tiles) for latency.  Two real
    exchange steps, both all-gathers over xGMI:
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


def shard_units(n_units: int, world: int, rank: int) -> List[int]:
    """Round-robin assignment of independent units (windows / stacks) to ranks."""
    if not 0 <= rank < world:
        raise ValueError("rank out of range")
    return list(range(rank, n_units, world))


def tile_of_rank(num_patch: Tuple[int, int], world: int, rank: int) -> int:
    if world != num_patch[0] * num_patch[1]:
        raise ValueError(f"tile-parallel needs world ({world}) == number of tiles {num_patch}")
    return rank


# ----------------------------------------------------------------------------- halo exchange
def border_ring(core, pb: int = PB):
    """core [sH,sW,C] -> one flat buffer [top pb rows | bottom pb rows | left pb cols | right pb cols]."""
    import torch
    return torch.cat([core[:pb].reshape(-1), core[-pb:].reshape(-1),
                      core[:, :pb].reshape(-1), core[:, -pb:].reshape(-1)])


def _ring_views(ring, sH: int, sW: int, c: int, pb: int):
    o = 0
    top = ring[o:o + pb * sW * c].view(pb, sW, c); o += pb * sW * c
    bot = ring[o:o + pb * sW * c].view(pb, sW, c); o += pb * sW * c
    left = ring[o:o + sH * pb * c].view(sH, pb, c); o += sH * pb * c
    right = ring[o:o + sH * pb * c].view(sH, pb, c)
    return top, bot, left, right


def assemble_tile_input(core, rings: Sequence, num_patch: Tuple[int, int], rank: int, pb: int = PB):
    """Build this rank's halo'd tile input (what FISRnet.py:865 slices out of the full frame) from
    its own core [sH,sW,C] and the all-gathered border rings of all ranks."""
    import torch
    nh, nw = num_patch
    pH, pW = rank // nw, rank % nw
    sH, sW, c = core.shape
    up, down, lft, rgt = pH > 0, pH < nh - 1, pW > 0, pW < nw - 1
    H = sH + pb * (up + down)
    W = sW + pb * (lft + rgt)
    out = torch.zeros((H, W, c), dtype=core.dtype, device=core.device)
    y0, x0 = pb * up, pb * lft
    out[y0:y0 + sH, x0:x0 + sW] = core

    def ring(r):
        return _ring_views(rings[r], sH, sW, c, pb)

    if up:
        out[:pb, x0:x0 + sW] = ring(rank - nw)[1]                     # neighbour's bottom rows
    if down:
        out[y0 + sH:, x0:x0 + sW] = ring(rank + nw)[0]                # neighbour's top rows
    if lft:
        out[y0:y0 + sH, :pb] = ring(rank - 1)[3]                      # neighbour's right cols
    if rgt:
        out[y0:y0 + sH, x0 + sW:] = ring(rank + 1)[2]                 # neighbour's left cols
    # corners come from the diagonal neighbour's top/bottom rows
    if up and lft:
        out[:pb, :pb] = ring(rank - nw - 1)[1][:, -pb:]
    if up and rgt:
        out[:pb, x0 + sW:] = ring(rank - nw + 1)[1][:, :pb]
    if down and lft:
        out[y0 + sH:, :pb] = ring(rank + nw - 1)[0][:, -pb:]
    if down and rgt:
        out[y0 + sH:, x0 + sW:] = ring(rank + nw + 1)[0][:, :pb]
    return out


def exchange_halos(core, num_patch: Tuple[int, int], group=None, pb: int = PB):
    """All-gather the 32-px border rings and return this rank's halo'd tile input."""
    import torch
    import torch.distributed as dist
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    tile_of_rank(num_patch, world, rank)
    mine = border_ring(core, pb).contiguous()
    rings = [torch.empty_like(mine) for _ in range(world)]
    dist.all_gather(rings, mine, group=group)
    return assemble_tile_input(core, rings, num_patch, rank, pb)


# ----------------------------------------------------------------------------- output gather
def gather_tiles(my_tile, num_patch: Tuple[int, int], group=None):
    """my_tile [sH*sf, sW*sf, C] (already trimmed) -> full frame [h*sf, w*sf, C] on every rank."""
    import torch
    import torch.distributed as dist
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    tile_of_rank(num_patch, world, rank)
    my_tile = my_tile.contiguous()
    parts = [torch.empty_like(my_tile) for _ in range(world)]
    dist.all_gather(parts, my_tile, group=group)
    nh, nw = num_patch
    rows = [torch.cat(parts[r * nw:(r + 1) * nw], dim=1) for r in range(nh)]
    return torch.cat(rows, dim=0)


def tile_parallel_window(core_input, num_patch: Tuple[int, int], forward: Callable, sf: int = 2, group=None,
                         postprocess: Callable | None = None, pb: int = PB):
    """One window, one tile per rank.  core_input [sH,sW,29] is this rank's core region of the
    packed input; `forward(tile_in [1,H,W,29]) -> [1,H*sf,W*sf,9]` is the FISRnet forward
    (net.model(...)[2] on the GPU path).  Returns the full frame on every rank."""
    import torch.distributed as dist
    rank = dist.get_rank(group)
    nh, nw = num_patch
    pH, pW = rank // nw, rank % nw
    tile_in = exchange_halos(core_input, num_patch, group, pb)
    pred = forward(tile_in.unsqueeze(0))[0]
    # trim_patch_boundary (utils.py:138-159): drop pb*sf HR pixels on the sides that carry a halo
    y0 = pb * sf if pH > 0 else 0
    x0 = pb * sf if pW > 0 else 0
    sH, sW = core_input.shape[0], core_input.shape[1]
    pred = pred[y0:y0 + sH * sf, x0:x0 + sW * sf]
    if postprocess is not None:
        pred = postprocess(pred)
    return gather_tiles(pred, num_patch, group)


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
