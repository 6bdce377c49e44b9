"""Tile geometry of the reference's test harness (host-side integer maths).

Mirrors `get_HW_boundary` (utils.py:118-135), `trim_patch_boundary` (utils.py:138-159) and
the crop / tile loop of `FISRnet.test` (FISRnet.py:820-825, 847-880): the frame is cut in
num_patch[0] x num_patch[1] patches, each extended by a 32-px halo on interior sides; the
x2 prediction of a patch is trimmed by 64 px on those sides and written to the full frame.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import List, Tuple

PATCH_BOUNDARY = 32  # FISRnet.py:779


@dataclass(frozen=True)
class Tile:
    index: int
    # input window (LR pixels), FISRnet.py:865
    h_lo: int
    h_hi: int
    w_lo: int
    w_hi: int
    # interior of the x-sf prediction that survives the trim (tile-local HR pixels)
    src_y: int
    src_x: int
    # destination rectangle in the full HR frame, FISRnet.py:879-880
    dst_y: int
    dst_x: int
    out_h: int
    out_w: int

    @property
    def in_h(self) -> int:
        return self.h_hi - self.h_lo

    @property
    def in_w(self) -> int:
        return self.w_hi - self.w_lo


def crop_hw(H: int, W: int, num_patch: Tuple[int, int]) -> Tuple[int, int]:
    """FISRnet.py:820-824: crop so that every patch is a multiple of 32."""
    return H - H % (32 * num_patch[0]), W - W % (32 * num_patch[1])


def pad_hw(H: int, W: int, num_patch: Tuple[int, int]) -> Tuple[int, int]:
    """`--pad_mode` (not in the reference): the size the frame is replicate-padded to, so that the rows / columns
    FISRnet.py:820-824 crops away (56 of 1080 rows with 2 x 2 patches) are predicted too."""
    mh, mw = 32 * num_patch[0], 32 * num_patch[1]
    return (H + mh - 1) // mh * mh, (W + mw - 1) // mw * mw


def get_hw_boundary(pb: int, h: int, w: int, pH: int, sH: int, pW: int, sW: int):
    """utils.py:118-135."""
    h_lo = max(pH * sH - pb, 0)
    h_hi = min((pH + 1) * sH + pb, h)
    w_lo = max(pW * sW - pb, 0)
    w_hi = min((pW + 1) * sW + pb, w)
    add_h = (pb if pH * sH >= pb else 0) + (pb if (pH + 1) * sH + pb <= h else 0)
    add_w = (pb if pW * sW >= pb else 0) + (pb if (pW + 1) * sW + pb <= w else 0)
    return h_lo, h_hi, w_lo, w_hi, add_h, add_w


def plan_tiles(h: int, w: int, num_patch: Tuple[int, int], sf: int = 2, pb: int = PATCH_BOUNDARY) -> List[Tile]:
    """All tiles of an h x w (already cropped) frame, in the reference's order (w fastest)."""
    nh, nw = num_patch
    if h % (32 * nh) or w % (32 * nw):
        raise ValueError(f"{h}x{w} is not a multiple of 32*num_patch {num_patch}; crop first (crop_hw)")
    sH, sW = h // nh, w // nw
    tiles = []
    for p in range(nh * nw):
        pH, pW = p // nw, p % nw
        h_lo, h_hi, w_lo, w_hi, _, _ = get_hw_boundary(pb, h, w, pH, sH, pW, sW)
        # trim_patch_boundary (utils.py:138-159): drop pb*sf HR pixels on sides that have a halo
        top = pb * sf if (pb and not pH * sH < pb) else 0
        bot = pb * sf if (pb and not (pH + 1) * sH + pb > h) else 0
        left = pb * sf if (pb and not pW * sW < pb) else 0
        right = pb * sf if (pb and not (pW + 1) * sW + pb > w) else 0
        out_h = (h_hi - h_lo) * sf - top - bot
        out_w = (w_hi - w_lo) * sf - left - right
        if out_h != sH * sf or out_w != sW * sf:
            # the reference's numpy assignment would raise a broadcast error here
            raise ValueError(f"tile {p}: trimmed size {out_h}x{out_w} != patch {sH * sf}x{sW * sf}")
        tiles.append(Tile(p, h_lo, h_hi, w_lo, w_hi, top, left, pH * sH * sf, pW * sW * sf, out_h, out_w))
    return tiles
