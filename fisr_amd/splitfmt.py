"""Host-side (numpy) view of the split-bf16 activation format used by FISR_PREC_BF16X3.

A value x is held as hi + lo with hi = bf16(x), lo = bf16(x - hi) (round-to-nearest-even), so
x ~ hi + lo to 2^-18 relative.  A tensor [..., C] (C % 16 == 0) is stored per pixel as C/16 groups
of {16 x bf16 hi (32 B), 16 x bf16 lo (32 B)}  (fisr_amd/csrc/conv3x3.h).  These helpers are for
tests and debugging; the kernels convert on the device.
"""
import numpy as np


def bf16_bits(x: np.ndarray) -> np.ndarray:
    u = np.ascontiguousarray(x, np.float32).view(np.uint32)
    return ((u + np.uint32(0x7FFF) + ((u >> np.uint32(16)) & np.uint32(1))) >> np.uint32(16)).astype(np.uint16)


def bf16_to_f32(b: np.ndarray) -> np.ndarray:
    return (b.astype(np.uint32) << np.uint32(16)).view(np.float32)


def to_split(x: np.ndarray) -> np.ndarray:
    """float32 [..., C] -> uint16 [..., C/16, 2, 16] (byte-exact device layout)."""
    x = np.ascontiguousarray(x, np.float32)
    c = x.shape[-1]
    assert c % 16 == 0
    hi = bf16_bits(x)
    lo = bf16_bits(x - bf16_to_f32(hi))
    g = x.shape[:-1] + (c // 16, 16)
    return np.ascontiguousarray(np.stack([hi.reshape(g), lo.reshape(g)], axis=-2))


def from_split(s: np.ndarray) -> np.ndarray:
    """uint16 [..., C/16, 2, 16] -> float32 [..., C]."""
    hi = bf16_to_f32(s[..., 0, :])
    lo = bf16_to_f32(s[..., 1, :])
    v = hi + lo
    return v.reshape(v.shape[:-2] + (v.shape[-2] * 16,))


def split_round(x: np.ndarray) -> np.ndarray:
    """x rounded to what the split format can hold (hi + lo)."""
    return from_split(to_split(x))
