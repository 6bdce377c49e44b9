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


# ----------------------------------------------------------------------------------------------
# fp16 + fp8 split ("f16f8", FISR_PREC_F16F8): per 16 channels 16 x fp16 h | l8, h8 of channels 0-7 | l8, h8 of channels 8-15
# (8 x fp8 each; conv3x3.h: the fp8 fields are interleaved per 8 channels)
# ----------------------------------------------------------------------------------------------
FS_LSHIFT = 14


def fp8_e4m3_encode(x: np.ndarray) -> np.ndarray:
    """float -> OCP fp8 e4m3fn bits, round-to-nearest-even, saturating at +-448 (as the kernels do)."""
    x = np.asarray(x, np.float64)
    sign = (np.signbit(x)).astype(np.uint8) << 7
    a = np.minimum(np.abs(x), 448.0)
    with np.errstate(divide="ignore"):
        ex = np.maximum(np.floor(np.log2(np.where(a > 0, a, 1.0))), -6).astype(np.int64)
    q = a * np.exp2(3 - ex)
    r = np.rint(q)
    bump = r >= 16
    r = np.where(bump, 8, r)
    ex = np.where(bump, ex + 1, ex)
    normal = r >= 8
    bits = np.where(normal, ((ex + 7) << 3) | (r.astype(np.int64) - 8), r.astype(np.int64))
    bits = np.where(a >= 448.0, 0x7E, bits)
    return (sign | bits.astype(np.uint8)).astype(np.uint8)


def fp8_e4m3_decode(b: np.ndarray) -> np.ndarray:
    b = np.asarray(b, np.uint8).astype(np.int64)
    e, m = (b >> 3) & 0xF, b & 7
    v = np.where(e == 0, m * 2.0 ** -9, (8 + m) * np.exp2(e - 10.0))
    return np.where(b & 0x80, -v, v).astype(np.float32)


def to_fsplit(x: np.ndarray) -> np.ndarray:
    """float32 [..., C] -> uint8 [..., C/16, 64] (byte-exact device layout of FISR_PREC_F16F8)."""
    x = np.ascontiguousarray(x, np.float32)
    c = x.shape[-1]
    assert c % 16 == 0
    g = x.reshape(x.shape[:-1] + (c // 16, 16))
    h = np.clip(g, -65504, 65504).astype(np.float16)          # saturating, like the device encoder
    hf = h.astype(np.float32)
    l8 = fp8_e4m3_encode(np.clip((g - hf) * np.float32(2 ** FS_LSHIFT), -448, 448))
    h8 = fp8_e4m3_encode(np.clip(hf, -448, 448))
    return np.ascontiguousarray(np.concatenate([h.view(np.uint8).reshape(g.shape[:-1] + (32,)), l8[..., :8], h8[..., :8], l8[..., 8:], h8[..., 8:]], axis=-1))


def from_fsplit(s: np.ndarray) -> np.ndarray:
    """uint8 [..., C/16, 64] -> float32 [..., C]  (x = h + l8 * 2^-14)."""
    s = np.ascontiguousarray(s, np.uint8)
    h = s[..., :32].copy().view(np.float16).astype(np.float32)
    l = fp8_e4m3_decode(np.concatenate([s[..., 32:40], s[..., 48:56]], axis=-1)) * np.float32(2.0 ** -FS_LSHIFT)
    v = h + l
    return v.reshape(v.shape[:-2] + (v.shape[-2] * 16,))
