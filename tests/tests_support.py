"""Shared helpers for the tests (seeded synthetic inputs that can be regenerated on any box)."""
import numpy as np


def make_full_size_input(seed: int, h: int, w: int, n: int = 1) -> np.ndarray:
    """[n,h,w,29] float32: 21 image-like channels in [0,1] (coarse structure + fine noise) and
    8 flow channels in [-0.2, 0.2], laid out as FISRnet.py:843 (9 img, 8 flow, 12 warp)."""
    rng = np.random.default_rng(seed)
    coarse = rng.random((n, h // 16 + 1, w // 16 + 1, 29)).astype(np.float32)
    x = np.repeat(np.repeat(coarse, 16, axis=1), 16, axis=2)[:, :h, :w, :]
    x = 0.75 * x + 0.25 * rng.random((n, h, w, 29), dtype=np.float32)
    x[..., 9:17] = (x[..., 9:17] - 0.5) * 0.4
    return np.ascontiguousarray(x, np.float32)


def make_flow_frames(seed: int, h: int, w: int):
    """Two YUV uint8 frames [h,w,3] with a real displacement between them: one blocky texture (8-px cells) seen through
    two different integer shifts plus independent fine noise -- the inputs of the full-size optical-flow golden
    (oracle/make_golden_pwc_fullsize.py) and of the test that checks it."""
    rng = np.random.default_rng(seed)
    base = rng.integers(16, 236, (h // 8 + 4, w // 8 + 4, 3)).astype(np.float32)
    img = np.repeat(np.repeat(base, 8, axis=0), 8, axis=1)
    a = img[12:12 + h, 12:12 + w] * 0.85 + rng.integers(0, 36, (h, w, 3)).astype(np.float32)
    b = img[9:9 + h, 17:17 + w] * 0.85 + rng.integers(0, 36, (h, w, 3)).astype(np.float32)
    return np.clip(a, 0, 255).astype(np.uint8), np.clip(b, 0, 255).astype(np.uint8)
