"""GPU parity of the split-bf16 Winograd kernel (conv3x3_wino8b.h, round 6; run with `-m gpu`): FISR_PREC_F32WB -- fp32 tensors, fp32
F(2x2,3x3) transforms, the Winograd-domain products as bf16 pairs (Uh + Ul)(Vh + Vl) on v_mfma_f32_32x32x16_bf16 -- through
fisr_op_conv3x3 against the fp64 direct oracle (reference operator: ops.py:7-11 with the fused relu / residual / concat /
depth_to_space of ops.py:39-44, FISRnet.py:99) and against the fp32 Winograd kernel it shares everything but the products with.

Tolerance: both operands of a product carry 2^-17 relative (two bf16 roundings), the products are exact in the fp32 accumulator,
so a conv output of magnitude O(1..6) over K = 9 Cin products errs by a few 1e-6 rms; the bound is the gate the round-5 review set
for this kernel, max |err| <= 5e-5 (measured: see the prints), with rms <= 1e-5."""
import ctypes

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")

import fisr_oracle as O  # noqa: E402
from fisr_amd import lib as flib  # noqa: E402

F32P = ctypes.POINTER(ctypes.c_float)
WB_TOL = 5e-5
WB_RMS = 1e-5


@pytest.fixture(scope="module", autouse=True)
def _gpu():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")


def _conv(prec_id, x0, w, b, x1=None, res=None, flags=0, in_place=False):
    n, h, wd, c0 = x0.shape
    cout = w.shape[3]
    d0 = torch.from_numpy(np.ascontiguousarray(x0, np.float32)).cuda()
    d1 = torch.from_numpy(np.ascontiguousarray(x1, np.float32)).cuda() if x1 is not None else None
    dr = torch.from_numpy(np.ascontiguousarray(res, np.float32)).cuda() if res is not None else None
    oshape = (n, 2 * h, 2 * wd, cout // 4) if flags & flib.CONV_D2S else (n, h, wd, cout)
    out = dr if in_place else torch.full(oshape, float("nan"), dtype=torch.float32, device="cuda")
    wc, bc = np.ascontiguousarray(w, np.float32), np.ascontiguousarray(b, np.float32)
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    flib.check(flib.lib().fisr_op_conv3x3(ctypes.c_void_p(d0.data_ptr()), c0, ctypes.c_void_p(d1.data_ptr() if d1 is not None else 0),
                                         x1.shape[3] if x1 is not None else 0, wc.ctypes.data_as(F32P), bc.ctypes.data_as(F32P), cout,
                                         ctypes.c_void_p(dr.data_ptr() if dr is not None else 0), ctypes.c_void_p(out.data_ptr()),
                                         n, h, wd, flags, prec_id, 0, st))
    torch.cuda.synchronize()
    return out.cpu().numpy().reshape(oshape)


def _ref(x0, w, b, x1=None, res=None, flags=0):
    x = (x0 if x1 is None else np.concatenate([x0, x1], axis=3)).astype(np.float64)
    if flags & flib.CONV_RELU_IN:
        x = O.relu(x)
    y = O.conv2d(x, w, b)
    if res is not None:
        y = res.astype(np.float64) + y
    if flags & flib.CONV_RELU_OUT:
        y = O.relu(y)
    if flags & flib.CONV_D2S:
        y = O.depth_to_space2(y)
    return y


def _case(shape, seed):
    n, h, w, c0, c1, cout, flags, use_res = shape
    rng = np.random.default_rng(seed)
    x0 = rng.standard_normal((n, h, w, c0)).astype(np.float32)
    x1 = rng.standard_normal((n, h, w, c1)).astype(np.float32) if c1 else None
    wt = (rng.standard_normal((3, 3, c0 + c1, cout)) * np.sqrt(2.0 / (9 * (c0 + c1)))).astype(np.float32)
    b = rng.standard_normal(cout).astype(np.float32)
    res = rng.standard_normal((n, h, w, cout)).astype(np.float32) if use_res else None
    return x0, x1, wt, b, res


SHAPES = [
    # n, h, w, c0, c1, cout, flags, use_res
    (1, 8, 32, 32, 0, 64, 0, False),            # exactly one work item, four 8-channel chunks (the kernel's minimum)
    (1, 8, 32, 64, 0, 64, 0, False),            # 8 chunks
    (1, 16, 64, 32, 0, 64, 3, True),            # relu in/out + residual, 2x2 items
    (2, 24, 24, 64, 0, 128, 1, False),          # batch 2, ragged width, two N-blocks
    (1, 3, 3, 32, 0, 64, 0, False),             # a map below one tile row
    (1, 12, 12, 128, 0, 256, 2, False),
    (1, 17, 45, 32, 0, 64, 0, False),           # odd sizes: half-filled 2x2 output tiles on both axes
    (1, 16, 40, 64, 64, 64, 0, False),          # dual-source concat (decoder conv/0)
    (1, 8, 32, 64, 0, 256, 7, False),           # relu, relu, depth_to_space store (heads conv/1)
    (1, 10, 33, 64, 0, 256, 6, False),          # d2s with ragged tile
    (1, 8, 32, 48, 0, 64, 0, False),            # level-2/3 first conv (38 -> pad 48)
    (1, 8, 8, 512, 0, 512, 0, True),            # bottleneck shape, 64 chunks
    (3, 40, 100, 64, 0, 64, 3, True),           # more items than one round of the persistent loop's XCD order touches once
    (2, 136, 248, 64, 0, 64, 1, False),         # 1054 items on 256 workgroups: the cross-item copy / transform streams, both buffer parities
]


@pytest.mark.parametrize("shape", SHAPES)
def test_conv3x3_winob_vs_oracle(shape):
    x0, x1, wt, b, res = _case(shape, hash(shape) % (2 ** 31) + 6)
    flags = shape[6]
    got = _conv(flib.PREC_F32WB, x0, wt, b, x1, res, flags)
    exp = _ref(x0, wt, b, x1, res, flags)
    err = np.abs(got.astype(np.float64) - exp)
    print(f"winob conv {shape}: max {np.nanmax(err):.3e} rms {np.sqrt(np.nanmean(err ** 2)):.3e}")
    assert not np.isnan(got).any(), f"{int(np.isnan(got).sum())} outputs were never written"
    assert err.max() <= WB_TOL, f"max err {err.max():.3e} at {np.unravel_index(err.argmax(), err.shape)}"
    assert np.sqrt((err ** 2).mean()) <= WB_RMS
    if flags & flib.CONV_RELU_OUT:
        assert got.min() >= 0


@pytest.mark.parametrize("shape", [SHAPES[2], SHAPES[7], SHAPES[8], SHAPES[13]])
def test_conv3x3_winob_close_to_fp32_winograd(shape):
    """The same transforms, the same summation over chunks: what differs from FISR_PREC_F32W is the 2^-17 of the operand split."""
    x0, x1, wt, b, res = _case(shape, 99)
    a = _conv(flib.PREC_F32WB, x0, wt, b, x1, res, shape[6])
    f = _conv(flib.PREC_F32W, x0, wt, b, x1, res, shape[6])
    d = np.abs(a.astype(np.float64) - f)
    print(f"winob vs fp32w {shape}: max {d.max():.3e} rms {np.sqrt((d ** 2).mean()):.3e}")
    assert d.max() <= WB_TOL


def test_conv3x3_winob_residual_in_place():
    """res_block's conv/1 writes onto its residual (ops.py:43): every record is read before the lane that owns it writes it."""
    rng = np.random.default_rng(78)
    n, h, w, c = 1, 24, 72, 64
    x = rng.standard_normal((n, h, w, c)).astype(np.float32)
    r = rng.standard_normal((n, h, w, c)).astype(np.float32)
    wt = (rng.standard_normal((3, 3, c, c)) * np.sqrt(2.0 / (9 * c))).astype(np.float32)
    b = rng.standard_normal(c).astype(np.float32)
    got = _conv(flib.PREC_F32WB, x, wt, b, None, r, 0, in_place=True)
    err = np.abs(got - _ref(x, wt, b, None, r, 0))
    assert err.max() <= WB_TOL


def test_conv3x3_winob_large_values_keep_relative_accuracy():
    """bf16 has fp32's exponent range: inputs of magnitude 1e4 and 1e-4 lose nothing but the same 2^-17 (no fp16-style saturation)."""
    shape = (1, 16, 64, 64, 0, 64, 0, False)
    x0, _, wt, b, _ = _case(shape, 5)
    for s in (1e4, 1e-4):
        got = _conv(flib.PREC_F32WB, x0 * np.float32(s), wt, b * np.float32(s))
        exp = _ref(x0 * np.float32(s), wt, b * np.float32(s))
        assert np.abs(got - exp).max() <= WB_TOL * s
