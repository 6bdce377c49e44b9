"""GPU parity of the persistent LDS-DMA f16f8 conv kernel (conv3x3_dma_fs.h, round 5; run with `-m gpu`): FISR_PREC_F16F8 routes every
convolution with Cout % 64 == 0 to it.  Against (a) the fp64 oracle conv on inputs / weights already rounded to what the f16f8 format
holds -- what is left is the 4-bit cross terms (~2^-16 of a product), the fp32 accumulation order and the rounding of the stored
output -- and (b) round 1's register-staged kernel (FISR_PREC_F16F8R), which computes the same products in another order.  Shapes
with more items than CUs walk the persistent loop (cross-item prefetch, both LDS stage parities at an item's start)."""
import ctypes

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")

import fisr_oracle as O  # noqa: E402
from fisr_amd import lib as flib  # noqa: E402
from fisr_amd import splitfmt  # noqa: E402

F32P = ctypes.POINTER(ctypes.c_float)


@pytest.fixture(scope="module", autouse=True)
def _gpu():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")


def _dev(x):
    return torch.from_numpy(splitfmt.to_fsplit(np.ascontiguousarray(x, np.float32))).cuda()


def _host(t, shape):
    return splitfmt.from_fsplit(t.cpu().numpy().reshape(shape[:-1] + (shape[-1] // 16, 64)))


def _conv(prec_id, x0, w, b, x1=None, res=None, flags=0, in_place=False, pool=False):
    n, h, wd, c0 = x0.shape
    cout = w.shape[3]
    d0 = _dev(x0)
    d1 = _dev(x1) if x1 is not None else None
    dr = _dev(res) if res is not None else None
    oshape = (n, 2 * h, 2 * wd, cout // 4) if flags & flib.CONV_D2S else (n, h, wd, cout)
    out = dr if in_place else torch.full(oshape[:-1] + (oshape[-1] // 16, 64), 0x7e, dtype=torch.uint8, device="cuda")
    wc, bc = np.ascontiguousarray(w, np.float32), np.ascontiguousarray(b, np.float32)
    L = flib.lib()
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    args = [ctypes.c_void_p(d0.data_ptr()), c0, ctypes.c_void_p(d1.data_ptr() if d1 is not None else 0),
            x1.shape[3] if x1 is not None else 0, wc.ctypes.data_as(F32P), bc.ctypes.data_as(F32P), cout,
            ctypes.c_void_p(dr.data_ptr() if dr is not None else 0), ctypes.c_void_p(out.data_ptr())]
    if pool:
        po = torch.full((n, h // 2, wd // 2, cout // 16, 64), 0x7e, dtype=torch.uint8, device="cuda")
        flib.check(L.fisr_op_conv3x3_pool(*args, ctypes.c_void_p(po.data_ptr()), n, h, wd, flags, prec_id, st))
        torch.cuda.synchronize()
        return _host(out, oshape), _host(po, (n, h // 2, wd // 2, cout))
    flib.check(L.fisr_op_conv3x3(*args, n, h, wd, flags, prec_id, 0, st))
    torch.cuda.synchronize()
    return _host(out, oshape)


def _fs(a):
    return splitfmt.from_fsplit(splitfmt.to_fsplit(np.ascontiguousarray(a, np.float32))).astype(np.float64)


def _w14(w):
    """what the packed weights hold: w_h + fp8 remainder (>= 14 significant bits; the fp8 copy of w_h serves the cross term only)"""
    h = w.astype(np.float16).astype(np.float64)
    return h + (w.astype(np.float64) - h)          # the remainder's own rounding (4 bits of a 2^-11 term) is below the test's tolerance


def _ref(x0, w, b, x1, res, flags):
    x = _fs(x0) if x1 is None else np.concatenate([_fs(x0), _fs(x1)], axis=3)
    if flags & flib.CONV_RELU_IN:
        x = O.relu(x)
    y = O.conv2d(x, _w14(w), b.astype(np.float64))
    if res is not None:
        y = _fs(res) + y
    if flags & flib.CONV_RELU_OUT:
        y = O.relu(y)
    if flags & flib.CONV_D2S:
        y = O.depth_to_space2(y)
    return y


SHAPES = [
    # n, h, w, c0, c1, cout, flags, use_res
    (1, 8, 64, 16, 0, 64, 0, False),            # exactly one 8x64 tile, one chunk
    (1, 8, 64, 64, 0, 64, 3, True),             # 4 chunks (both LDS stages twice), relu in/out + residual
    (1, 16, 128, 48, 0, 64, 1, False),          # 2x2 tiles, odd chunk count (the first conv of level 3: 48 padded channels)
    (2, 24, 24, 64, 0, 128, 1, False),          # batch 2, ragged width, two N blocks
    (1, 3, 3, 32, 0, 64, 0, False),             # a tile that is almost all padding
    (1, 17, 45, 16, 0, 64, 0, False),           # odd sizes: masks on both axes
    (1, 16, 40, 64, 64, 64, 0, False),          # dual-source concat (decoder conv/0)
    (1, 12, 70, 128, 128, 128, 3, True),        # concat + residual + relus
    (1, 8, 64, 64, 0, 256, 7, False),           # relu, relu, depth_to_space store (heads conv/1)
    (1, 10, 33, 64, 0, 256, 6, False),          # d2s with ragged tile
    (1, 8, 8, 512, 0, 512, 0, True),            # 32 chunks, 8 N blocks
    (3, 160, 200, 64, 0, 128, 3, True),         # 480 items on 256 workgroups: the persistent loop, residual + relus
    (5, 100, 130, 48, 0, 64, 2, False),         # 195 tiles... x 1 block; odd chunk count flips the stage parity between items
    (4, 200, 260, 48, 0, 64, 1, False),         # 500 items, three chunks each: items start on either LDS stage
]


@pytest.mark.parametrize("shape", SHAPES)
def test_conv3x3_dma_f16f8_vs_oracle_and_vs_register_staged_kernel(shape):
    n, h, wd, c0, c1, cout, flags, use_res = shape
    rng = np.random.default_rng(abs(hash(shape)) % (2 ** 31))
    x0 = rng.standard_normal((n, h, wd, c0)).astype(np.float32)
    x1 = rng.standard_normal((n, h, wd, c1)).astype(np.float32) if c1 else None
    w = (rng.standard_normal((3, 3, c0 + c1, cout)) * np.sqrt(2.0 / (9 * (c0 + c1)))).astype(np.float32)
    b = (rng.standard_normal(cout) * 0.1).astype(np.float32)
    res = rng.standard_normal((n, h, wd, cout)).astype(np.float32) if use_res else None
    got = _conv(flib.PREC_F16F8, x0, w, b, x1, res, flags)
    old = _conv(flib.PREC_F16F8R, x0, w, b, x1, res, flags)
    exp = _ref(x0, w, b, x1, res, flags)
    assert not np.isnan(got).any()
    err, err_old = np.abs(got - exp), np.abs(old - exp)
    rms, rms_old = np.sqrt((err ** 2).mean()), np.sqrt((err_old ** 2).mean())
    print(f"dma f16f8 {shape}: max|err| {err.max():.3e} rms {rms:.3e}; register-staged kernel max {err_old.max():.3e} rms {rms_old:.3e}; "
          f"|new - old| max {np.abs(got - old).max():.3e}")
    # the operands are exact here (the reference convolves what the format holds): what remains is the 4-bit cross terms and the
    # rounding of the stored result (2^-15 relative)
    scale = max(1.0, float(np.abs(exp).max()))
    assert err.max() < 2.5e-4 * scale and rms < 4e-5 * scale
    assert rms < 1.3 * rms_old + 1e-6              # the same rounding noise as the old kernel's, not more
    if flags & flib.CONV_RELU_OUT:
        assert got.min() >= 0


def test_conv3x3_f16f8_d2s_with_residual_falls_back_to_the_direct_kernel():
    """depth_to_space + residual: the persistent kernel would read the residual in the plain layout while storing in the
    shuffled one -- dmafs_takes() refuses it, so FISR_PREC_F16F8 runs the call on the direct kernel and must give, bit for bit, what
    FISR_PREC_F16F8R (that kernel by name) gives, and the oracle's d2s(conv + res) to the format's tolerance."""
    shape = (1, 16, 64, 64, 0, 256, flib.CONV_D2S | flib.CONV_RELU_OUT, True)
    n, h, wd, c0, c1, cout, flags, _ = shape
    rng = np.random.default_rng(41)
    x0 = rng.standard_normal((n, h, wd, c0)).astype(np.float32)
    w = (rng.standard_normal((3, 3, c0, cout)) * np.sqrt(2.0 / (9 * c0))).astype(np.float32)
    b = (rng.standard_normal(cout) * 0.1).astype(np.float32)
    res = rng.standard_normal((n, h, wd, cout)).astype(np.float32)
    got = _conv(flib.PREC_F16F8, x0, w, b, None, res, flags)
    old = _conv(flib.PREC_F16F8R, x0, w, b, None, res, flags)
    exp = _ref(x0, w, b, None, res, flags)
    assert got.shape == (n, 2 * h, 2 * wd, cout // 4) and np.array_equal(got, old)
    assert np.abs(got - exp).max() < 2.5e-4 * max(1.0, float(np.abs(exp).max()))


def test_conv3x3_dma_f16f8_in_place_residual():
    """res_block conv/1 writes its result over its residual input (ops.py:43): every record is read before it is written."""
    rng = np.random.default_rng(3)
    x = rng.standard_normal((3, 90, 200, 64)).astype(np.float32)
    w = (rng.standard_normal((3, 3, 64, 64)) * 0.06).astype(np.float32)
    b = (rng.standard_normal(64) * 0.1).astype(np.float32)
    res = rng.standard_normal((3, 90, 200, 64)).astype(np.float32)
    a = _conv(flib.PREC_F16F8, x, w, b, None, res, flib.CONV_RELU_OUT)
    c = _conv(flib.PREC_F16F8, x, w, b, None, res, flib.CONV_RELU_OUT, in_place=True)
    assert np.array_equal(a, c)
    assert np.array_equal(a, _conv(flib.PREC_F16F8, x, w, b, None, res, flib.CONV_RELU_OUT))       # and launches repeat bit for bit


@pytest.mark.parametrize("shape", [(1, 16, 64, 64, 64), (2, 36, 130, 64, 128), (3, 160, 200, 32, 64)])
def test_conv3x3_dma_f16f8_pooled_second_store(shape):
    """ops.py:54 as a second store of the level's last conv (residual + relu): the full-resolution store bit-identical to the plain
    launch, the pooled map value for value the 2x2 maxima of the stored map."""
    n, h, wd, c, cout = shape
    rng = np.random.default_rng(11 + n)
    x = rng.standard_normal((n, h, wd, c)).astype(np.float32)
    w = (rng.standard_normal((3, 3, c, cout)) * np.sqrt(2.0 / (9 * c))).astype(np.float32)
    b = (rng.standard_normal(cout) * 0.1).astype(np.float32)
    res = rng.standard_normal((n, h, wd, cout)).astype(np.float32)
    plain = _conv(flib.PREC_F16F8, x, w, b, None, res, flib.CONV_RELU_OUT)
    full, pooled = _conv(flib.PREC_F16F8, x, w, b, None, res, flib.CONV_RELU_OUT, pool=True)
    assert np.array_equal(full, plain)
    assert np.array_equal(pooled, O.max_pool2(full.astype(np.float64)).astype(np.float32))


def test_f16f8_engine_runs_the_dma_kernel_and_matches_the_register_staged_engine():
    """Whole forward: FISR_PREC_F16F8 (132 convolutions on the persistent kernel) against FISR_PREC_F16F8R and against the fp32 engine."""
    from fisr_amd import weights
    from fisr_amd.fisrnet import FISRnet
    from tests_support import make_full_size_input
    W = weights.synthetic_weights(2020)
    x = torch.from_numpy(make_full_size_input(21, 96, 160, 2)).cuda()
    outs, prof = {}, {}
    # (the register-staged engine is an A/B engine of the diagnostics build since r06: compared only when that library is loaded)
    diag = b"DIAG" in flib.lib().fisr_version()
    for prec in ("fp32", "f16f8") + (("f16f8r",) if diag else ()):
        net = FISRnet(device="cuda:0", precision=prec)
        net.set_weights(W)
        net.profile(1)
        outs[prec] = [t.float().cpu().numpy() for t in net.model(x)]
        torch.cuda.synchronize()
        prof[prec] = {p["name"]: p["launches"] for p in net.profile_read()}
        net.close()
    n_new = sum(v for k, v in prof["f16f8"].items() if k.startswith("conv3x3_dma_fs<"))
    assert n_new == 132 and (not diag or not any(k.startswith("conv3x3_dma_fs<") for k in prof["f16f8r"])), prof["f16f8"]
    for k, name in enumerate(("pred_l1", "pred_l2", "pred_l3")):
        d_32 = np.sqrt(((outs["f16f8"][k] - outs["fp32"][k]) ** 2).mean())
        print(f"{name}: rms vs fp32: dma {d_32:.2e}")
        assert d_32 < 4e-5                        # the split format's noise on this network (measured ~1e-5)
        if diag:
            d_old = np.abs(outs["f16f8"][k] - outs["f16f8r"][k])
            d_32r = np.sqrt(((outs["f16f8r"][k] - outs["fp32"][k]) ** 2).mean())
            print(f"{name}: |dma - register-staged| max {d_old.max():.2e}; rms vs fp32: register-staged {d_32r:.2e}")
            assert d_32 < 1.3 * d_32r + 2e-6
