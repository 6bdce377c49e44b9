"""GPU parity of the LDS-DMA fp16 conv kernel (conv3x3_dma.h; run with `-m gpu`): FISR_PREC_F16 routes every convolution with
Cout > 32 to it.  Against (a) the fp64 oracle conv on inputs / weights already rounded to fp16 -- only the fp32 accumulation
order and the final rounding of the output to fp16 are left: <= 1 fp16 ulp of the result -- and (b) round 1's register-staged
kernel (FISR_PREC_F16R), which computes the same products in another order."""
import ctypes

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")

import fisr_oracle as O  # noqa: E402
from fisr_amd import lib as flib  # noqa: E402

F32P = ctypes.POINTER(ctypes.c_float)


@pytest.fixture(scope="module", autouse=True)
def _gpu():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")


def _conv(prec_id, x0, w, b, x1=None, res=None, flags=0, in_place=False):
    n, h, wd, c0 = x0.shape
    cout = w.shape[3]
    d0 = torch.from_numpy(x0).cuda().half().contiguous()
    d1 = torch.from_numpy(x1).cuda().half().contiguous() if x1 is not None else None
    dr = torch.from_numpy(res).cuda().half().contiguous() if res is not None else None
    oshape = (n, 2 * h, 2 * wd, cout // 4) if flags & flib.CONV_D2S else (n, h, wd, cout)
    out = dr if in_place else torch.full(oshape, float("nan"), dtype=torch.float16, device="cuda")
    wc, bc = np.ascontiguousarray(w, np.float32), np.ascontiguousarray(b, np.float32)
    rc = flib.lib().fisr_op_conv3x3(ctypes.c_void_p(d0.data_ptr()), c0, ctypes.c_void_p(d1.data_ptr() if d1 is not None else 0),
                                    x1.shape[3] if x1 is not None else 0, wc.ctypes.data_as(F32P), bc.ctypes.data_as(F32P), cout,
                                    ctypes.c_void_p(dr.data_ptr() if dr is not None else 0), ctypes.c_void_p(out.data_ptr()),
                                    n, h, wd, flags, prec_id, 0, ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
    flib.check(rc)
    torch.cuda.synchronize()
    return out.float().cpu().numpy()


def _ref(x0, w, b, x1, res, flags):
    h16 = lambda a: a.astype(np.float16).astype(np.float64)
    x = h16(x0) if x1 is None else np.concatenate([h16(x0), h16(x1)], axis=3)
    if flags & flib.CONV_RELU_IN:
        x = O.relu(x)
    y = O.conv2d(x, h16(w), b.astype(np.float64))
    if res is not None:
        y = h16(res) + y
    if flags & flib.CONV_RELU_OUT:
        y = O.relu(y)
    if flags & flib.CONV_D2S:
        y = O.depth_to_space2(y)
    return y


@pytest.mark.parametrize("shape", [
    # n, h, w, c0, c1, cout, flags, use_res
    (1, 8, 64, 16, 0, 64, 0, False),            # exactly one 8x64 tile, one chunk
    (1, 8, 64, 64, 0, 64, 3, True),             # 4 chunks (both LDS stages twice), relu in/out + residual
    (1, 16, 128, 48, 0, 64, 1, False),          # 2x2 tiles, odd chunk count (the loop leaves through its middle exit)
    (2, 24, 24, 64, 0, 128, 1, False),          # batch 2, ragged width, two N blocks
    (1, 3, 3, 32, 0, 64, 0, False),             # deepest level of the 96x96 config: a tile that is almost all padding
    (1, 17, 45, 16, 0, 64, 0, False),           # odd sizes: masks on both axes
    (1, 10, 133, 32, 0, 96, 2, False),          # ragged tiles, Cout = 1.5 N blocks
    (1, 16, 40, 64, 64, 64, 0, False),          # dual-source concat (decoder conv/0)
    (1, 12, 70, 128, 128, 128, 3, True),        # concat + residual + relus
    (1, 8, 64, 64, 0, 256, 7, False),           # relu, relu, depth_to_space store (heads conv/1)
    (1, 10, 33, 64, 0, 256, 6, False),          # d2s with ragged tile
    (1, 8, 8, 512, 0, 512, 0, True),            # bottleneck shape: 32 chunks, 8 N blocks
    (3, 68, 124, 256, 0, 256, 3, True),         # a level-3 / R/8-sized layer: many tiles per XCD range
])
def test_conv3x3_dma_fp16_vs_oracle_and_vs_register_staged_kernel(shape):
    n, h, wd, c0, c1, cout, flags, use_res = shape
    rng = np.random.default_rng(abs(hash(shape)) % (2 ** 31))
    x0 = rng.standard_normal((n, h, wd, c0)).astype(np.float32)
    x1 = rng.standard_normal((n, h, wd, c1)).astype(np.float32) if c1 else None
    w = (rng.standard_normal((3, 3, c0 + c1, cout)) * np.sqrt(2.0 / (9 * (c0 + c1)))).astype(np.float32)
    b = (rng.standard_normal(cout) * 0.1).astype(np.float32)
    res = rng.standard_normal((n, h, wd, cout)).astype(np.float32) if use_res else None
    got = _conv(flib.PREC_F16, x0, w, b, x1, res, flags)
    # (the register-staged kernel works in 32-channel chunks)
    old = _conv(flib.PREC_F16R, x0, w, b, x1, res, flags) if c0 % 32 == 0 and c1 % 32 == 0 else got
    exp = _ref(x0, w, b, x1, res, flags)
    assert not np.isnan(got).any()
    ulp = np.maximum(np.abs(exp), 2.0 ** -14) * 2.0 ** -10          # one fp16 ulp of the expected value (>= the smallest normal's)
    err = np.abs(got - exp)
    print(f"dma fp16 {shape}: max|err| {err.max():.3e} = {np.max(err / ulp):.2f} ulp; vs the register-staged kernel {np.abs(got - old).max():.3e}")
    assert (err <= 0.51 * ulp + 3e-6 * np.abs(exp).max()).all(), float(np.max(err / ulp))
    assert (np.abs(got - old) <= ulp).all()


def test_conv3x3_dma_in_place_residual():
    """res_block conv/1 writes its result over its residual input (ops.py:43): every record is read before it is written."""
    rng = np.random.default_rng(3)
    x = rng.standard_normal((2, 20, 70, 64)).astype(np.float32)
    w = (rng.standard_normal((3, 3, 64, 64)) * 0.06).astype(np.float32)
    b = (rng.standard_normal(64) * 0.1).astype(np.float32)
    res = rng.standard_normal((2, 20, 70, 64)).astype(np.float32)
    a = _conv(flib.PREC_F16, x, w, b, None, res, flib.CONV_RELU_OUT)
    c = _conv(flib.PREC_F16, x, w, b, None, res, flib.CONV_RELU_OUT, in_place=True)
    assert np.array_equal(a, c)


def test_fp16_engine_runs_the_dma_kernel_and_matches_the_register_staged_engine():
    """Whole forward: FISR_PREC_F16 (132 convolutions on the LDS-DMA kernel) against FISR_PREC_F16R and against the fp32 engine."""
    from fisr_amd import weights
    from fisr_amd.fisrnet import FISRnet
    from tests_support import make_full_size_input
    W = weights.synthetic_weights(2020)
    x = torch.from_numpy(make_full_size_input(21, 96, 160, 2)).cuda()
    outs, prof = {}, {}
    # (the register-staged engine is an A/B engine of the diagnostics build since r06: compared only when that library is loaded)
    diag = b"DIAG" in flib.lib().fisr_version()
    for prec in ("fp32", "fp16") + (("fp16r",) if diag else ()):
        net = FISRnet(device="cuda:0", precision=prec)
        net.set_weights(W)
        net.profile(1)
        outs[prec] = [t.float().cpu().numpy() for t in net.model(x)]
        torch.cuda.synchronize()
        prof[prec] = {p["name"]: p["launches"] for p in net.profile_read()}
        net.close()
    assert prof["fp16"].get("conv3x3_dma<f16>", 0) == 132 and (not diag or "conv3x3_dma<f16>" not in prof["fp16r"])
    for k, name in enumerate(("pred_l1", "pred_l2", "pred_l3")):
        d_32 = np.sqrt(((outs["fp16"][k] - outs["fp32"][k]) ** 2).mean())
        print(f"{name}: rms vs fp32: dma {d_32:.2e}")
        assert d_32 < 8e-4                        # fp16's rounding noise on this network (measured 2e-4 .. 4e-4), not a wrong tap (>= 1e-2)
        if diag:
            d_old = np.abs(outs["fp16"][k] - outs["fp16r"][k])
            d_32r = np.sqrt(((outs["fp16r"][k] - outs["fp32"][k]) ** 2).mean())
            print(f"{name}: |dma - register-staged| max {d_old.max():.2e}; rms vs fp32: register-staged {d_32r:.2e}")
            assert d_32 < 1.3 * d_32r + 1e-5      # the same rounding noise as the old kernel's, not more
