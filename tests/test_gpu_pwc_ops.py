"""Op-level GPU parity of the flow network's kernels at the sizes the cfg5 bench runs them at (run with `-m gpu`).

At 2176x3840 (a x2 up-scaled 1080p frame, padded to 64) the levels are 544x960 (2), 272x480 (3), 136x240 (4), 68x120
(5), 34x60 (6).  test_gpu_pwc.py checks the whole network at <= 128x192, where a dilation-16 layer sees 2x3-pixel
sub-images and no sub-image spans more than one 8x32 Winograd tile; here every layer TYPE runs through its own kernel
on maps of the level-3 / level-4 size, so that every d x d sub-image of the dilated layers spans several tiles, with
channel-range inputs / outputs of a wider buffer as the dense blocks use them, against the float64 oracle ops of
oracle/pwcnet_oracle.py (model_pwcnet.py:1084-1100 stride-2 pyramid, :1178 warp, :1196 transpose conv, :1277 cost volume,
:1426-1449 dense estimator, :1506-1519 dilated context network)."""
import ctypes

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")

import pwcnet_oracle as P  # noqa: E402
from fisr_amd import lib as flib  # noqa: E402

F32P = ctypes.POINTER(ctypes.c_float)
I32P = ctypes.POINTER(ctypes.c_int)


@pytest.fixture(scope="module", autouse=True)
def _gpu():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _ptr(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else None


def _oracle_conv(x_nhwc, w, b, stride, dil, slope, add=None):
    """tf.layers.conv2d 'same' + leaky relu (+ add) in float64 through the oracle's conv2d_same."""
    W = {"op/kernel": w, "op/bias": b}
    y = P.conv2d_same(torch.from_numpy(x_nhwc).double().permute(0, 3, 1, 2), W, "op", stride=stride, dilation=dil)
    if slope != 1.0:
        y = torch.nn.functional.leaky_relu(y, slope)
    y = y.permute(0, 2, 3, 1).numpy()
    return y + add if add is not None else y


def _dev(a, prec):
    t = torch.from_numpy(np.ascontiguousarray(a, np.float32)).cuda()
    return t.half().contiguous() if prec == "fp16" else t


def _host(t):
    return t.float().cpu().numpy()


PID = {"fp32": flib.PREC_F32W, "fp32w4": flib.PREC_F32W4, "fp16": flib.PREC_F16}


def _run_conv(x_buf, in_co, cin_buf, w, b, chmap, out_buf, out_co, n, h, wd, stride, dil, slope, route, add_buf=None, add_co=0, prec="fp32"):
    ci, cout = w.shape[2], w.shape[3]
    wc = np.ascontiguousarray(w, np.float32)
    bc = np.ascontiguousarray(b, np.float32)
    cm = None if chmap is None else (ctypes.c_int * ci)(*[int(v) for v in chmap])
    out_f32 = int(out_buf.dtype == torch.float32)
    rc = flib.lib().fisr_pwc_op_conv(_ptr(x_buf), x_buf.shape[-1], in_co, cin_buf, wc.ctypes.data_as(F32P), bc.ctypes.data_as(F32P),
                                     ci, cout, cm, _ptr(out_buf), out_f32, out_buf.shape[-1], out_co, _ptr(add_buf),
                                     0 if add_buf is None else add_buf.shape[-1], add_co, n, h, wd, stride, dil, slope, route, PID[prec], _stream())
    flib.check(rc)
    torch.cuda.synchronize()
    return rc


# (h, w): level 3 and level 4 of the 2176x3840 bench frame, plus a ragged size (not a multiple of the 8x32 tile or of the dilation)
@pytest.mark.parametrize("dil,shape", [(1, (272, 480)), (2, (272, 480)), (4, (272, 480)), (8, (136, 240)), (16, (136, 240)),
                                       (16, (272, 480)), (4, (75, 133)), (8, (139, 251))])
def test_winograd_general_dilated_vs_oracle(dil, shape):
    """The GENERAL instantiation of the persistent Winograd kernel (97 % of the flow's FLOPs): dilation as d x d interleaved
    sub-images, every sub-image several 8x32 tiles wide and high, channel-range input and output, leaky relu, a Cout that is not a
    multiple of the 64-channel block."""
    h, wd = shape
    rng = np.random.default_rng(100 * dil + h)
    in_cs, in_co, cin = 160, 32, 96            # reads channels 32..127 of a 160-wide buffer
    out_cs, out_co, cout = 192, 64, 96         # writes channels 64..159 of a 192-wide buffer (96: one and a half N blocks)
    x = (rng.standard_normal((1, h, wd, in_cs)) * 0.5).astype(np.float32)
    w = (rng.standard_normal((3, 3, cin, cout)) * np.sqrt(2.0 / (9 * cin))).astype(np.float32)
    b = (rng.standard_normal(cout) * 0.05).astype(np.float32)
    xb = torch.from_numpy(x).cuda()
    ob = torch.full((1, h, wd, out_cs), 7.25, dtype=torch.float32, device="cuda")
    took = _run_conv(xb, in_co, cin, w, b, None, ob, out_co, 1, h, wd, 1, dil, 0.1, 2)
    assert took == 2
    got = ob.cpu().numpy()
    exp = _oracle_conv(x[..., in_co:in_co + cin], w, b, 1, dil, 0.1)
    err = np.abs(got[..., out_co:out_co + cout] - exp).max()
    print(f"wino GENERAL dil {dil} {h}x{wd}: max|err| {err:.2e} (|y| max {np.abs(exp).max():.2f})")
    assert err < 2e-5
    assert (got[..., :out_co] == 7.25).all() and (got[..., out_co + cout:] == 7.25).all()     # neighbours of the range untouched


@pytest.mark.parametrize("dil,shape", [(1, (272, 480)), (2, (272, 480)), (4, (272, 480)), (1, (136, 240)), (2, (139, 251)), (4, (300, 270)), (1, (75, 133))])
def test_winograd_f4_general_dilated_vs_oracle(dil, shape):
    """conv3x3_wf4_kernel<GENERAL> (r04: the fp32 flow engine's dense and context layers on F(4x4,3x3), route 5): the same layer
    as the F(2x2) test above -- dilation as d x d interleaved sub-images of several 16x32 items each, channel-range input and
    output, leaky relu, a Cout of one and a half 64-channel blocks, ragged sizes -- against the float64 oracle, and the channels
    next to the output range untouched.  (F(4,3)'s transforms: ~10x F(2x2)'s rounding, the bound of tests/test_gpu_parity.py.)"""
    h, wd = shape
    rng = np.random.default_rng(100 * dil + h + 7)
    in_cs, in_co, cin = 160, 32, 96
    out_cs, out_co, cout = 192, 64, 96
    x = (rng.standard_normal((1, h, wd, in_cs)) * 0.5).astype(np.float32)
    w = (rng.standard_normal((3, 3, cin, cout)) * np.sqrt(2.0 / (9 * cin))).astype(np.float32)
    b = (rng.standard_normal(cout) * 0.05).astype(np.float32)
    xb = torch.from_numpy(x).cuda()
    ob = torch.full((1, h, wd, out_cs), 7.25, dtype=torch.float32, device="cuda")
    took = _run_conv(xb, in_co, cin, w, b, None, ob, out_co, 1, h, wd, 1, dil, 0.1, 5, prec="fp32w4")
    assert took == 5
    got = ob.cpu().numpy()
    exp = _oracle_conv(x[..., in_co:in_co + cin], w, b, 1, dil, 0.1)
    err = np.abs(got[..., out_co:out_co + cout] - exp)
    print(f"F(4x4) GENERAL dil {dil} {h}x{wd}: max|err| {err.max():.2e} rms {np.sqrt((err ** 2).mean()):.2e} (|y| max {np.abs(exp).max():.2f})")
    assert err.max() < 1.2e-4 and np.sqrt((err ** 2).mean()) < 1e-5
    assert (got[..., :out_co] == 7.25).all() and (got[..., out_co + cout:] == 7.25).all()


def test_winograd_f4_general_batch_padded_groups_and_routing():
    """Two images, a dense block's padded channel groups (chmap), Cout = 32 (half a block; the other 32 channels of the block are
    computed from zero weights and never stored) on the F(4x4) route; and the engine's own choice (route 0): F(4x4) where the
    (sub-)image is at least 48 x 64, F(2x2) below -- a dilation of 16 on a 272 x 480 map leaves 17 x 30 sub-images."""
    h, wd, n = 136, 240, 2
    rng = np.random.default_rng(6)
    cin_buf = 128
    chmap = list(range(0, 40)) + list(range(48, 48 + 41)) + list(range(96, 96 + 30))
    ci, cout = len(chmap), 32
    x = (rng.standard_normal((n, h, wd, cin_buf)) * 0.5).astype(np.float32)
    x[..., 40:48] = 3.0
    w = (rng.standard_normal((3, 3, ci, cout)) * np.sqrt(2.0 / (9 * ci))).astype(np.float32)
    b = (rng.standard_normal(cout) * 0.05).astype(np.float32)
    xb = torch.from_numpy(x).cuda()
    ob = torch.zeros((n, h, wd, cout), dtype=torch.float32, device="cuda")
    assert _run_conv(xb, 0, cin_buf, w, b, chmap, ob, 0, n, h, wd, 1, 1, 0.1, 0, prec="fp32w4") == 5
    exp = _oracle_conv(x[..., chmap], w, b, 1, 1, 0.1)
    err = np.abs(ob.cpu().numpy() - exp).max()
    print(f"F(4x4) GENERAL batch 2, padded groups: max|err| {err:.2e}")
    assert err < 1.2e-4
    # routing by (sub-)image size
    x2 = torch.zeros((1, 272, 480, 64), device="cuda")
    o2 = torch.zeros((1, 272, 480, 64), device="cuda")
    w2, b2 = np.zeros((3, 3, 64, 64), np.float32), np.zeros(64, np.float32)
    for dil, want in ((1, 5), (4, 5), (8, 2), (16, 2)):
        assert _run_conv(x2, 0, 64, w2, b2, None, o2, 0, 1, 272, 480, 1, dil, 0.1, 0, prec="fp32w4") == want, dil
    assert _run_conv(x2, 0, 64, w2, b2, None, o2, 0, 1, 272, 480, 1, 1, 0.1, 0, prec="fp32") == 2        # FISR_PREC_F32W keeps F(2x2)


def test_winograd_f4_general_padding_waves_idle_is_bit_identical():
    """Late r04 (conv3x3_wf4.h, pad_wave): in a layer with at most 32 output channels the waves whose 16-channel quarter is padding run
    without fragment reads, MFMAs and epilogue, and the quarters of the second tile half are rotated so that every SIMD keeps one
    multiplying wave.  The multiplying waves must produce exactly what they produce when all eight waves multiply: the Cout = 32
    launch against the leading channels of a Cout = 64 launch with the same weights zero-padded (ragged map, dilation 1 / 2,
    channel-range output with untouched neighbours)."""
    rng = np.random.default_rng(16)
    h, wd, cin = 107, 150, 64            # (dilation 2: 54 x 75 sub-images, still the F(4x4) kernel's)
    x = (rng.standard_normal((1, h, wd, cin)) * 0.5).astype(np.float32)
    xb = torch.from_numpy(x).cuda()
    for cout, dil in ((32, 1), (32, 2)):
        w = (rng.standard_normal((3, 3, cin, cout)) * np.sqrt(2.0 / (9 * cin))).astype(np.float32)
        b = (rng.standard_normal(cout) * 0.05).astype(np.float32)
        w64 = np.zeros((3, 3, cin, 64), np.float32); w64[..., :cout] = w
        b64 = np.zeros(64, np.float32); b64[:cout] = b
        full = torch.zeros((1, h, wd, 64), dtype=torch.float32, device="cuda")
        assert _run_conv(xb, 0, cin, w64, b64, None, full, 0, 1, h, wd, 1, dil, 0.1, 5, prec="fp32w4") == 5
        part = torch.full((1, h, wd, 80), 7.25, dtype=torch.float32, device="cuda")
        assert _run_conv(xb, 0, cin, w, b, None, part, 8, 1, h, wd, 1, dil, 0.1, 5, prec="fp32w4") == 5
        got = part.cpu().numpy()
        assert np.array_equal(got[..., 8:8 + cout], full.cpu().numpy()[..., :cout]), (cout, dil)
        assert (got[..., :8] == 7.25).all() and (got[..., 8 + cout:] == 7.25).all()
        exp = _oracle_conv(x, w, b, 1, dil, 0.1)
        assert np.abs(got[..., 8:8 + cout] - exp).max() < 1.2e-4


def test_winograd_general_batch_and_padded_groups_vs_oracle():
    """Two images (the two flow directions of the feature pyramid run as one batch), a dense block's input with zero-weight
    padding channels between its groups (chmap), Cout = 32 (half an N block)."""
    h, wd, n = 136, 240, 2
    rng = np.random.default_rng(5)
    cin_buf = 128
    chmap = list(range(0, 40)) + list(range(48, 48 + 41)) + list(range(96, 96 + 30))        # three groups with gaps
    ci, cout = len(chmap), 32
    x = (rng.standard_normal((n, h, wd, cin_buf)) * 0.5).astype(np.float32)
    x[..., 40:48] = np.nan_to_num(x[..., 40:48]) * 0 + 3.0         # padding channels hold finite garbage: their weights are zero
    w = (rng.standard_normal((3, 3, ci, cout)) * np.sqrt(2.0 / (9 * ci))).astype(np.float32)
    b = (rng.standard_normal(cout) * 0.05).astype(np.float32)
    xb = torch.from_numpy(x).cuda()
    ob = torch.zeros((n, h, wd, cout), dtype=torch.float32, device="cuda")
    took = _run_conv(xb, 0, cin_buf, w, b, chmap, ob, 0, n, h, wd, 1, 1, 0.1, 0)
    assert took == 2
    exp = _oracle_conv(x[..., chmap], w, b, 1, 1, 0.1)
    err = np.abs(ob.cpu().numpy() - exp).max()
    print(f"wino GENERAL batch 2, padded groups: max|err| {err:.2e}")
    assert err < 2e-5


@pytest.mark.parametrize("shape", [(272, 480), (271, 479), (136, 241), (137, 240)])
def test_stride2_generic_kernel_vs_oracle(shape):
    """pwc_convg_kernel, stride 2 (the pyramid's conv{l}a, model_pwcnet.py:1092): TF 'SAME' pads (0, 1) on an even size and (1, 1)
    on an odd one."""
    h, wd = shape
    rng = np.random.default_rng(h * 7 + wd)
    cin, cout = 32, 64
    x = (rng.standard_normal((2, h, wd, cin)) * 0.5).astype(np.float32)
    w = (rng.standard_normal((3, 3, cin, cout)) * np.sqrt(2.0 / (9 * cin))).astype(np.float32)
    b = (rng.standard_normal(cout) * 0.05).astype(np.float32)
    oh, ow = -(-h // 2), -(-wd // 2)
    ob = torch.zeros((2, oh, ow, cout), dtype=torch.float32, device="cuda")
    xb = torch.from_numpy(x).cuda()
    took = _run_conv(xb, 0, cin, w, b, None, ob, 0, 2, h, wd, 2, 1, 0.1, 0)
    assert took == 1
    exp = _oracle_conv(x, w, b, 2, 1, 0.1)
    assert exp.shape == (2, oh, ow, cout)
    err = np.abs(ob.cpu().numpy() - exp).max()
    print(f"stride 2 {h}x{wd}: max|err| {err:.2e}")
    assert err < 2e-5


@pytest.mark.parametrize("prec", ["fp32", "fp16"])
def test_conv1a_mfma_kernel_vs_oracle(prec):
    """pwc_conv1a_kernel (route 7; model_pwcnet.py:1092: 3 -> 16 channels, stride 2, leaky relu) on one fp32 MFMA per tap: a pixel count
    that is not a multiple of the 16-pixel MFMA group, batch 3, the unused fourth channel of the input holding garbage."""
    n, h, wd = 3, 26, 38
    rng = np.random.default_rng(5)
    x = (rng.standard_normal((n, h, wd, 4)) * 0.5).astype(np.float32)
    x[..., 3] = 1e3
    w = (rng.standard_normal((3, 3, 3, 16)) * 0.2).astype(np.float32)
    b = (rng.standard_normal(16) * 0.05).astype(np.float32)
    dt = torch.float32 if prec == "fp32" else torch.float16
    xb = torch.from_numpy(x).cuda().to(dt)
    ob = torch.zeros((n, h // 2, wd // 2, 16), dtype=dt, device="cuda")
    took = _run_conv(xb, 0, 4, w, b, None, ob, 0, n, h, wd, 2, 1, 0.1, 0, prec=prec)
    assert took == 7
    exp = _oracle_conv(_host(xb)[..., :3], w, b, 2, 1, 0.1)
    err = np.abs(_host(ob) - exp).max()
    print(f"conv1a {prec}: max|err| {err:.2e}")
    assert err < (2e-5 if prec == "fp32" else 4e-3)
    ob2 = torch.zeros_like(ob)
    assert _run_conv(xb, 0, 4, w, b, None, ob2, 0, n, h, wd, 2, 1, 0.1, 1, prec=prec) == 1      # the generic kernel agrees
    assert np.abs(_host(ob2) - _host(ob)).max() < (2e-5 if prec == "fp32" else 4e-3)


def test_generic_kernel_residual_add_and_dilation_vs_oracle():
    """dc_conv7 (2 channels, linear, + the predicted flow: refine_flow model_pwcnet.py:1519-1521) and a dilated layer forced
    through the generic kernel (the Winograd path's fall-back), 272x480."""
    h, wd = 272, 480
    rng = np.random.default_rng(11)
    cin = 32
    x = (rng.standard_normal((1, h, wd, cin)) * 0.5).astype(np.float32)
    w = (rng.standard_normal((3, 3, cin, 2)) * 0.05).astype(np.float32)
    b = (rng.standard_normal(2) * 0.05).astype(np.float32)
    flow = (rng.standard_normal((1, h, wd, 4)) * 2).astype(np.float32)
    ob = torch.zeros((1, h, wd, 4), dtype=torch.float32, device="cuda")
    xb, fb = torch.from_numpy(x).cuda(), torch.from_numpy(flow).cuda()
    exp = _oracle_conv(x, w, b, 1, 1, 1.0, add=flow[..., :2])
    for route, expect in ((1, 1), (0, 6)):       # the generic kernel forced; the engine's own choice (r04): pointwise map + gather
        ob.zero_()
        took = _run_conv(xb, 0, cin, w, b, None, ob, 0, 1, h, wd, 1, 1, 1.0, route, add_buf=fb)
        assert took == expect
        err = np.abs(ob.cpu().numpy()[..., :2] - exp).max()
        print(f"dc_conv7 + flow, route {took}: max|err| {err:.2e}")
        assert err < 2e-5, err
        assert not ob.cpu().numpy()[..., 2:].any()
    w2 = (rng.standard_normal((3, 3, cin, 64)) * np.sqrt(2.0 / (9 * cin))).astype(np.float32)
    b2 = (rng.standard_normal(64) * 0.05).astype(np.float32)
    ob2 = torch.zeros((1, h, wd, 64), dtype=torch.float32, device="cuda")
    took = _run_conv(xb, 0, cin, w2, b2, None, ob2, 0, 1, h, wd, 1, 4, 0.1, 1)
    assert took == 1
    err = np.abs(ob2.cpu().numpy() - _oracle_conv(x, w2, b2, 1, 4, 0.1)).max()
    print(f"generic kernel, dilation 4: max|err| {err:.2e}")
    assert err < 2e-5


def test_flow_head_direct_kernel_vs_oracle():
    """The 2-channel flow heads (predict_flow/flow{l}, model_pwcnet.py:1447) on FISRnet's direct kernel: a 565-channel dense-block
    buffer (level 3 layout: 448 + 88 + 64 + 4 + 4 = 608 padded) read whole, linear output into a 4-wide flow buffer."""
    h, wd = 272, 480
    rng = np.random.default_rng(12)
    cin_buf = 608
    chmap = list(range(0, 448)) + list(range(448, 448 + 81)) + list(range(536, 600)) + [600, 601, 604, 605]
    ci = len(chmap)
    x = (rng.standard_normal((1, h, wd, cin_buf)) * 0.3).astype(np.float32)
    w = (rng.standard_normal((3, 3, ci, 2)) * 0.01).astype(np.float32)
    b = (rng.standard_normal(2) * 0.05).astype(np.float32)
    ob = torch.zeros((1, h, wd, 4), dtype=torch.float32, device="cuda")
    xb = torch.from_numpy(x).cuda()
    exp = _oracle_conv(x[..., chmap], w, b, 1, 1, 1.0)
    for route, expect in ((3, 3), (0, 6)):       # the direct kernel forced; the engine's own choice (r04): pointwise map + gather
        ob.zero_()
        took = _run_conv(xb, 0, cin_buf, w, b, chmap, ob, 0, 1, h, wd, 1, 1, 1.0, route)
        assert took == expect
        got = ob.cpu().numpy()
        err = np.abs(got[..., :2] - exp).max()
        print(f"flow head, route {took}: max|err| {err:.2e}")
        assert err < 2e-5 and not got[..., 2:].any()


@pytest.mark.parametrize("shape", [(2, 67, 119), (1, 9, 13), (3, 16, 16)])
def test_pointwise_two_channel_layers_ragged_sizes_vs_oracle(shape):
    """pwc_pointwise_f32_kernel<20> + pwc_conv3_combine_kernel (route 6) on pixel counts that are not multiples of the 256-pixel
    tile, a channel range in the middle of a wider buffer, batch > 1 (the gather must not cross image borders), with and without
    the added flow."""
    n, h, wd = shape
    rng = np.random.default_rng(n * 1000 + h)
    cin_buf, in_co, in_cs = 64, 32, 128
    x = (rng.standard_normal((n, h, wd, in_cs)) * 0.5).astype(np.float32)
    chmap = [c for c in range(cin_buf) if c % 7 != 3]
    w = (rng.standard_normal((3, 3, len(chmap), 2)) * 0.05).astype(np.float32)
    b = (rng.standard_normal(2) * 0.05).astype(np.float32)
    flow = (rng.standard_normal((n, h, wd, 4)) * 2).astype(np.float32)
    xb, fb = torch.from_numpy(x).cuda(), torch.from_numpy(flow).cuda()
    xin = x[..., in_co:in_co + cin_buf][..., chmap]
    for add in (None, fb):
        ob = torch.zeros((n, h, wd, 4), dtype=torch.float32, device="cuda")
        took = _run_conv(xb, in_co, cin_buf, w, b, chmap, ob, 0, n, h, wd, 1, 1, 1.0, 6, add_buf=add)
        assert took == 6
        exp = _oracle_conv(xin, w, b, 1, 1, 1.0, add=None if add is None else flow[..., :2])
        err = np.abs(ob.cpu().numpy()[..., :2] - exp).max()
        assert err < 2e-5, (shape, err)


@pytest.mark.parametrize("shape", [(136, 240), (67, 119)])
def test_deconv_vs_oracle(shape):
    """pwc_deconv_kernel = tf.layers.conv2d_transpose(x, 2, 4, 2, 'same') (model_pwcnet.py:1196) on a padded dense-block buffer
    (up_feat) and on the 4-wide flow buffer (up_flow)."""
    h, wd = shape
    rng = np.random.default_rng(h)
    cin4 = 96
    chmap = list(range(0, 50)) + list(range(56, 56 + 33))
    ci = len(chmap)
    x = (rng.standard_normal((1, h, wd, cin4 + 8)) * 0.5).astype(np.float32)
    w = (rng.standard_normal((4, 4, 2, ci)) * 0.1).astype(np.float32)
    b = (rng.standard_normal(2) * 0.05).astype(np.float32)
    ob = torch.full((1, 2 * h, 2 * wd, 12), -3.0, dtype=torch.float32, device="cuda")
    cm = (ctypes.c_int * ci)(*chmap)
    xb = torch.from_numpy(x).cuda()          # (named: a temporary would be freed, and its block re-used, before the launch)
    flib.check(flib.lib().fisr_pwc_op_deconv(_ptr(xb), 1, cin4 + 8, 4, cin4, w.ctypes.data_as(F32P), b.ctypes.data_as(F32P),
                                             ci, cm, _ptr(ob), 12, 8, 1, h, wd, flib.PREC_F32W, _stream()))
    torch.cuda.synchronize()
    xs = x[..., 4:4 + cin4][..., chmap]
    exp = P.deconv(torch.from_numpy(xs).double().permute(0, 3, 1, 2), {"op/kernel": w, "op/bias": b}, "op").permute(0, 2, 3, 1).numpy()
    got = ob.cpu().numpy()
    err = np.abs(got[..., 8:10] - exp).max()
    print(f"deconv {h}x{wd}: max|err| {err:.2e}")
    assert err < 2e-5 and (got[..., :8] == -3.0).all() and (got[..., 10:] == -3.0).all()


@pytest.mark.parametrize("shape,c", [((136, 240), 96), ((272, 480), 64), ((35, 61), 196), ((67, 119), 128)])
def test_costvol_vs_oracle(shape, c):
    """pwc_costvol_kernel = core_costvol.cost_volume + leaky relu (model_pwcnet.py:1277): 81 displacements, zero outside the image,
    on level-sized maps incl. ragged ones (the 8x32 tiles + 4-pixel halo straddle every border) and the 196-channel top level."""
    h, wd = shape
    rng = np.random.default_rng(h + c)
    c1 = (rng.standard_normal((2, h, wd, c)) * 0.7).astype(np.float32)
    c2 = (rng.standard_normal((2, h, wd, c)) * 0.7).astype(np.float32)
    ob = torch.full((2, h, wd, 96), 9.5, dtype=torch.float32, device="cuda")
    c1b, c2b = torch.from_numpy(c1).cuda(), torch.from_numpy(c2).cuda()
    flib.check(flib.lib().fisr_pwc_op_costvol(_ptr(c1b), _ptr(c2b), c, _ptr(ob), 96, 8,
                                              2, h, wd, flib.PREC_F32W, _stream()))
    torch.cuda.synchronize()
    t = lambda a: torch.from_numpy(a).double().permute(0, 3, 1, 2)
    exp = P.lrelu(P.cost_volume(t(c1), t(c2))).permute(0, 2, 3, 1).numpy()
    got = ob.cpu().numpy()
    err = np.abs(got[..., 8:89] - exp).max()
    print(f"cost volume {h}x{wd}x{c}: max|err| {err:.2e}")
    assert err < 1e-5 and (got[..., :8] == 9.5).all() and (got[..., 89:] == 9.5).all()


@pytest.mark.parametrize("shape", [(136, 240), (67, 119)])
def test_warp_with_flows_leaving_the_image_vs_oracle(shape):
    """pwc_warp_kernel = core_warp.dense_image_warp (model_pwcnet.py:1178): queries far outside every border (clamped), exactly
    on the last row / column, and sub-pixel everywhere else; the flow is read from channels 4, 5 of a wider buffer and scaled
    (20 / 2^lvl, :1560)."""
    h, wd = shape
    rng = np.random.default_rng(wd)
    c = 96
    img = rng.standard_normal((2, h, wd, c)).astype(np.float32)
    flow = (rng.standard_normal((2, h, wd, 8)) * 6).astype(np.float32)
    flow[:, : h // 4, :, 4:6] *= 40                                  # far outside
    flow[0, h // 2, :, 4] = (wd - 1 - np.arange(wd)) / 2.5           # x + scale*u == wd - 1 exactly
    flow[0, :, wd // 2, 5] = (h - 1 - np.arange(h)) / 2.5
    ob = torch.zeros((2, h, wd, c), dtype=torch.float32, device="cuda")
    imb, flb = torch.from_numpy(img).cuda(), torch.from_numpy(flow).cuda()
    flib.check(flib.lib().fisr_pwc_op_warp(_ptr(imb), c, _ptr(flb), 8, 4, 2.5,
                                           _ptr(ob), 2, h, wd, flib.PREC_F32W, _stream()))
    torch.cuda.synchronize()
    f = torch.from_numpy(flow[..., 4:6].astype(np.float32) * np.float32(2.5)).double().permute(0, 3, 1, 2)
    exp = P.dense_image_warp(torch.from_numpy(img).double().permute(0, 3, 1, 2), f).permute(0, 2, 3, 1).numpy()
    err = np.abs(ob.cpu().numpy() - exp).max()
    print(f"warp {h}x{wd}: max|err| {err:.2e}")
    # the query position is computed in fp32 on the GPU (x + u: ~1e-5 px at these magnitudes), times the image gradient
    assert err < 2e-4


@pytest.mark.parametrize("shape,c", [((17, 30), 196), ((34, 61), 128), ((75, 133), 32), ((9, 40), 20)])
def test_costvol_fp16_dot2_kernel_vs_oracle(shape, c):
    """The fp16 engine's cost volume (r05: halo chunk kept in fp16 in LDS, v_dot2_f32_f16 products): channel counts that are not a
    multiple of the 16-channel chunk (196 = 12 x 16 + 4, 20), ragged tiles, an output slot inside a wider buffer whose neighbours stay."""
    h, wd = shape
    rng = np.random.default_rng(h * 1000 + c)
    c1 = _h16(rng.standard_normal((2, h, wd, c)) * 0.7)
    c2 = _h16(rng.standard_normal((2, h, wd, c)) * 0.7)
    c1b, c2b = _dev(c1, "fp16"), _dev(c2, "fp16")
    cv = torch.full((2, h, wd, 96), 9.5, dtype=torch.float16, device="cuda")
    flib.check(flib.lib().fisr_pwc_op_costvol(_ptr(c1b), _ptr(c2b), c, _ptr(cv), 96, 8, 2, h, wd, flib.PREC_F16, _stream()))
    torch.cuda.synchronize()
    t = lambda a: torch.from_numpy(a).double().permute(0, 3, 1, 2)
    exp = P.lrelu(P.cost_volume(t(c1), t(c2))).permute(0, 2, 3, 1).numpy()
    got = _host(cv)
    err = np.abs(got[..., 8:89] - exp).max()
    print(f"fp16 cost volume {shape} x {c}: max|err| {err:.2e}")
    assert err < 1e-3 * max(1.0, np.abs(exp).max())
    assert (got[..., :8] == 9.5).all() and (got[..., 89:] == 9.5).all()


def test_op_entry_errors():
    x = torch.zeros((1, 16, 32, 32), device="cuda")
    o = torch.zeros((1, 16, 32, 64), device="cuda")
    w = np.zeros((3, 3, 32, 64), np.float32)
    b = np.zeros(64, np.float32)
    L = flib.lib()
    call = lambda in_co, stride, route, prec: L.fisr_pwc_op_conv(_ptr(x), 32, in_co, 32, w.ctypes.data_as(F32P), b.ctypes.data_as(F32P), 32, 64, None,
                                                                  _ptr(o), 1, 64, 0, None, 0, 0, 1, 16, 32, stride, 1, 0.1, route, prec, _stream())
    assert call(0, 3, 0, flib.PREC_F32W) < 0                      # stride 3
    assert call(0, 2, 2, flib.PREC_F32W) < 0                      # Winograd forced on a stride-2 layer
    assert call(2, 1, 0, flib.PREC_F32W) < 0                      # unaligned channel offset
    assert call(0, 1, 4, flib.PREC_F32W) < 0                      # the fp16 kernel asked of the fp32 engine
    assert call(0, 1, 0, flib.PREC_BF16X3) < 0                    # not a precision of the flow network


# ----------------------------------------------------------------------------- the fp16 flow engine (FISR_PREC_F16, cfg5)
def _h16(a):
    return np.asarray(a, np.float32).astype(np.float16).astype(np.float32)


@pytest.mark.parametrize("cout", [96, 128])
@pytest.mark.parametrize("dil,shape", [(1, (272, 480)), (2, (272, 480)), (4, (136, 240)), (8, (136, 240)), (16, (136, 240)), (16, (272, 480)),
                                       (4, (75, 133)), (8, (139, 251))])
def test_lds_dma_general_dilated_fp16_vs_oracle(dil, shape, cout):
    """The GENERAL instantiation of the fp16 LDS-DMA kernel (conv3x3_dma.h), which carries the dense and context layers of the fp16
    flow engine: dilation as d x d interleaved sub-images, channel-range input and output of wider buffers, leaky relu, batch 2;
    Cout = 96 runs the 32-channel N blocks (NT = 1: the network's 32 / 96-channel layers), Cout = 128 the 64-channel ones.  Inputs
    and weights are fp16 values, so the float64 oracle sees the same operands: what is left is the fp32 accumulation and ONE
    rounding of the result to fp16."""
    h, wd = shape
    rng = np.random.default_rng(1000 * dil + h + cout)
    in_cs, in_co, cin = 160, 32, 96
    out_cs, out_co = 256, 64
    x = _h16(rng.standard_normal((2, h, wd, in_cs)) * 0.5)
    w = _h16(rng.standard_normal((3, 3, cin, cout)) * np.sqrt(2.0 / (9 * cin)))
    b = (rng.standard_normal(cout) * 0.05).astype(np.float32)
    xb = _dev(x, "fp16")
    ob = torch.full((2, h, wd, out_cs), 7.25, dtype=torch.float16, device="cuda")
    took = _run_conv(xb, in_co, cin, w, b, None, ob, out_co, 2, h, wd, 1, dil, 0.1, 4, prec="fp16")
    assert took == 4
    got = _host(ob)
    exp = _oracle_conv(x[..., in_co:in_co + cin], w, b, 1, dil, 0.1)
    err = np.abs(got[..., out_co:out_co + cout] - exp)
    ulp = np.maximum(np.abs(exp), 2.0 ** -14) * 2.0 ** -10
    print(f"lds-dma GENERAL fp16 dil {dil} {h}x{wd}: max|err| {err.max():.2e} (|y| max {np.abs(exp).max():.2f})")
    assert (err <= 0.51 * ulp + 3e-6 * np.abs(exp).max()).all()
    assert (got[..., :out_co] == 7.25).all() and (got[..., out_co + cout:] == 7.25).all()


@pytest.mark.parametrize("case", [
    # h, w, cin, cout, stride, in_cs, in_co, out_cs, out_co
    (271, 479, 16, 32, 2, 16, 0, 32, 0),        # conv2a's shape on an odd size: Cout 32 of a 64-channel block, TF 'SAME' pads (1, 1)
    (136, 241, 96, 128, 2, 96, 0, 128, 0),      # six chunks
    (19, 31, 196, 196, 1, 196, 0, 196, 0),      # level 6: 196 channels = 12 chunks + 4, pixel records 8-byte aligned only
    (23, 40, 52, 48, 1, 60, 4, 52, 4),          # a channel range of a wider buffer at an 8-byte offset, ragged last chunk, untouched neighbours
    (40, 70, 64, 16, 2, 72, 8, 16, 0),          # 16-byte aligned range of a wider buffer (the aligned instantiation), 16 outputs
])
def test_fp16_generic_kernel_on_the_matrix_pipe_vs_oracle(case):
    """pwc_convg_f16_kernel<A16> (r05): the fp16 engine's stride-2 pyramid convolutions and level 6's 196-channel layers -- fp16 operands
    (inputs and weights exactly representable here), fp32 accumulation: what is left against the float64 oracle is the accumulation
    order and the fp16 rounding of the stored result."""
    h, wd, cin, cout, stride, in_cs, in_co, out_cs, out_co = case
    rng = np.random.default_rng(h * 131 + cin)
    xfull = _h16(rng.standard_normal((2, h, wd, in_cs)) * 0.5)
    w = _h16(rng.standard_normal((3, 3, cin, cout)) * np.sqrt(2.0 / (9 * cin)))
    b = (rng.standard_normal(cout) * 0.05).astype(np.float32)
    oh, ow = -(-h // stride), -(-wd // stride)
    xb = _dev(xfull, "fp16")
    ob = torch.full((2, oh, ow, out_cs), 7.25, dtype=torch.float16, device="cuda")
    took = _run_conv(xb, in_co, cin, w, b, None, ob, out_co, 2, h, wd, stride, 1, 0.1, 1, prec="fp16")
    assert took == 1
    exp = _oracle_conv(xfull[..., in_co:in_co + cin], w, b, stride, 1, 0.1)
    got = _host(ob)
    err = np.abs(got[..., out_co:out_co + cout] - exp).max()
    print(f"fp16 generic kernel {case}: max|err| {err:.2e}")
    assert err < 1.5e-3 * max(1.0, np.abs(exp).max())
    assert (got[..., :out_co] == 7.25).all() and (got[..., out_co + cout:] == 7.25).all()


def test_fp16_engine_routes_and_small_kernels_vs_oracle():
    """The other layer types of the fp16 flow engine on level-sized maps: stride-2 pyramid conv (generic kernel, fp16 in / out),
    the 2-channel flow head and dc_conv7 (FISRnet's 16-row fp16 kernel, float32 out, + the float32 flow),
    transpose conv from the float32 flow and from fp16 features, cost volume, warp."""
    rng = np.random.default_rng(77)
    h, wd = 136, 240
    # stride 2
    x = _h16(rng.standard_normal((2, h, wd, 32)) * 0.5)
    w = _h16(rng.standard_normal((3, 3, 32, 64)) * 0.06)
    b = (rng.standard_normal(64) * 0.05).astype(np.float32)
    xb = _dev(x, "fp16")
    ob = torch.zeros((2, h // 2, wd // 2, 64), dtype=torch.float16, device="cuda")
    assert _run_conv(xb, 0, 32, w, b, None, ob, 0, 2, h, wd, 2, 1, 0.1, 0, prec="fp16") == 1
    exp = _oracle_conv(x, w, b, 2, 1, 0.1)
    assert np.abs(_host(ob) - exp).max() < 2e-3 * max(1.0, np.abs(exp).max())
    # flow head: 608-channel buffer -> 2 channels, float32
    cin_buf = 608
    chmap = list(range(0, 448)) + list(range(448, 448 + 81)) + list(range(536, 600)) + [600, 601, 604, 605]
    xd = _h16(rng.standard_normal((2, h, wd, cin_buf)) * 0.3)
    wh = _h16(rng.standard_normal((3, 3, len(chmap), 2)) * 0.01)
    bh = (rng.standard_normal(2) * 0.05).astype(np.float32)
    xdb = _dev(xd, "fp16")
    fl = torch.zeros((2, h, wd, 4), dtype=torch.float32, device="cuda")
    assert _run_conv(xdb, 0, cin_buf, wh, bh, chmap, fl, 0, 2, h, wd, 1, 1, 1.0, 0, prec="fp16") == 3
    exph = _oracle_conv(xd[..., chmap], wh, bh, 1, 1, 1.0)
    got = _host(fl)
    print(f"fp16 flow head: max|err| {np.abs(got[..., :2] - exph).max():.2e}")
    assert np.abs(got[..., :2] - exph).max() < 2e-5 and not got[..., 2:].any()     # fp32 accumulation of exact fp16 products
    # dc_conv7 + flow
    x7 = _h16(rng.standard_normal((2, h, wd, 32)) * 0.5)
    w7 = _h16(rng.standard_normal((3, 3, 32, 2)) * 0.05)
    x7b = _dev(x7, "fp16")
    ref = torch.zeros((2, h, wd, 4), dtype=torch.float32, device="cuda")
    assert _run_conv(x7b, 0, 32, w7, bh, None, ref, 0, 2, h, wd, 1, 1, 1.0, 0, add_buf=fl, prec="fp16") == 3
    exp7 = _oracle_conv(x7, w7, bh, 1, 1, 1.0, add=got[..., :2])
    assert np.abs(_host(ref)[..., :2] - exp7).max() < 2e-5
    # transpose convs
    wdv = (rng.standard_normal((4, 4, 2, 2)) * 0.1).astype(np.float32)
    up = torch.full((2, 2 * h, 2 * wd, 8), -3.0, dtype=torch.float16, device="cuda")
    flib.check(flib.lib().fisr_pwc_op_deconv(_ptr(fl), 1, 4, 0, 4, wdv.ctypes.data_as(F32P), bh.ctypes.data_as(F32P), 2, None, _ptr(up), 8, 4,
                                             2, h, wd, flib.PREC_F16, _stream()))
    torch.cuda.synchronize()
    expd = P.deconv(torch.from_numpy(got[..., :2]).double().permute(0, 3, 1, 2), {"op/kernel": wdv, "op/bias": bh}, "op").permute(0, 2, 3, 1).numpy()
    gu = _host(up)
    assert np.abs(gu[..., 4:6] - expd).max() < 1e-3 * max(1.0, np.abs(expd).max()) and (gu[..., :4] == -3.0).all() and (gu[..., 6:] == -3.0).all()
    wdf = (rng.standard_normal((4, 4, 2, 64)) * 0.05).astype(np.float32)
    upf = torch.zeros((2, 2 * h, 2 * wd, 4), dtype=torch.float16, device="cuda")
    flib.check(flib.lib().fisr_pwc_op_deconv(_ptr(xb), 0, 32, 0, 32, wdf[..., :32].copy().ctypes.data_as(F32P),
                                             bh.ctypes.data_as(F32P), 32, None, _ptr(upf), 4, 0, 2, h, wd, flib.PREC_F16, _stream()))
    torch.cuda.synchronize()
    expf = P.deconv(torch.from_numpy(x).double().permute(0, 3, 1, 2), {"op/kernel": wdf[..., :32].copy(), "op/bias": bh}, "op").permute(0, 2, 3, 1).numpy()
    assert np.abs(_host(upf)[..., :2] - expf).max() < 1e-3 * max(1.0, np.abs(expf).max())
    # cost volume and warp
    c1 = _h16(rng.standard_normal((2, h, wd, 96)) * 0.7)
    c2 = _h16(rng.standard_normal((2, h, wd, 96)) * 0.7)
    c1b, c2b = _dev(c1, "fp16"), _dev(c2, "fp16")
    cv = torch.full((2, h, wd, 96), 9.5, dtype=torch.float16, device="cuda")
    flib.check(flib.lib().fisr_pwc_op_costvol(_ptr(c1b), _ptr(c2b), 96, _ptr(cv), 96, 8, 2, h, wd, flib.PREC_F16, _stream()))
    torch.cuda.synchronize()
    t = lambda a: torch.from_numpy(a).double().permute(0, 3, 1, 2)
    expc = P.lrelu(P.cost_volume(t(c1), t(c2))).permute(0, 2, 3, 1).numpy()
    gc = _host(cv)
    assert np.abs(gc[..., 8:89] - expc).max() < 1e-3 * max(1.0, np.abs(expc).max()) and (gc[..., :8] == 9.5).all() and (gc[..., 89:] == 9.5).all()
    fw = _h16(rng.standard_normal((2, h, wd, 8)) * 3)
    fwb = _dev(fw, "fp16")
    wo = torch.zeros((2, h, wd, 96), dtype=torch.float16, device="cuda")
    flib.check(flib.lib().fisr_pwc_op_warp(_ptr(c2b), 96, _ptr(fwb), 8, 4, 2.5, _ptr(wo), 2, h, wd, flib.PREC_F16, _stream()))
    torch.cuda.synchronize()
    f = torch.from_numpy(fw[..., 4:6] * np.float32(2.5)).double().permute(0, 3, 1, 2)
    expw = P.dense_image_warp(t(c2), f).permute(0, 2, 3, 1).numpy()
    assert np.abs(_host(wo) - expw).max() < 4e-3
