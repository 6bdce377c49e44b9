"""GPU parity of the on-GPU optical flow (PWC-Net-large, cfg5 of BASELINE.json; run with `-m gpu`): the HIP path through
the `fisr_pwc_*` C-ABI against the CPU oracle (oracle/pwcnet_oracle.py, float64) on seeded synthetic weights.
"Parity unpinned" for the network arithmetic (TensorFlow, core_warp, core_costvol and the checkpoint are absent; see
the oracle's header); the two scikit-image resizes are pinned against scikit-image 0.18.3 fixtures."""
import ctypes
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")

import pwcnet_oracle as P  # noqa: E402
from fisr_amd import lib as flib  # noqa: E402
from fisr_amd import pwcnet  # noqa: E402


@pytest.fixture(scope="module")
def pwc():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    W = pwcnet.synthetic_weights(595000, flow_gain=3.0)
    net = pwcnet.PWCNet("cuda:0")
    net.set_weights(W)
    yield net, W
    net.close()


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def test_variable_inventory_matches_oracle():
    assert list(pwcnet.variable_shapes().items()) == list(P.variable_shapes().items())
    assert len(pwcnet.variable_shapes()) == 182


def test_prep_kernel_vs_oracle(gold_dir):
    """YUV uint8 -> RGB -> x2 scikit-image resize -> uint8 truncation -> /255 -> zero pad to 64 (script :121-131,
    adapt_x model_pwcnet.py:399-411).  The truncation makes a value within rounding of an integer flip by 1/255:
    allowed on at most 1e-4 of the samples."""
    g = np.load(os.path.join(gold_dir, "scene1_crop96.npz"))
    key = [k for k in g.files if g[k].ndim == 4 and g[k].dtype == np.uint8][0]
    yuv = g[key][0]                                           # [96, 96, 3] uint8
    rng = np.random.default_rng(3)
    for frame in (yuv, rng.integers(0, 256, (40, 72, 3), dtype=np.uint8)):
        h, w = frame.shape[:2]
        PH, PW = -(-2 * h // 64) * 64, -(-2 * w // 64) * 64
        out = torch.full((PH, PW, 4), float("nan"), device="cuda")
        flib.check(flib.lib().fisr_pwc_prep(ctypes.c_void_p(torch.from_numpy(frame).cuda().data_ptr()), h, w,
                                            ctypes.c_void_p(out.data_ptr()), PH, PW, _stream()))
        torch.cuda.synchronize()
        got = out.cpu().numpy()
        exp = np.zeros((PH, PW, 4), np.float32)
        exp[:2 * h, :2 * w, :3] = np.array(P.resize_up2_skimage(P.yuv2rgb(frame.astype(np.float32))), dtype=np.uint8).astype(np.float32) / 255.0
        d = np.abs(got - exp)
        assert not np.isnan(got).any()
        assert d.max() <= 1.0 / 255 + 1e-7 and (d > 1e-7).mean() <= 1e-4, (d.max(), (d > 1e-7).mean())
        assert not got[2 * h:].any() and not got[:, 2 * w:].any() and not got[..., 3].any()


def test_flow_out_kernel_vs_oracle():
    """x4 legacy bilinear * 4 (model_pwcnet.py:1587-1590), crop to 2h x 2w, scikit-image anti-aliased /2 resize, / 2."""
    rng = np.random.default_rng(4)
    for (h, w) in ((40, 72), (96, 160)):
        FH, FW = -(-2 * h // 64) * 16, -(-2 * w // 64) * 16
        f2 = (rng.standard_normal((FH, FW, 2)) * 3).astype(np.float32)
        out = torch.empty((h, w, 2), device="cuda")
        flib.check(flib.lib().fisr_pwc_flow_out(ctypes.c_void_p(torch.from_numpy(f2).cuda().data_ptr()), FH, FW,
                                                ctypes.c_void_p(out.data_ptr()), h, w, _stream()))
        torch.cuda.synchronize()
        up = P.resize_bilinear_legacy(torch.from_numpy(f2).double().permute(2, 0, 1)[None], 4) * 4
        up = up[0].permute(1, 2, 0).numpy()[None, :2 * h, :2 * w]
        exp = P.resize_down2_skimage_aa(up)[0] / 2.0
        assert np.abs(out.cpu().numpy() - exp).max() < 2e-5


@pytest.mark.parametrize("shape", [(64, 128), (128, 192)])
def test_pwcnet_nn_vs_oracle(pwc, shape):
    """The network alone, both directions, every pyramid level and the final x4 flow against the float64 oracle."""
    net, W = pwc
    H, Wd = shape
    rng = np.random.default_rng(H)
    base = rng.random((H // 8 + 2, Wd // 8 + 2, 3))
    img = np.kron(base, np.ones((8, 8, 1)))                   # blocky texture ...
    a = img[4:4 + H, 4:4 + Wd] * 0.8 + rng.random((H, Wd, 3)) * 0.2
    b = img[2:2 + H, 7:7 + Wd] * 0.8 + rng.random((H, Wd, 3)) * 0.2      # ... shifted: a real displacement to find
    im = np.zeros((2, H, Wd, 4), np.float32)
    im[0, ..., :3], im[1, ..., :3] = a, b
    pairs = np.stack([np.stack([im[0, ..., :3], im[1, ..., :3]]), np.stack([im[1, ..., :3], im[0, ..., :3]])])
    taps = {}
    ref_pred, ref_pyr = P.nn(torch.from_numpy(pairs).double(), W, taps)
    got_pred, got_pyr = net.nn(torch.from_numpy(im).cuda(), want_pyramid=True)
    torch.cuda.synchronize()
    for d in range(2):
        for k, lvl in enumerate(range(6, 1, -1)):
            exp = ref_pyr[k][d].permute(1, 2, 0).numpy()
            got = got_pyr[d][k].cpu().numpy()
            err = np.abs(got - exp).max()
            print(f"dir {d} flow{lvl}: max|err| {err:.2e}  (|flow| max {np.abs(exp).max():.3f})")
            assert err < 3e-4, (d, lvl, err)
        err = np.abs(got_pred[d].cpu().numpy() - ref_pred[d].numpy()).max()
        print(f"dir {d} flow_pred: max|err| {err:.2e} (|flow| max {np.abs(ref_pred[d].numpy()).max():.3f})")
        assert err < 1.5e-3


def test_flow_pair_end_to_end_vs_oracle(pwc, gold_dir):
    """One iteration of the reference script's loop on two real LR frames (scene1 crop): YUV uint8 in, LR-pixel flows
    in both directions out."""
    net, W = pwc
    g = np.load(os.path.join(gold_dir, "scene1_crop96.npz"))
    key = [k for k in g.files if g[k].ndim == 4 and g[k].dtype == np.uint8][0]
    fa, fb = g[key][0], g[key][1]
    exp = P.compute_flow_pair(fa, fb, W)
    ab, ba = net.flow_pair(torch.from_numpy(fa), torch.from_numpy(fb))
    torch.cuda.synchronize()
    for name, got, e in (("a->b", ab, exp[0]), ("b->a", ba, exp[1])):
        err = np.abs(got.cpu().numpy() - e).max()
        print(f"{name}: max|err| {err:.2e}, |flow| max {np.abs(e).max():.3f}")
        # a uint8 flip in the pre-processing (see test_prep_kernel_vs_oracle) moves the flow by ~1e-3 at most
        assert err < 5e-3
    flows = net.compute_flow([torch.from_numpy(g[key][k]) for k in range(3)])
    assert tuple(flows.shape) == (2, 2, 96, 96, 2) and torch.isfinite(flows).all()
    assert torch.equal(flows[0, 0], ab) and torch.equal(flows[0, 1], ba)


def test_pwc_errors(pwc):
    net, W = pwc
    with pytest.raises(ValueError):
        net.nn(torch.zeros((2, 64, 64, 3)))
    with pytest.raises(Exception):
        net.nn(torch.zeros((2, 96, 64, 4)))              # not a multiple of 64
    bad = dict(W)
    bad.pop("pwcnet/ctxt/dc_conv27/bias")
    n2 = pwcnet.PWCNet("cuda:0")
    with pytest.raises(KeyError):
        n2.set_weights(bad)
    n2.close()


def test_flow_pair_full_size_vs_oracle_golden(pwc, gold_dir):
    """cfg5 at the size the bench times it: one 1080x1920 frame pair -> x2 up-scaled 2176x3840 network input, levels of
    544x960 .. 34x60 (every dilated context layer runs on sub-images of many 8x32 tiles), both directions, against the float64
    oracle's flows committed on every 8th LR pixel (tests/golden/pwc_flow_1080p_sparse.npz, made once in the build container
    by oracle/make_golden_pwc_fullsize.py) and against its refined level-6 .. level-2 flows of direction a->b."""
    import zlib
    from tests_support import make_flow_frames
    path = os.path.join(gold_dir, "pwc_flow_1080p_sparse.npz")
    if not os.path.isfile(path):
        pytest.fail("tests/golden/pwc_flow_1080p_sparse.npz missing: run oracle/make_golden_pwc_fullsize.py")
    g = np.load(path)
    fa, fb = make_flow_frames(int(g["seed"]), 1080, 1920)
    assert zlib.crc32(fa.tobytes() + fb.tobytes()) == int(g["frames_crc"]), "the frame generator drifted from the golden's"
    assert float(g["flow_gain"]) == 3.0                      # the fixture's weight set
    net, W = pwc
    ab, ba = net.flow_pair(torch.from_numpy(fa), torch.from_numpy(fb))
    torch.cuda.synchronize()
    st = int(g["stride"])
    for name, got, exp in (("a->b", ab, g["flow_sparse"][0]), ("b->a", ba, g["flow_sparse"][1])):
        d = np.abs(got[::st, ::st].cpu().numpy().astype(np.float64) - exp)
        print(f"1080p {name}: max|err| {d.max():.2e} rms {np.sqrt((d ** 2).mean()):.2e} px, |flow| max {np.abs(exp).max():.2f} px")
        # a uint8 flip of the pre-processing (value within fp rounding of an integer) moves the flow locally by ~1e-3 px
        assert d.max() < 5e-3 and np.sqrt((d ** 2).mean()) < 2e-4
    # the pyramid of direction a->b through the same pre-processing kernel
    H, Wd = 2176, 3840
    im = torch.empty((2, H, Wd, 4), device="cuda")
    for k, f in enumerate((fa, fb)):
        flib.check(flib.lib().fisr_pwc_prep(ctypes.c_void_p(torch.from_numpy(f).cuda().data_ptr()), 1080, 1920,
                                            ctypes.c_void_p(im[k].data_ptr()), H, Wd, _stream()))
    _, pyr = net.nn(im, want_pyramid=True)
    torch.cuda.synchronize()
    for k, lvl in enumerate(range(6, 1, -1)):
        s2 = 2 if k > 2 else 1
        exp = g[f"flow{lvl}_ab_sparse"]
        got = pyr[0][k][::s2, ::s2].cpu().numpy()
        err = np.abs(got - exp).max()
        print(f"1080p flow{lvl} ({pyr[0][k].shape[0]}x{pyr[0][k].shape[1]}): max|err| {err:.2e}, |flow| max {np.abs(exp).max():.3f}")
        assert err < 2e-3, (lvl, err)


# ----------------------------------------------------------------------------- the 16-bit flow engine (cfg5: FISR_PREC_F16)
@pytest.fixture(scope="module")
def pwc16():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    W = pwcnet.synthetic_weights(595000, flow_gain=3.0)
    net = pwcnet.PWCNet("cuda:0", precision="fp16")
    net.set_weights(W)
    yield net, W
    net.close()


def test_flow_stack_equals_the_pairwise_loop(pwc, pwc16, gold_dir):
    """fisr_pwc_flow_stack (one pyramid per frame, the directions batched through the decoder) against the reference script's
    pair-by-pair loop (flow_pair): the same kernels on the same data in another batch arrangement -> identical flows, in both
    engines; 6 frames = 10 directions = two decoder batches."""
    g = np.load(os.path.join(gold_dir, "scene1_crop96.npz"))
    key = [k for k in g.files if g[k].ndim == 4 and g[k].dtype == np.uint8][0]
    frames = [torch.from_numpy(g[key][k]).cuda() for k in range(5)]
    frames.append(torch.from_numpy(np.ascontiguousarray(g[key][2][::-1])).cuda())
    for net, _ in (pwc, pwc16):
        st = net.flow_stack(frames)
        torch.cuda.synchronize()
        assert tuple(st.shape) == (5, 2, 96, 96, 2)
        for fr in range(5):
            ab, ba = net.flow_pair(frames[fr], frames[fr + 1])
            assert torch.equal(st[fr, 0], ab) and torch.equal(st[fr, 1], ba), (net.precision, fr)
        assert torch.equal(net.compute_flow(frames, chunk=3), st)


@pytest.mark.parametrize("shape", [(128, 192)])
def test_pwcnet_nn_fp16_vs_oracle(pwc16, shape):
    """The fp16 flow engine against the float64 oracle: fp16 feature tensors (11 significant bits) through a 6-level coarse-to-fine
    cascade -- the flow error is what cfg5 trades for speed; it must stay far below what moves the network's input (flows enter
    FISRnet as flow / 192, FISRnet.py:835: 0.02 px = 1e-4 of the input range)."""
    net, W = pwc16
    H, Wd = shape
    rng = np.random.default_rng(H)
    base = rng.random((H // 8 + 2, Wd // 8 + 2, 3))
    img = np.kron(base, np.ones((8, 8, 1)))
    a = img[4:4 + H, 4:4 + Wd] * 0.8 + rng.random((H, Wd, 3)) * 0.2
    b = img[2:2 + H, 7:7 + Wd] * 0.8 + rng.random((H, Wd, 3)) * 0.2
    im = np.zeros((2, H, Wd, 4), np.float32)
    im[0, ..., :3], im[1, ..., :3] = a, b
    pairs = np.stack([np.stack([im[0, ..., :3], im[1, ..., :3]]), np.stack([im[1, ..., :3], im[0, ..., :3]])])
    ref_pred, ref_pyr = P.nn(torch.from_numpy(pairs).double(), W)
    got_pred, got_pyr = net.nn(torch.from_numpy(im).cuda(), want_pyramid=True)
    torch.cuda.synchronize()
    for d in range(2):
        for k, lvl in enumerate(range(6, 1, -1)):
            err = np.abs(got_pyr[d][k].cpu().numpy() - ref_pyr[k][d].permute(1, 2, 0).numpy())
            print(f"fp16 dir {d} flow{lvl}: max|err| {err.max():.2e} rms {np.sqrt((err ** 2).mean()):.2e}")
        err = np.abs(got_pred[d].cpu().numpy() - ref_pred[d].numpy())
        print(f"fp16 dir {d} flow_pred (x2-frame pixels): max|err| {err.max():.2e} rms {np.sqrt((err ** 2).mean()):.2e} (|flow| max {np.abs(ref_pred[d].numpy()).max():.3f})")
        assert err.max() < 0.08 and np.sqrt((err ** 2).mean()) < 0.01


def test_flow_pair_full_size_fp16_vs_oracle_golden(pwc16, gold_dir):
    """cfg5's flow at benchmark size in the 16-bit engine against the float64 oracle's golden (every 8th LR pixel of a 1080p pair)."""
    from tests_support import make_flow_frames
    g = np.load(os.path.join(gold_dir, "pwc_flow_1080p_sparse.npz"))
    fa, fb = make_flow_frames(int(g["seed"]), 1080, 1920)
    net, W = pwc16
    fl = net.flow_stack([torch.from_numpy(fa), torch.from_numpy(fb)])
    torch.cuda.synchronize()
    st = int(g["stride"])
    for k, name in enumerate(("a->b", "b->a")):
        d = np.abs(fl[0, k, ::st, ::st].cpu().numpy().astype(np.float64) - g["flow_sparse"][k])
        print(f"1080p fp16 {name}: max|err| {d.max():.2e} rms {np.sqrt((d ** 2).mean()):.2e} LR px, |flow| max {np.abs(g['flow_sparse'][k]).max():.2f} px")
        assert d.max() < 0.05 and np.sqrt((d ** 2).mean()) < 0.005
