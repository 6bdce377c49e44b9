"""CPU tests of the oracle: known answers for every op semantic, cross-checks
against torch-CPU and between the numpy and C restatements, and the fixtures captured from the
reference's own numpy helpers (tests/golden/ref_utils.npz, made by oracle/make_golden.py)."""
import os

import numpy as np
import pytest

import c_oracle as C
import fisr_oracle as O


# ------------------------------------------------------------- op known answers
def test_bicubic_down_is_strided_subsample():  # App. B.2
    x = np.arange(16, dtype=np.float64).reshape(1, 1, 16, 1)
    assert O.resize_bicubic_down(np.tile(x, (1, 4, 1, 1)), 4)[0, 0, :, 0].tolist() == [0, 4, 8, 12]


def test_bilinear_x2_legacy():  # App. B.3
    x = np.array([0., 2., 4.]).reshape(1, 1, 3, 1)
    assert O.resize_bilinear_x2(x)[0, 0, :, 0].tolist() == [0, 1, 2, 3, 4, 4]
    assert O.resize_bilinear_x2(x)[0, 1, :, 0].tolist() == [0, 1, 2, 3, 4, 4]  # last row replicates
    y = np.array([[1., 3.], [5., 7.]]).reshape(1, 2, 2, 1)
    out = O.resize_bilinear_x2(y)[0, :, :, 0]
    assert out.tolist() == [[1, 2, 3, 3], [3, 4, 5, 5], [5, 6, 7, 7], [5, 6, 7, 7]]


def test_depth_to_space_dcr():  # App. B.5
    x = np.arange(256, dtype=np.float64).reshape(1, 1, 1, 256)
    o = O.depth_to_space2(x)
    assert o.shape == (1, 2, 2, 64)
    assert o[0, 0, 0].tolist() == list(range(0, 64))
    assert o[0, 0, 1].tolist() == list(range(64, 128))
    assert o[0, 1, 0].tolist() == list(range(128, 192))
    assert o[0, 1, 1].tolist() == list(range(192, 256))


def test_max_pool():
    x = np.arange(16, dtype=np.float64).reshape(1, 4, 4, 1)
    assert O.max_pool2(x)[0, :, :, 0].tolist() == [[5, 7], [13, 15]]


def test_conv_same_is_cross_correlation():  # App. B.1
    x = np.zeros((1, 3, 3, 1)); x[0, 1, 1, 0] = 1.0
    w = np.arange(9, dtype=np.float32).reshape(3, 3, 1, 1)
    y = O.conv2d(x, w, np.zeros(1, np.float32))[0, :, :, 0]
    # impulse response of a cross-correlation is the flipped kernel
    assert y.tolist() == [[8, 7, 6], [5, 4, 3], [2, 1, 0]]


def test_conv_and_pool_vs_torch():
    torch = pytest.importorskip("torch")
    rng = np.random.default_rng(0)
    x = rng.standard_normal((2, 9, 11, 5))
    w = rng.standard_normal((3, 3, 5, 7)).astype(np.float32)
    b = rng.standard_normal(7).astype(np.float32)
    ref = torch.nn.functional.conv2d(torch.from_numpy(x).permute(0, 3, 1, 2),
                                     torch.from_numpy(w.astype(np.float64)).permute(3, 2, 0, 1),
                                     torch.from_numpy(b.astype(np.float64)), padding=1).permute(0, 2, 3, 1).numpy()
    assert np.abs(O.conv2d(x, w, b) - ref).max() < 1e-12
    assert np.abs(C.conv3x3(x, w, b) - ref).max() < 1e-12
    assert np.abs(C.conv3x3(x, w, b, relu_in=True) - O.conv2d(O.relu(x), w, b)).max() < 1e-12
    x2 = rng.standard_normal((1, 8, 6, 3))
    refp = torch.nn.functional.max_pool2d(torch.from_numpy(x2).permute(0, 3, 1, 2), 2).permute(0, 2, 3, 1).numpy()
    assert np.array_equal(O.max_pool2(x2), refp)


def test_c_oracle_matches_numpy_twin(syn_weights, syn_blob):
    rng = np.random.default_rng(3)
    x = rng.random((1, 32, 32, 29)).astype(np.float32)
    a = O.model(x, syn_weights)
    b = C.forward(x, syn_blob, double=True)
    c = C.forward(x, syn_blob, double=False)
    for u, v, s in zip(a, b, c):
        assert np.abs(u - v).max() < 1e-12
        assert np.abs(u - s).max() < 5e-5


def test_golden_model_fixtures(gold_dir, syn_blob):
    g = np.load(os.path.join(gold_dir, "model_32x64.npz"))
    l1, l2, l3 = C.forward(g["x"], syn_blob, double=True)
    assert np.abs(l1 - g["l1"]).max() < 1e-12
    assert np.abs(l2 - g["l2"]).max() < 1e-12
    assert np.abs(l3 - g["l3"]).max() < 1e-12
    g96 = np.load(os.path.join(gold_dir, "model_96.npz"))
    out = C.forward(g96["inp"][:1], syn_blob, double=True)[2]
    assert np.abs(out[0] - g96["l3"][0]).max() < 1e-6


# ------------------------------------------------------------- pinned to the reference's helpers
@pytest.fixture(scope="module")
def ref(gold_dir):
    return np.load(os.path.join(gold_dir, "ref_utils.npz"))


def test_tiling_vs_reference(ref):
    for row in ref["tiling"]:
        h, w, nh, nw, p, hl, hh, wl, wh, ah, aw, th, tw = (int(v) for v in row)
        pH, pW = p // nw, p % nw
        assert O.get_hw_boundary(32, h, w, pH, h // nh, pW, w // nw) == (hl, hh, wl, wh, ah, aw)
        t = np.zeros((1, (hh - hl) * 2, (wh - wl) * 2, 1), np.int8)
        assert O.trim_patch_boundary(t, 32, h, w, pH, h // nh, pW, w // nw, 2).shape[1:3] == (th, tw)
    # SURVEY 8c known answers for the default 1080p config
    assert O.get_hw_boundary(32, 1024, 1920, 1, 512, 1, 960) == (480, 1024, 928, 1920, 32, 32)
    for p in range(4):
        b = O.get_hw_boundary(32, 128, 192, p // 2, 64, p % 2, 96)
        th, tw = (b[1] - b[0]) * 2, (b[3] - b[2]) * 2
        t = np.arange(th * tw, dtype=np.int64).reshape(1, th, tw, 1)
        assert np.array_equal(O.trim_patch_boundary(t, 32, 128, 192, p // 2, 64, p % 2, 96, 2), ref[f"trim_{p}"])


def test_seq_dim_vs_reference(ref):
    assert np.array_equal(O.merge_seq_dim(ref["seq_in"]), ref["merge_seq"])
    assert np.array_equal(O.split_seq_dim(ref["merge_seq"]), ref["split_seq"])
    m = O.merge_seq_dim(np.arange(2 * 3 * 2 * 2 * 3).reshape(2, 3, 2, 2, 3))
    assert m[0, 0, 0].tolist() == [0, 1, 2, 12, 13, 14, 24, 25, 26]


def test_colour_vs_reference(ref):
    assert np.array_equal(O.yuv2rgb_matlab(ref["yuv_u8"]), ref["yuv2rgb_matlab"])
    assert np.array_equal(O.yuv2rgb_matlab(ref["yuv_f32"]), ref["warp_yuv2rgb"])
    assert np.array_equal(O.rgb2yuv(ref["rgb_f64"]), ref["warp_rgb2yuv"])
    k = O.yuv2rgb_matlab(np.array([[[128, 128, 128], [16, 128, 128], [235, 128, 128]]], np.uint8))
    assert abs(k[0, 0, 0] - 130.4109576) < 1e-6 and k[0, 1, 0] == 0 and abs(k[0, 2, 0] - 254.99999745) < 1e-6


def test_psnr_vs_reference(ref):
    assert O.compute_psnr(ref["psnr_a"], ref["psnr_b"], 1.0) == float(ref["psnr"])
    assert abs(O.compute_psnr(np.zeros(4), np.full(4, 0.1), 1.0) - 20.0) < 1e-12


def test_flo_vs_reference(ref, tmp_path):
    p = str(tmp_path / "t.flo")
    O.write_flo5(p, ref["flo_in"])
    with open(p, "rb") as f:
        assert np.array_equal(np.frombuffer(f.read(), np.uint8), ref["flo_bytes"])
    assert np.array_equal(O.read_flo5(p), ref["flo_read_by_ref"])
    with open(p, "r+b") as f:
        f.write(b"\0\0\0\0")
    with pytest.raises(ValueError):
        O.read_flo5(p)


# ------------------------------------------------------------- warp restatement
def test_remap_identity_and_border():
    rng = np.random.default_rng(1)
    src = rng.random((5, 7, 3)) * 255
    gx, gy = np.meshgrid(np.arange(7, dtype=np.float32), np.arange(5, dtype=np.float32))
    assert np.array_equal(O.remap_linear_replicate(src, gx, gy), src)
    # half-pixel shift = average of neighbours; beyond the border replicates the edge
    out = O.remap_linear_replicate(src, gx + np.float32(0.5), gy)
    assert np.allclose(out[:, :-1], 0.5 * (src[:, :-1] + src[:, 1:]))
    assert np.allclose(out[:, -1], src[:, -1])
    far = O.remap_linear_replicate(src, gx - 100, gy + 100)
    assert np.allclose(far, np.broadcast_to(src[-1:, :1], src.shape))
    # 1/32-px quantisation: a shift of 1/64 rounds half-to-even to 0
    q = O.remap_linear_replicate(src, gx + np.float32(1 / 64), gy)
    assert np.array_equal(q, src)


def test_warp_fixture_reproduces(gold_dir):
    g = np.load(os.path.join(gold_dir, "scene1_crop96.npz"))
    w0 = O.warp_frame(g["frames"][1], g["flows"][0, 0])
    assert np.array_equal(w0, g["warps"][0, 0])
    assert w0.dtype == np.float32 and w0.min() >= 0 and w0.max() <= 255


def test_ssim_identity():
    rng = np.random.default_rng(2)
    a = rng.integers(0, 256, (21, 28, 3)).astype(np.uint8)
    assert abs(O.ssim_pil(a, a) - 1.0) < 1e-12
    assert O.ssim_pil(a, 255 - a) < 0.5


def test_depth_to_space_vs_torch_pixel_shuffle():
    """tf.depth_to_space is 'DCR' (block index major), torch.pixel_shuffle is 'CRD' (channel major):
    permuting channels c*4+(2i+j) <- (2i+j)*C+c must make them agree (independent index-math check)."""
    torch = pytest.importorskip("torch")
    rng = np.random.default_rng(5)
    C = 6
    x = rng.standard_normal((2, 3, 5, 4 * C))
    mine = O.depth_to_space2(x)
    perm = np.array([(b * C + c) for c in range(C) for b in range(4)])          # CRD position -> DCR source
    t = torch.from_numpy(x[..., perm]).permute(0, 3, 1, 2)
    ref = torch.nn.functional.pixel_shuffle(t, 2).permute(0, 2, 3, 1).numpy()
    assert np.array_equal(mine, ref)


def test_bilinear_x2_vs_torch_align_corners_grid():
    """The TF-1.x legacy kernel samples in = out*0.5 (no half-pixel offset).  torch's grid_sample with an
    explicit grid at those coordinates (border-clamped) is an independent bilinear implementation."""
    torch = pytest.importorskip("torch")
    rng = np.random.default_rng(6)
    x = rng.standard_normal((1, 5, 7, 3))
    mine = O.resize_bilinear_x2(x)
    h, w = 5, 7
    ys = np.minimum(np.arange(2 * h) * 0.5, h - 1)
    xs = np.minimum(np.arange(2 * w) * 0.5, w - 1)
    gy, gx = np.meshgrid(ys, xs, indexing="ij")
    grid = np.stack([gx / (w - 1) * 2 - 1, gy / (h - 1) * 2 - 1], -1)[None]
    ref = torch.nn.functional.grid_sample(torch.from_numpy(x).permute(0, 3, 1, 2), torch.from_numpy(grid),
                                          mode="bilinear", padding_mode="border", align_corners=True)
    assert np.abs(mine - ref.permute(0, 2, 3, 1).numpy()).max() < 1e-12


def test_remap_exact_mode_vs_scipy_map_coordinates():
    """Exact-coordinate mode of the remap restatement vs scipy's independent bilinear sampler with
    edge replication; the default (cv2) mode differs only by the 1/32-px coordinate quantisation."""
    ndi = pytest.importorskip("scipy.ndimage")
    rng = np.random.default_rng(9)
    src = rng.random((11, 13, 3)) * 255
    mapx = (rng.random((11, 13)) * 16 - 2).astype(np.float32)
    mapy = (rng.random((11, 13)) * 14 - 2).astype(np.float32)
    mine = O.remap_linear_replicate(src, mapx, mapy, quantized=False)
    ref = np.stack([ndi.map_coordinates(src[..., c], [mapy.astype(np.float64), mapx.astype(np.float64)], order=1, mode="nearest")
                    for c in range(3)], -1)
    assert np.abs(mine - ref).max() < 1e-4      # float32 bilinear weights (as cv2) vs scipy's float64
    q = O.remap_linear_replicate(src, mapx, mapy, quantized=True)
    # quantising coordinates to 1/32 px moves a sample by at most 1/64 px in x and y
    assert np.abs(q - mine).max() <= 255 * (1 / 64) * 2 + 1e-9


def test_torch_cpu_twin_matches_numpy_oracle(syn_weights):
    """oracle/torch_cpu.py (the oneDNN CPU baseline of bench.py) is the same graph as the numpy oracle:
    float64 agreement to rounding on all three levels (conv/pool through torch's CPU kernels, the TF-specific
    resize / depth_to_space / strided sub-sampling restated by hand in both)."""
    torch = pytest.importorskip("torch")
    import torch_cpu as T
    x = np.random.default_rng(11).random((2, 32, 64, 29)).astype(np.float32)
    x[..., 9:17] = (x[..., 9:17] - 0.5) * 0.4
    ref = O.model(x, syn_weights)
    got = T.forward(x, T.prepare_weights(syn_weights, torch.float64))
    for a, b in zip(got, ref):
        assert a.shape == b.shape
        assert np.abs(a - b).max() < 1e-12


def test_pwc_oracle_resizes_pinned_to_scikit_image(gold_dir):
    """The two scikit-image resizes of the reference's flow script (FISR_for_video_pwcnet_predict_from_img_test.py
    :129-130, :139) as restated in oracle/pwcnet_oracle.py against outputs of a real scikit-image 0.18.3
    (oracle/make_golden_pwc.py, run with the anaconda interpreter of the build image)."""
    pytest.importorskip("torch")
    import pwcnet_oracle as P
    g = np.load(os.path.join(gold_dir, "pwc_resize.npz"))
    assert np.abs(P.resize_up2_skimage(g["rgb"]) - g["up"]).max() < 1e-9
    assert np.abs(P.resize_down2_skimage_aa(g["flow"]) - g["down"]).max() < 2e-6     # skimage returns float32 here
    # the script's YUV2RGB is the warp script's (pinned to the imported reference function in ref_utils.npz)
    yuv = np.random.default_rng(2).integers(0, 256, (7, 9, 3)).astype(np.float32)
    assert np.abs(P.yuv2rgb(yuv) - O.yuv2rgb_matlab(yuv)).max() < 1e-9      # same constants as utils.py:106-115 (pinned)


def test_pwc_oracle_network_smoke():
    """Shapes / structure of the PWC-Net restatement: 182 variables, 14.08 M parameters, flows of levels 6..2 and a
    full-resolution prediction; flipping the pair negates nothing by construction but must change the result."""
    torch = pytest.importorskip("torch")
    import pwcnet_oracle as P
    shapes = P.variable_shapes()
    assert len(shapes) == 182 and sum(int(np.prod(s)) for s in shapes.values()) == 14079050
    assert shapes["pwcnet/predict_flow/conv6_0/kernel"] == (3, 3, 81, 128)
    assert shapes["pwcnet/predict_flow/conv2_4/kernel"] == (3, 3, 81 + 32 + 4 + 128 + 128 + 96 + 64, 32)
    assert shapes["pwcnet/upsample/up_feat3/kernel"] == (4, 4, 2, 81 + 64 + 4 + 448)
    W = P.synthetic_weights()
    x = torch.from_numpy(np.random.default_rng(0).random((1, 2, 64, 64, 3)))
    pred, pyr = P.nn(x, W)
    assert tuple(pred.shape) == (1, 64, 64, 2) and [tuple(p.shape[2:]) for p in pyr] == [(1, 1), (2, 2), (4, 4), (8, 8), (16, 16)]
    pred2, _ = P.nn(torch.flip(x, dims=[1]), W)
    assert float((pred - pred2).abs().max()) > 1e-6


def test_training_oracle_gradients_vs_finite_differences(syn_weights):
    """oracle/fisr_train_oracle.py (the four-pass loss of FISRnet.py:283-485) has no TensorFlow to be pinned to: its
    analytic gradients are checked against central differences of the restated loss, and the loss terms against a
    direct numpy evaluation of two of them."""
    import torch
    import fisr_train_oracle as fo
    W = syn_weights
    batch = fo.synthetic_batch(11, 1, 32, 32)
    loss, terms, grads = fo.loss_and_grads(W, batch)
    assert np.isfinite(loss) and set(grads) == set(W)
    args = [fo.to_nchw(batch[k]) for k in ("data15", "label21", "flow16", "warp24", "flow_ss2", "warp_ss2")]

    def total(Wd):
        Wt = {k: (torch.from_numpy(v).double().permute(3, 2, 0, 1).contiguous() if k.endswith("/w") else torch.from_numpy(v).double())
              for k, v in Wd.items()}
        with torch.no_grad():
            return float(fo.total_loss(Wt, *args)[0])

    for name, idx in (("FISRnet/level_3/SR/conv/2/b", (1,)), ("FISRnet/level_2/FI-SR/conv/2/w", (1, 1, 7, 4)),
                      ("FISRnet/level_3/dec/level_0/conv/0/w", (0, 2, 70, 3))):
        Wd = {k: np.array(v, np.float64) for k, v in W.items()}
        eps = 1e-5
        Wd[name][idx] += eps
        lp = total(Wd)
        Wd[name][idx] -= 2 * eps
        lm = total(Wd)
        fd = (lp - lm) / (2 * eps)
        assert abs(fd - grads[name][idx]) <= 2e-3 * abs(fd) + 1e-7, (name, fd, grads[name][idx])
    # total = s1 + ss2 * s2 with the default lambdas (FISRnet.py:389-391, 481-485)
    lam = fo.LAMBDAS
    s1 = lam["recn"] * terms["recn"] + lam["tm1"] * terms["tm"] + lam["tmm"] * terms["tmm"] + lam["td"] * terms["td"]
    s2 = lam["recn"] * terms["recn_ss2"] + lam["td"] * terms["td_ss2"] + lam["tm2"] * terms["tm_ss2"]
    assert abs(s1 + lam["ss2"] * s2 - loss) < 1e-9 * abs(loss)


def test_c_oracle_under_address_sanitizer():
    """SURVEY.md section 5 (sanitizers): the checker's own C code -- oracle/fisr_oracle.c, the thing every parity claim leans on --
    rebuilt with -fsanitize=address,undefined runs its conv / pool / forward / golden tests without a report (`make -C oracle
    asan-test`; a finding aborts the child process: -fno-sanitize-recover, ASan's default abort)."""
    import shutil
    import subprocess
    here = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cc = os.environ.get("CC", "gcc")
    if shutil.which(cc) is None or not os.path.isabs(subprocess.run([cc, "-print-file-name=libasan.so"], capture_output=True, text=True).stdout.strip()):
        pytest.skip("no gcc / libasan in this image")
    if os.environ.get("FISR_ORACLE_SO"):
        pytest.skip("already inside the sanitizer run")
    p = subprocess.run(["make", "-s", "-C", os.path.join(here, "oracle"), "asan-test"], capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, (p.stdout[-2000:], p.stderr[-3000:])
    assert " passed" in p.stdout and "ERROR: AddressSanitizer" not in p.stderr and "runtime error" not in p.stderr
