"""CPU ORACLE (numpy twin) -- TEST INFRASTRUCTURE ONLY, never the product path.

Only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may
import this file.  `fisr_amd/` must never import anything from `oracle/`.

A restatement, in numpy float64 (or float32 on request), of the reference's
FISRnet inference path.  Every function cites the reference file:line it follows
(paths relative to /root/reference).

PARITY PINNING STATUS
  * The reference is Python/TensorFlow-1.13 + OpenCV; TensorFlow, cv2, h5py and
    SSIM_PIL are not installed here and there is no network, and the reference has
    no tests, golden vectors or checkpoint.
  * PINNED (against the reference's own code imported in the build container with
    stubbed tensorflow/h5py -- see oracle/make_golden.py, fixtures under
    tests/golden/ref_utils.npz): get_HW_boundary, trim_patch_boundary,
    merge_seq_dim, split_seq_dim, YUV2RGB_matlab, _compute_psnr,
    read_flo_file_5dim, YUV2RGB / RGB2YUV of the warp script.
  * PARITY UNPINNED (third-party arithmetic absent from the tree, restated from the
    published algorithm of the pinned dependency version):
      - tensorflow==1.13.1: conv2d SAME, max_pool, legacy resize_images
        (bilinear/bicubic, align_corners=False, no half-pixel centres),
        depth_to_space, concat/split.  conv2d/max_pool are cross-checked against
        torch-CPU float64 (tests/test_oracle.py).
      - opencv_python==4.2.0.32: cv2.remap(INTER_LINEAR, BORDER_REPLICATE).
      - SSIM_PIL==1.0.10: compare_ssim.
"""
from __future__ import annotations

import numpy as np

# ----------------------------------------------------------------------------------
# ops.py layers
# ----------------------------------------------------------------------------------


def conv2d(x, w, b):
    """ops.py:7-11 Conv2d: tf.nn.conv2d(x, w, strides 1, 'SAME') + b.
    x [N,H,W,Ci], w [3,3,Ci,Co] (HWIO), cross-correlation, zero padding 1."""
    n, h, wd, ci = x.shape
    xp = np.zeros((n, h + 2, wd + 2, ci), x.dtype)
    xp[:, 1:-1, 1:-1, :] = x
    y = np.zeros((n, h, wd, w.shape[3]), x.dtype)
    for dy in range(3):
        for dx in range(3):
            y += xp[:, dy:dy + h, dx:dx + wd, :] @ w[dy, dx].astype(x.dtype)
    return y + b.astype(x.dtype)


def relu(x):
    """ops.py:17-18."""
    return np.maximum(x, 0)


def max_pool2(x):
    """ops.py:54 tf.nn.max_pool 2x2 stride 2 'SAME'; sizes are even on this path."""
    n, h, w, c = x.shape
    assert h % 2 == 0 and w % 2 == 0
    return x.reshape(n, h // 2, 2, w // 2, 2, c).max(axis=(2, 4))


def resize_bilinear_x2(x):
    """ops.py:69 tf.image.resize_images(BILINEAR) to exactly 2x, TF-1.13 legacy
    kernel (align_corners=False, no half-pixel centres): in = out*0.5,
    lo=floor(in), hi=min(lo+1,n-1), t=in-lo; evaluation order
    top=tl+(tr-tl)*tx; bot=bl+(br-bl)*tx; out=top+(bot-top)*ty."""
    n, h, w, c = x.shape
    oy = np.arange(2 * h)
    ox = np.arange(2 * w)
    y0 = oy // 2
    y1 = np.minimum(y0 + 1, h - 1)
    ty = ((oy % 2) * 0.5).astype(x.dtype)[None, :, None, None]
    x0 = ox // 2
    x1 = np.minimum(x0 + 1, w - 1)
    tx = ((ox % 2) * 0.5).astype(x.dtype)[None, None, :, None]
    tl = x[:, y0][:, :, x0]
    tr = x[:, y0][:, :, x1]
    bl = x[:, y1][:, :, x0]
    br = x[:, y1][:, :, x1]
    top = tl + (tr - tl) * tx
    bot = bl + (br - bl) * tx
    return top + (bot - top) * ty


def resize_bicubic_down(x, s):
    """FISRnet.py:81,112 tf.image.resize_images(BICUBIC) to 1/s, TF-1.13 legacy:
    in = out*s exactly, fractional part 0 -> cubic weights (0,1,0,0) -> pure strided
    sub-sampling."""
    return x[:, ::s, ::s, :]


def depth_to_space2(x):
    """FISRnet.py:99 tf.depth_to_space(x, 2), NHWC 'DCR':
    out[n,2h+i,2w+j,c] = x[n,h,w,(2i+j)*C+c]."""
    n, h, w, c4 = x.shape
    c = c4 // 4
    return x.reshape(n, h, w, 2, 2, c).transpose(0, 1, 3, 2, 4, 5).reshape(n, 2 * h, 2 * w, c)


def res_block(x, W, name):
    """ops.py:39-44."""
    n = conv2d(relu(x), W[name + "/conv/0/w"], W[name + "/conv/0/b"])
    n = conv2d(relu(n), W[name + "/conv/1/w"], W[name + "/conv/1/b"])
    return x + n


def enc_level_res(x, W, name):
    """ops.py:48-55 -> (pooled, skip)."""
    n = conv2d(x, W[name + "/conv/0/w"], W[name + "/conv/0/b"])
    n = res_block(n, W, name + "/res_block/0")
    n = relu(res_block(n, W, name + "/res_block/1"))
    return max_pool2(n), n


def bottleneck_res(x, W, name):
    """ops.py:59-63."""
    n = conv2d(x, W[name + "/conv/0/w"], W[name + "/conv/0/b"])
    return relu(res_block(n, W, name + "/res_block/0"))


def dec_level_res(x, skip, W, name):
    """ops.py:67-76 (size is always exactly 2x: FISRnet.py:91-93)."""
    n = resize_bilinear_x2(x)
    n = relu(conv2d(n, W[name + "/resize/w"], W[name + "/resize/b"]))
    n = np.concatenate([n, skip], axis=3)
    n = conv2d(n, W[name + "/conv/0/w"], W[name + "/conv/0/b"])
    n = res_block(n, W, name + "/res_block/0")
    return relu(res_block(n, W, name + "/res_block/1"))


def _level(x, W, p, taps=None):
    """One U-Net + heads at scope p = 'FISRnet/level_k' (FISRnet.py:83-108)."""
    skip = {}
    n, skip[0] = enc_level_res(x, W, p + "/enc/level_0")
    n, skip[1] = enc_level_res(n, W, p + "/enc/level_1")
    n, skip[2] = enc_level_res(n, W, p + "/enc/level_2")
    n = bottleneck_res(n, W, p + "/bottleneck")
    n = dec_level_res(n, skip[2], W, p + "/dec/level_2")
    n = dec_level_res(n, skip[1], W, p + "/dec/level_1")
    n = dec_level_res(n, skip[0], W, p + "/dec/level_0")
    if taps is not None:
        taps[p + "/dec_out"] = n
    outs = []
    for head in ("FI-SR", "SR"):
        h = p + "/" + head
        a = conv2d(n, W[h + "/conv/0/w"], W[h + "/conv/0/b"])
        a = res_block(a, W, h + "/res_block/0")
        a = conv2d(relu(a), W[h + "/conv/1/w"], W[h + "/conv/1/b"])
        a = depth_to_space2(relu(a))
        # FI-SR applies relu again (idempotent), SR does not (FISRnet.py:100,106)
        a = conv2d(relu(a) if head == "FI-SR" else a, W[h + "/conv/2/w"], W[h + "/conv/2/b"])
        outs.append(a)
    fisr, sr = outs
    return np.concatenate([fisr[..., 0:3], sr, fisr[..., 3:6]], axis=3)  # FISRnet.py:107-108


def model(img, W, dtype=np.float64, taps=None):
    """FISRnet.py:73-173 model(img[N,H,W,29], sf=2) -> (pred_l1, pred_l2, pred_l3)."""
    img = np.asarray(img, dtype)
    assert img.shape[1] % 32 == 0 and img.shape[2] % 32 == 0
    p1 = _level(resize_bicubic_down(img, 4), W, "FISRnet/level_1", taps)
    x2 = np.concatenate([resize_bicubic_down(img, 2), p1], axis=3)
    p2 = _level(x2, W, "FISRnet/level_2", taps)
    x3 = np.concatenate([img, p2], axis=3)
    p3 = _level(x3, W, "FISRnet/level_3", taps)
    return p1, p2, p3


# ----------------------------------------------------------------------------------
# utils.py helpers (pinned against the imported reference, see header)
# ----------------------------------------------------------------------------------


def compute_psnr(a, b, peak=1.0):
    """utils.py:23-26."""
    mse = np.mean(np.square(np.asarray(a, np.float64) - np.asarray(b, np.float64)))
    return 10 * np.log10(peak * peak / mse)


def merge_seq_dim(data):
    """utils.py:78-83: [N,S,H,W,C] -> [N,H,W,S*C]."""
    sz = data.shape
    return np.transpose(data, (0, 2, 3, 1, 4)).reshape(sz[0], sz[2], sz[3], sz[1] * sz[4])


def split_seq_dim(data):
    """utils.py:86-91."""
    sz = data.shape
    return np.transpose(data.reshape(sz[0], sz[1], sz[2], sz[3] // 3, 3), (0, 3, 1, 2, 4))


_TINV = np.array([[0.00456621, 0., 0.00625893],
                  [0.00456621, -0.00153632, -0.00318811],
                  [0.00456621, 0.00791071, 0.]])


def yuv2rgb_matlab(yuv):
    """utils.py:106-115 (== warp script YUV2RGB :35-45).  float64, clip [0,255]."""
    yuv = np.asarray(yuv)
    T = 255 * _TINV
    offset = 255 * _TINV @ np.array([[16], [128], [128]])
    rgb = np.zeros(yuv.shape)
    for p in range(3):
        rgb[:, :, p] = T[p, 0] * yuv[:, :, 0] + T[p, 1] * yuv[:, :, 1] + T[p, 2] * yuv[:, :, 2] - offset[p]
    return np.clip(rgb, 0, 255)


def rgb2yuv(rgb):
    """FISR_tfoptflow/FISR_for_video_warp_img_with_flo.py:48-57."""
    T = np.array([[65.481, 128.553, 24.966], [-37.797, -74.203, 112], [112, -93.786, -18.214]]) / 255
    offset = [16, 128, 128]
    yuv = np.zeros(rgb.shape)
    for p in range(3):
        yuv[:, :, p] = T[p, 0] * rgb[:, :, 0] + T[p, 1] * rgb[:, :, 1] + T[p, 2] * rgb[:, :, 2] + offset[p]
    return np.clip(yuv, 0, 255)


def get_hw_boundary(pb, h, w, pH, sH, pW, sW):
    """utils.py:118-135."""
    hl = max(pH * sH - pb, 0)
    hh = min((pH + 1) * sH + pb, h)
    wl = max(pW * sW - pb, 0)
    wh = min((pW + 1) * sW + pb, w)
    add_h = (pb if pH * sH >= pb else 0) + (pb if (pH + 1) * sH + pb <= h else 0)
    add_w = (pb if pW * sW >= pb else 0) + (pb if (pW + 1) * sW + pb <= w else 0)
    return hl, hh, wl, wh, add_h, add_w


def trim_patch_boundary(img, pb, h, w, pH, sH, pW, sW, sf):
    """utils.py:138-159."""
    if pb == 0:
        return img
    if not pH * sH < pb:
        img = img[:, pb * sf:, :, :]
    if not (pH + 1) * sH + pb > h:
        img = img[:, :-pb * sf, :, :]
    if not pW * sW < pb:
        img = img[:, :, pb * sf:, :]
    if not (pW + 1) * sW + pb > w:
        img = img[:, :, :-pb * sf, :]
    return img


# ----------------------------------------------------------------------------------
# Harness pieces: input assembly, tiled forward, post-processing
# ----------------------------------------------------------------------------------


def assemble_input(frames_u8, flow, warp):
    """FISRnet.py:828-843.  frames_u8 [h,w,9] uint8 (3 YUV frames), flow [h,w,8]
    float32 pixels (4 flows), warp [h,w,12] float32 already /255 (utils.py:51).
    -> [1,h,w,29] float64."""
    img = np.clip(np.array(frames_u8, dtype=np.double) / 255., 0, 1)
    fl = np.clip(np.asarray(flow) / 96 / 2, -1, 1)          # FISRnet.py:835-836 (96 hard-coded)
    wp = np.clip(np.asarray(warp), 0, 1)                     # :839-840
    return np.concatenate([img, fl, wp], axis=2)[None]


def crop_hw(H, W, num_patch):
    """FISRnet.py:820-824."""
    return H - H % (32 * num_patch[0]), W - W % (32 * num_patch[1])


def tiled_forward(inp, W, num_patch=(2, 2), pb=32, sf=2, forward=None, dtype=np.float64):
    """FISRnet.py:845-883 tile loop + stitch + clip.  inp [1,h,w,29] -> [h*sf,w*sf,9]."""
    _, h, w, _ = inp.shape
    full = np.zeros((h * sf, w * sf, 9))
    if forward is None:
        forward = lambda t: model(t, W, dtype)[2]
    for p in range(num_patch[0] * num_patch[1]):
        pH, pW = p // num_patch[1], p % num_patch[1]
        sH, sW = h // num_patch[0], w // num_patch[1]
        hl, hh, wl, wh, _, _ = get_hw_boundary(pb, h, w, pH, sH, pW, sW)
        pred = forward(inp[:, hl:hh, wl:wh, :])
        pred = trim_patch_boundary(np.asarray(pred), pb, h, w, pH, sH, pW, sW, sf)
        full[pH * sH * sf:(pH + 1) * sH * sf, pW * sW * sf:(pW + 1) * sW * sf, :] = np.squeeze(pred, 0)
    return np.clip(full, 0, 1)


def quantize_u8(pred):
    """FISRnet.py:903 np.uint8(pred*255): truncation toward zero."""
    return np.uint8(np.asarray(pred, np.float64) * 255)


def yuv_u8_to_rgb_u8(yuv_u8):
    """FISRnet.py:908-909: YUV2RGB_matlab(uint8 yuv).astype('uint8') (truncation)."""
    return yuv2rgb_matlab(yuv_u8).astype("uint8")


# ----------------------------------------------------------------------------------
# Frame warp (cv2.remap restatement)
# ----------------------------------------------------------------------------------


def remap_linear_replicate(src, mapx, mapy, quantized=True):
    """cv2.remap(src, map, None, INTER_LINEAR, None, BORDER_REPLICATE) for a float64
    HxWxC src and float32 maps, as OpenCV 4.2 imgwarp.cpp computes it:
    fixed-point coordinates sx = cvRound(x*32) (round-half-even), integer part
    sx>>5, fraction (sx&31)/32 from the float bilinear table
    w = {(1-fy)(1-fx), (1-fy)fx, fy(1-fx), fy fx} (float32 products, exact), sum in
    the work type of the source (double for a float64 source), taps clamped to the
    image (BORDER_REPLICATE).  quantized=False gives exact-coordinate bilinear."""
    h, w = src.shape[:2]
    mapx = np.asarray(mapx, np.float32)
    mapy = np.asarray(mapy, np.float32)
    if quantized:
        sx = np.rint(mapx * np.float32(32)).astype(np.int64)
        sy = np.rint(mapy * np.float32(32)).astype(np.int64)
        ix, iy = sx >> 5, sy >> 5
        fx = ((sx & 31).astype(np.float32) / np.float32(32))
        fy = ((sy & 31).astype(np.float32) / np.float32(32))
    else:
        ix = np.floor(mapx).astype(np.int64)
        iy = np.floor(mapy).astype(np.int64)
        fx = (mapx - ix).astype(np.float32)
        fy = (mapy - iy).astype(np.float32)
    one = np.float32(1)
    w00 = ((one - fy) * (one - fx)).astype(np.float64)[..., None]
    w01 = ((one - fy) * fx).astype(np.float64)[..., None]
    w10 = (fy * (one - fx)).astype(np.float64)[..., None]
    w11 = (fy * fx).astype(np.float64)[..., None]
    x0 = np.clip(ix, 0, w - 1)
    x1 = np.clip(ix + 1, 0, w - 1)
    y0 = np.clip(iy, 0, h - 1)
    y1 = np.clip(iy + 1, 0, h - 1)
    s = np.asarray(src, np.float64)
    return s[y0, x0] * w00 + s[y0, x1] * w01 + s[y1, x0] * w10 + s[y1, x1] * w11


def warp_frame(src_yuv, flow, quantized=True):
    """FISR_for_video_warp_img_with_flo.py:112-129 for one direction:
    src_yuv [h,w,3] (uint8 or float YUV 0..255 of the *other* frame of the pair),
    flow [h,w,2] float32 pixels.  YUV2RGB (:35-45, float64) -> warp_flow(rgb, 0.5*flow)
    (:61-67: map = 0.5*flow + pixel grid, float32) -> RGB2YUV (:48-57) -> float32 store
    (pred array dtype, :107).  Returns float32 [h,w,3] in 0..255."""
    h, w = flow.shape[:2]
    rgb = yuv2rgb_matlab(np.array(src_yuv, dtype=np.float32))
    f = np.asarray(flow, np.float32) * np.float32(0.5)
    mapx = (f[:, :, 0] + np.arange(w, dtype=np.float32)[None, :]).astype(np.float32)
    mapy = (f[:, :, 1] + np.arange(h, dtype=np.float32)[:, None]).astype(np.float32)
    res = remap_linear_replicate(rgb, mapx, mapy, quantized)
    return rgb2yuv(res).astype(np.float32)


# ----------------------------------------------------------------------------------
# SSIM_PIL restatement (PARITY UNPINNED; from the published SSIM-PIL 1.0.10 CPU path)
# ----------------------------------------------------------------------------------


def ssim_pil(a_u8, b_u8, tile=7):
    """FISRnet.py:890-891 compare_ssim(Image(a), Image(b)): non-overlapping 7x7 tiles,
    per channel, C1=(0.01*255)^2, C2=(0.03*255)^2, unbiased (N-1) variance/covariance,
    mean over tiles x channels."""
    a = np.asarray(a_u8, np.float64)
    b = np.asarray(b_u8, np.float64)
    h, w, c = a.shape
    th, tw = h // tile, w // tile
    a = a[:th * tile, :tw * tile].reshape(th, tile, tw, tile, c)
    b = b[:th * tile, :tw * tile].reshape(th, tile, tw, tile, c)
    n = tile * tile
    ma = a.mean(axis=(1, 3), keepdims=True)
    mb = b.mean(axis=(1, 3), keepdims=True)
    va = ((a - ma) ** 2).sum(axis=(1, 3)) / (n - 1)
    vb = ((b - mb) ** 2).sum(axis=(1, 3)) / (n - 1)
    cov = ((a - ma) * (b - mb)).sum(axis=(1, 3)) / (n - 1)
    ma = ma[:, 0, :, 0]
    mb = mb[:, 0, :, 0]
    c1, c2 = (255 * 0.01) ** 2, (255 * 0.03) ** 2
    s = ((2 * ma * mb + c1) * (2 * cov + c2)) / ((ma ** 2 + mb ** 2 + c1) * (va + vb + c2))
    return float(s.mean())


# ----------------------------------------------------------------------------------
# .flo (FISR 5-D variant) reader/writer restatement
# ----------------------------------------------------------------------------------


def write_flo5(path, flow):
    """FISR_tfoptflow/FISR_for_video_pwcnet_predict_from_img_test.py:57-81 write_flow:
    float32 magic 202021.25, int32 N, N_seq, h, w, float32 data [N,N_seq,h,w,2]."""
    flow = np.asarray(flow, np.float32)
    n, s, h, w, _ = flow.shape
    with open(path, "wb") as f:
        np.array([202021.25], np.float32).tofile(f)
        np.array([n, s, h, w], np.int32).tofile(f)
        flow.tofile(f)


def read_flo5(path):
    """utils.py:57-74 read_flo_file_5dim."""
    with open(path, "rb") as f:
        magic = np.fromfile(f, np.float32, count=1)
        if magic.size != 1 or magic[0] != np.float32(202021.25):
            raise ValueError("Magic number incorrect. Invalid .flo file")
        n, s, h, w = (int(v) for v in np.fromfile(f, np.int32, count=4))
        data = np.fromfile(f, np.float32, count=n * s * h * w * 2)
    return np.resize(data, (n, s, h, w, 2))
