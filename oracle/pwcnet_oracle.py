"""CPU ORACLE of the optical-flow path (PWC-Net-large, 6-level pyramid, flow predicted at level 2) --
TEST INFRASTRUCTURE ONLY, never the product path.

Only `tests/` may import this file (fisr_amd/ must not).

A torch-CPU restatement (float64 by default) of what `main.py --phase FISR_for_video` runs before FISRnet
(reference main.py:207-211): `FISR_for_video_Compute_Flow`
(FISR_tfoptflow/FISR_for_video_pwcnet_predict_from_img_test.py:84-147) on top of the vendored
`ModelPWCNet.nn` (FISR_tfoptflow/model_pwcnet.py:1525-1593) with the options of that script (:96-100:
use_dense_cx, use_res_cx, pyr_lvls 6, flow_pred_lvl 2, search_range 4).  Every function cites the
reference lines it follows (paths relative to /root/reference).

PARITY PINNING STATUS -- all "parity unpinned" except the colour conversion and the two skimage resizes:
  * TensorFlow 1.13 arithmetic (tf.layers.conv2d 'same' incl. stride 2 = pad (0,1), dilation,
    conv2d_transpose 4x4/2 'same', leaky_relu, legacy resize_bilinear) is restated from the published
    semantics; TensorFlow is not installed and cannot be (no network).
  * `core_warp.dense_image_warp` and `core_costvol.cost_volume` are imported by model_pwcnet.py:25-26 from
    philferriere/tfoptflow but NOT vendored in the reference tree.  Restated from that project's published
    code: cost volume = mean over channels of c1 * shifted(zero-padded warp), 9x9 displacements, channel
    index (dy+4)*9 + (dx+4); warp = bilinear sample of c2 at (x + u, y + v) with the query clamped to the
    image (tf.contrib.image.dense_image_warp's interpolation), flow channels (u, v) = (x, y) displacement.
  * The PWC-Net checkpoint (`pwcnet-lg-6-2-multisteps-chairsthingsmix`, script :31) is absent: weights are
    synthetic (He-normal, seeded), keyed by the TF variable names the reference graph creates.
  * PINNED: YUV2RGB (script :43-52) against the imported reference function; the x2 up-resize and the
    anti-aliased down-resize of the script (:129-130, :139) against scikit-image 0.18.3 from the anaconda
    tree of this image (oracle/make_golden_pwc.py -> tests/golden/pwc_resize.npz).
"""
from __future__ import annotations

from collections import OrderedDict

import numpy as np
import torch
import torch.nn.functional as F

NUM_CHANN = [None, 16, 32, 64, 96, 128, 196]      # model_pwcnet.py:1084
PYR_LVLS, FLOW_PRED_LVL, SEARCH_RANGE = 6, 2, 4   # script :97-100, model_pwcnet.py:193
DENSE_CH = (128, 128, 96, 64, 32)                 # model_pwcnet.py:1426-1446
CTXT = ((128, 1), (128, 2), (128, 4), (96, 8), (64, 16), (32, 1), (2, 1))   # (filters, dilation) :1506-1519


def variable_shapes() -> "OrderedDict[str, tuple]":
    """TF variable names and shapes of the inference graph (tf.layers.conv2d: '<scope>/<name>/kernel' HWIO and
    '/bias'; conv2d_transpose kernel is [kh, kw, out, in]), scopes as nested by model_pwcnet.py:1084-1100 (featpyr),
    :1415-1447 (predict_flow), :1193-1196 (upsample), :1504-1521 (ctxt) under 'pwcnet' (:1525)."""
    d = OrderedDict()

    def conv(name, ci, co, k=3):
        d[name + "/kernel"] = (k, k, ci, co)
        d[name + "/bias"] = (co,)

    for lvl in range(1, PYR_LVLS + 1):
        ci = 3 if lvl == 1 else NUM_CHANN[lvl - 1]
        conv(f"pwcnet/featpyr/conv{lvl}a", ci, NUM_CHANN[lvl])
        conv(f"pwcnet/featpyr/conv{lvl}aa", NUM_CHANN[lvl], NUM_CHANN[lvl])
        conv(f"pwcnet/featpyr/conv{lvl}b", NUM_CHANN[lvl], NUM_CHANN[lvl])
    for lvl in range(PYR_LVLS, FLOW_PRED_LVL - 1, -1):
        c = (2 * SEARCH_RANGE + 1) ** 2 + (0 if lvl == PYR_LVLS else NUM_CHANN[lvl] + 2 + 2)
        for i, f in enumerate(DENSE_CH):
            conv(f"pwcnet/predict_flow/conv{lvl}_{i}", c, f)
            c += f
        conv(f"pwcnet/predict_flow/flow{lvl}", c, 2)
        ci = c
        for i, (f, _) in enumerate(CTXT):
            conv(f"pwcnet/ctxt/dc_conv{lvl}{i + 1}", ci, f)
            ci = f
        if lvl != FLOW_PRED_LVL:
            d[f"pwcnet/upsample/up_flow{lvl}/kernel"] = (4, 4, 2, 2)
            d[f"pwcnet/upsample/up_flow{lvl}/bias"] = (2,)
            d[f"pwcnet/upsample/up_feat{lvl}/kernel"] = (4, 4, 2, c)
            d[f"pwcnet/upsample/up_feat{lvl}/bias"] = (2,)
    return d


def synthetic_weights(seed: int = 595000, flow_gain: float = 1.0) -> "OrderedDict[str, np.ndarray]":
    """He-normal kernels (the reference's initialiser, model_pwcnet.py:1085), small biases, seeded.  The `flow*`,
    `dc_conv*7` and up-sampling kernels are scaled so that the predicted flows stay within a few pixels per
    level (a random network would otherwise warp far outside the image at every level)."""
    rng = np.random.default_rng(seed)
    out = OrderedDict()
    for name, shape in variable_shapes().items():
        if name.endswith("/bias"):
            out[name] = (rng.standard_normal(shape) * 0.01).astype(np.float32)
            continue
        fan_in = shape[0] * shape[1] * (shape[3] if "upsample" in name else shape[2])
        std = np.sqrt(2.0 / fan_in)
        if "/flow" in name or name.endswith("7/kernel") and "dc_conv" in name:
            std *= 0.5 * flow_gain
        if "upsample" in name:
            std = 0.25 * np.sqrt(1.0 / fan_in)
        out[name] = (rng.standard_normal(shape) * std).astype(np.float32)
    return out


# ---------------------------------------------------------------------------------- TF ops
def _t(w, dtype):                      # HWIO -> OIHW
    return torch.from_numpy(np.ascontiguousarray(w)).to(dtype).permute(3, 2, 0, 1).contiguous()


def conv2d_same(x, W, name, stride=1, dilation=1):
    """tf.layers.conv2d(x, f, 3, stride, 'same', dilation_rate=dilation) (model_pwcnet.py:1092-1097, 1426-1449,
    1506-1519).  TF 'SAME': out = ceil(in / stride); pad_total = max((out-1)*stride + (k-1)*dilation + 1 - in, 0),
    before = pad_total // 2, after = the rest -- for stride 2 on an even size that is (0, 1), NOT (1, 1)."""
    k = W[name + "/kernel"].shape[0]
    n, c, h, w = x.shape
    pads = []
    for size in (w, h):                                  # F.pad order: last dim first
        out = -(-size // stride)
        tot = max((out - 1) * stride + (k - 1) * dilation + 1 - size, 0)
        pads += [tot // 2, tot - tot // 2]
    x = F.pad(x, pads)
    return F.conv2d(x, _t(W[name + "/kernel"], x.dtype), torch.from_numpy(W[name + "/bias"]).to(x.dtype),
                    stride=stride, dilation=dilation)


def lrelu(x):
    """tf.nn.leaky_relu(x, alpha=0.1) (model_pwcnet.py:1093)."""
    return F.leaky_relu(x, 0.1)


def deconv(x, W, name):
    """tf.layers.conv2d_transpose(x, 2, 4, 2, 'same') (model_pwcnet.py:1196): output 2H x 2W; the transpose of a
    4x4 stride-2 'SAME' convolution whose padding is (1, 1): out[2*i + k - 1] += in[i] * kernel[k].  TF kernel
    layout [kh, kw, out, in] -> torch conv_transpose2d weight [in, out, kh, kw], padding 1."""
    w = torch.from_numpy(np.ascontiguousarray(W[name + "/kernel"])).to(x.dtype).permute(3, 2, 0, 1).contiguous()
    return F.conv_transpose2d(x, w, torch.from_numpy(W[name + "/bias"]).to(x.dtype), stride=2, padding=1)


def cost_volume(c1, warp, search_range=SEARCH_RANGE):
    """core_costvol.cost_volume (imported at model_pwcnet.py:26, called :1277; not vendored): zero-pad `warp` by the
    search range, for every displacement (dy, dx) in row-major order mean over channels of c1 * shifted warp."""
    n, c, h, w = c1.shape
    p = F.pad(warp, (search_range,) * 4)
    out = []
    for y in range(2 * search_range + 1):
        for x in range(2 * search_range + 1):
            out.append((c1 * p[:, :, y:y + h, x:x + w]).mean(dim=1, keepdim=True))
    return torch.cat(out, dim=1)


def dense_image_warp(img, flow):
    """core_warp.dense_image_warp (imported at model_pwcnet.py:25, called :1178; not vendored): bilinear sample of
    `img` at (x + u, y + v), flow[:, 0] = u (x), flow[:, 1] = v (y), with tf.contrib's interpolation: floor index
    clamped to [0, size-2], weight clamped to [0, 1] (= the query clamped to the image)."""
    n, c, h, w = img.shape
    gy, gx = torch.meshgrid(torch.arange(h, dtype=img.dtype), torch.arange(w, dtype=img.dtype), indexing="ij")
    qx = gx[None] + flow[:, 0]
    qy = gy[None] + flow[:, 1]

    def split(q, size):
        f = torch.clamp(torch.floor(q), 0, size - 2)
        a = torch.clamp(q - f, 0, 1)
        return f.long(), a

    x0, ax = split(qx, w)
    y0, ay = split(qy, h)
    flat = img.reshape(n, c, h * w)

    def g(yy, xx):
        idx = (yy * w + xx).reshape(n, 1, h * w).expand(n, c, h * w)
        return torch.gather(flat, 2, idx).reshape(n, c, h, w)

    tl, tr, bl, br = g(y0, x0), g(y0, x0 + 1), g(y0 + 1, x0), g(y0 + 1, x0 + 1)
    ax, ay = ax[:, None], ay[:, None]
    top = ax * (tr - tl) + tl                       # tf.contrib _interpolate_bilinear evaluation order
    bot = ax * (br - bl) + bl
    return ay * (bot - top) + top


def resize_bilinear_legacy(x, scale):
    """tf.image.resize_bilinear(flow, size) (model_pwcnet.py:1590), TF-1.13 legacy kernel (align_corners=False, no
    half-pixel centres): src = dst / scale, lo = floor(src), hi = min(lo + 1, n - 1), lerp; rows of x first
    (top = tl + (tr - tl) * fx ...) as in fisr_oracle.resize_bilinear_x2."""
    n, c, h, w = x.shape
    oy = torch.arange(h * scale, dtype=x.dtype) / scale
    ox = torch.arange(w * scale, dtype=x.dtype) / scale
    y0 = torch.floor(oy).long(); y1 = torch.clamp(y0 + 1, max=h - 1); ty = (oy - y0)[None, None, :, None]
    x0 = torch.floor(ox).long(); x1 = torch.clamp(x0 + 1, max=w - 1); tx = (ox - x0)[None, None, None, :]
    tl, tr = x[:, :, y0][:, :, :, x0], x[:, :, y0][:, :, :, x1]
    bl, br = x[:, :, y1][:, :, :, x0], x[:, :, y1][:, :, :, x1]
    top = tl + (tr - tl) * tx
    bot = bl + (br - bl) * tx
    return top + (bot - top) * ty


# ---------------------------------------------------------------------------------- the network
def extract_features(x, W):
    """model_pwcnet.py:1012-1101: six levels of conv(stride 2) - conv - conv, each + leaky relu."""
    pyr = [None]
    for lvl in range(1, PYR_LVLS + 1):
        x = lrelu(conv2d_same(x, W, f"pwcnet/featpyr/conv{lvl}a", stride=2))
        x = lrelu(conv2d_same(x, W, f"pwcnet/featpyr/conv{lvl}aa"))
        x = lrelu(conv2d_same(x, W, f"pwcnet/featpyr/conv{lvl}b"))
        pyr.append(x)
    return pyr


def predict_flow(corr, c1, up_flow, up_feat, lvl, W):
    """model_pwcnet.py:1282-1451 with use_dense_cx: every conv's activation is concatenated IN FRONT of its input."""
    x = corr if c1 is None else torch.cat([corr, c1, up_flow, up_feat], dim=1)
    for i in range(5):
        act = lrelu(conv2d_same(x, W, f"pwcnet/predict_flow/conv{lvl}_{i}"))
        x = torch.cat([act, x], dim=1)
    return x, conv2d_same(x, W, f"pwcnet/predict_flow/flow{lvl}")


def refine_flow(feat, flow, lvl, W):
    """model_pwcnet.py:1453-1523: dilated context network, added to the flow."""
    x = feat
    for i, (_, dil) in enumerate(CTXT):
        x = conv2d_same(x, W, f"pwcnet/ctxt/dc_conv{lvl}{i + 1}", dilation=dil)
        if i < len(CTXT) - 1:
            x = lrelu(x)
    return flow + x


def nn(x_pair, W, taps=None):
    """model_pwcnet.py:1525-1593.  x_pair [N, 2, H, W, 3] in 0..1, H and W multiples of 64 -> (flow_pred [N, H, W, 2],
    [flow6 .. flow2])."""
    dtype = x_pair.dtype
    im1 = x_pair[:, 0].permute(0, 3, 1, 2)
    im2 = x_pair[:, 1].permute(0, 3, 1, 2)
    c1, c2 = extract_features(im1, W), extract_features(im2, W)
    pyr = []
    up_flow = up_feat = None
    for lvl in range(PYR_LVLS, FLOW_PRED_LVL - 1, -1):
        if lvl == PYR_LVLS:
            corr = lrelu(cost_volume(c1[lvl], c2[lvl]))          # core_costvol.cost_volume ends with leaky_relu(0.1)
            upfeat, flow = predict_flow(corr, None, None, None, lvl, W)
        else:
            scaler = 20.0 / 2 ** lvl                              # :1560
            warp = dense_image_warp(c2[lvl], up_flow * scaler)
            corr = lrelu(cost_volume(c1[lvl], warp))
            upfeat, flow = predict_flow(corr, c1[lvl], up_flow, up_feat, lvl, W)
        flow = refine_flow(upfeat, flow, lvl, W)                  # use_res_cx: every level (:1573-1574, :1585)
        pyr.append(flow)
        if taps is not None:
            taps[f"flow{lvl}"] = flow
        if lvl != FLOW_PRED_LVL:
            up_flow = deconv(flow, W, f"pwcnet/upsample/up_flow{lvl}")
            up_feat = deconv(upfeat, W, f"pwcnet/upsample/up_feat{lvl}")
    scaler = 2 ** FLOW_PRED_LVL
    flow_pred = resize_bilinear_legacy(flow, scaler) * scaler     # :1587-1590
    return flow_pred.permute(0, 2, 3, 1), pyr


# ---------------------------------------------------------------------------------- the driver script
def yuv2rgb(yuv):
    """script :43-52 (float64 maths, clip 0..255, no rounding)."""
    tinv = np.array([[0.00456621, 0., 0.00625893], [0.00456621, -0.00153632, -0.00318811], [0.00456621, 0.00791071, 0.]])
    t = 255 * tinv
    off = 255 * tinv @ np.array([[16], [128], [128]])
    rgb = np.zeros(yuv.shape)
    for p in range(3):
        rgb[:, :, p] = t[p, 0] * yuv[:, :, 0] + t[p, 1] * yuv[:, :, 1] + t[p, 2] * yuv[:, :, 2] - off[p]
    return np.clip(rgb, 0, 255)


def resize_up2_skimage(img):
    """skimage.transform.resize(img, (2h, 2w)) of the script (:129-130): order 1, mode 'reflect', half-pixel centres
    (src = (dst + 0.5) / 2 - 0.5), no anti-aliasing when up-sampling.  With scale exactly 2 the weights are
    (0.25, 0.75): out[2i] = 0.25*in[i-1] + 0.75*in[i], out[2i+1] = 0.75*in[i] + 0.25*in[i+1]; skimage's 'reflect'
    is scipy.ndimage's 'mirror' (d c b | a b c d | c b a: about the edge pixel's centre), i.e. in[-1] = in[1],
    in[n] = in[n-2] (pinned against scikit-image, tests/golden/pwc_resize.npz)."""
    x = np.asarray(img, np.float64)
    for ax in (0, 1):
        lo = np.concatenate([np.take(x, [1], axis=ax), np.take(x, range(0, x.shape[ax] - 1), axis=ax)], axis=ax)
        hi = np.concatenate([np.take(x, range(1, x.shape[ax]), axis=ax), np.take(x, [x.shape[ax] - 2], axis=ax)], axis=ax)
        even = 0.25 * lo + 0.75 * x
        odd = 0.75 * x + 0.25 * hi
        x = np.stack([even, odd], axis=ax + 1).reshape(x.shape[:ax] + (2 * x.shape[ax],) + x.shape[ax + 1:])
    return x


def resize_down2_skimage_aa(flow):
    """skimage.transform.resize(flow, (N, h, w, 2), anti_aliasing=True) of the script (:139) on [N, 2h, 2w, 2]:
    Gaussian pre-filter with sigma = (factor - 1) / 2 = 0.5 along the two down-scaled axes (scipy.ndimage.gaussian_filter,
    truncate 4.0 -> radius 2, mode 'mirror' as skimage passes it), then order-1 resampling with half-pixel centres:
    src = (dst + 0.5) * 2 - 0.5 = 2*dst + 0.5 -> the mean of samples 2i and 2i+1."""
    x = np.asarray(flow, np.float64)
    r = 2
    k = np.exp(-0.5 * (np.arange(-r, r + 1) / 0.5) ** 2)
    k /= k.sum()
    for ax in (1, 2):
        n = x.shape[ax]
        idx = np.arange(-r, n + r)
        idx = np.where(idx < 0, -idx, idx)                       # 'mirror': d c b | a b c d | c b a  (no edge repeat)
        idx = np.where(idx >= n, 2 * n - 2 - idx, idx)
        xp = np.take(x, idx, axis=ax)
        x = sum(k[j] * np.take(xp, range(j, j + n), axis=ax) for j in range(2 * r + 1))
    x = 0.5 * (x[:, 0::2] + x[:, 1::2])
    x = 0.5 * (x[:, :, 0::2] + x[:, :, 1::2])
    return x


def compute_flow_pair(yuv_a, yuv_b, W, dtype=torch.float64):
    """One iteration of the script's loop (:118-140): -> flows [2, h, w, 2] (a->b, b->a) in LR pixels."""
    h, w = yuv_a.shape[:2]
    imgs = [np.array(resize_up2_skimage(yuv2rgb(np.asarray(f, np.float32))), dtype=np.uint8) for f in (yuv_a, yuv_b)]
    pairs = np.stack([np.stack([imgs[0], imgs[1]]), np.stack([imgs[1], imgs[0]])]).astype(np.float32) / 255.0   # adapt_x :399
    H, Wd = pairs.shape[2:4]
    ph, pw = (-H) % 64, (-Wd) % 64
    pairs = np.pad(pairs, [(0, 0), (0, 0), (0, ph), (0, pw), (0, 0)])                                            # :401-411
    flow, _ = nn(torch.from_numpy(pairs).to(dtype), W)
    flow = flow.numpy()[:, :H, :Wd]                                                                              # :461-463
    return resize_down2_skimage_aa(flow) / 2.0                                                                   # :139
