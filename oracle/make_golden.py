"""Generate the committed fixtures under tests/golden/  (run in the BUILD container only).

  python oracle/make_golden.py

1. ref_utils.npz  -- known answers captured by importing the reference's own pure-numpy
   helpers from /root/reference (utils.py, FISR_tfoptflow/FISR_for_video_warp_img_with_flo.py)
   with stub modules for tensorflow / h5py / cv2 / hdf5storage.  This pins the oracle's
   (and the product host code's) tiling, layout, colour and PSNR helpers to the reference.
2. scene1_crop96.npz -- the 96x96 LR crop (rows 492:588, cols 912:1008) of the five
   FISR_test_folder/scene1 PNGs (the only real data in the reference tree), plus seeded
   synthetic flows and the oracle's warps of those frames (cfg1 of BASELINE.json).
3. model_32x64.npz / model_96.npz -- oracle (float64) outputs of the FISRnet forward on
   seeded inputs with `fisr_amd.weights.synthetic_weights(2020)` (weights are regenerated
   from the seed, not stored).

/root/reference never travels to the GPU box; only these data files do.
"""
import glob
import os
import sys
import types

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
GOLD = os.path.join(ROOT, "tests", "golden")
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))


def import_reference_helpers():
    for name in ("tensorflow", "tensorflow.contrib", "tensorflow.contrib.slim", "h5py", "cv2",
                 "hdf5storage", "matplotlib", "matplotlib.pyplot"):
        if name not in sys.modules:
            sys.modules[name] = types.ModuleType(name)
    sys.modules["tensorflow"].contrib = sys.modules["tensorflow.contrib"]
    sys.modules["tensorflow.contrib"].slim = sys.modules["tensorflow.contrib.slim"]
    sys.modules["matplotlib"].pyplot = sys.modules["matplotlib.pyplot"]
    sys.path.insert(0, REF)
    import utils as ref_utils  # noqa
    sys.path.insert(0, os.path.join(REF, "FISR_tfoptflow"))
    import FISR_for_video_warp_img_with_flo as ref_warp  # noqa
    return ref_utils, ref_warp


def gen_ref_utils():
    ru, rw = import_reference_helpers()
    rng = np.random.default_rng(11)
    out = {}
    # tiling maths for the default 1080p config and some ragged ones
    cases = [(1024, 1920, 2, 2), (1056, 1920, 1, 1), (96, 96, 1, 1), (128, 192, 2, 3), (64, 64, 2, 2)]
    rows = []
    for h, w, nh, nw in cases:
        for p in range(nh * nw):
            pH, pW = p // nw, p % nw
            sH, sW = h // nh, w // nw
            b = ru.get_HW_boundary(32, h, w, pH, sH, pW, sW)
            t = np.zeros((1, (b[1] - b[0]) * 2, (b[3] - b[2]) * 2, 1), np.int8)
            tr = ru.trim_patch_boundary(t, 32, h, w, pH, sH, pW, sW, 2)
            rows.append([h, w, nh, nw, p, *b, tr.shape[1], tr.shape[2]])
    out["tiling"] = np.array(rows, np.int64)
    # exact content of a trim (which rows/cols survive)
    h, w = 128, 192
    for p in range(4):
        pH, pW = p // 2, p % 2
        b = ru.get_HW_boundary(32, h, w, pH, h // 2, pW, w // 2)
        th, tw = (b[1] - b[0]) * 2, (b[3] - b[2]) * 2
        t = np.arange(th * tw, dtype=np.int64).reshape(1, th, tw, 1)
        out[f"trim_{p}"] = ru.trim_patch_boundary(t, 32, h, w, pH, h // 2, pW, w // 2, 2)
    x5 = rng.standard_normal((2, 3, 4, 5, 3)).astype(np.float32)
    out["seq_in"] = x5
    out["merge_seq"] = ru.merge_seq_dim(x5)
    out["split_seq"] = ru.split_seq_dim(ru.merge_seq_dim(x5))
    yuv = rng.integers(0, 256, (6, 7, 3)).astype(np.uint8)
    out["yuv_u8"] = yuv
    out["yuv2rgb_matlab"] = ru.YUV2RGB_matlab(yuv)
    yuvf = (rng.random((6, 7, 3)) * 255).astype(np.float32)
    out["yuv_f32"] = yuvf
    out["warp_yuv2rgb"] = rw.YUV2RGB(yuvf)
    rgbf = rng.random((6, 7, 3)) * 255
    out["rgb_f64"] = rgbf
    out["warp_rgb2yuv"] = rw.RGB2YUV(rgbf)
    a, b = rng.random((5, 5, 3)), rng.random((5, 5, 3))
    out["psnr_a"], out["psnr_b"] = a, b
    out["psnr"] = np.array(ru._compute_psnr(a, b, 1.0))
    # .flo round trip through the reference reader
    import fisr_oracle as O
    fl = rng.standard_normal((1, 2, 3, 4, 2)).astype(np.float32)
    tmp = os.path.join(GOLD, "_tmp.flo")
    O.write_flo5(tmp, fl)
    out["flo_in"] = fl
    out["flo_read_by_ref"] = ru.read_flo_file_5dim(tmp)
    with open(tmp, "rb") as f:
        out["flo_bytes"] = np.frombuffer(f.read(), np.uint8)
    os.remove(tmp)
    np.savez_compressed(os.path.join(GOLD, "ref_utils.npz"), **out)
    print("ref_utils.npz:", len(out), "arrays")


def smooth_flow(rng, n, h, w, sigma, amp):
    from scipy.ndimage import gaussian_filter
    f = rng.standard_normal((n, h, w, 2)) * amp
    f = gaussian_filter(f, sigma=(0, sigma, sigma, 0), mode="nearest")
    return (f * (amp / max(f.std(), 1e-9))).astype(np.float32)


def gen_scene_crop():
    from PIL import Image
    import fisr_oracle as O
    files = sorted(glob.glob(os.path.join(REF, "FISR_test_folder", "scene1", "*.png")),
                   key=lambda p: int(p.rsplit("_", 1)[1][:-4]))
    assert len(files) == 5
    frames = np.stack([np.array(Image.open(f))[492:588, 912:1008, :] for f in files])  # [5,96,96,3] u8 YUV
    rng = np.random.default_rng(1234)
    # 8 flows: pair p (frames p,p+1): [p->p+1, p+1->p]; SURVEY 8d cfg1
    flows = smooth_flow(rng, 8, 96, 96, 8.0, 4.0).reshape(1, 8, 96, 96, 2)
    warps = np.zeros((1, 8, 96, 96, 3), np.float32)
    for p in range(4):
        # FISR_for_video_warp_img_with_flo.py:121-128: 1->2 warps frame 2, 2->1 warps frame 1
        warps[0, 2 * p] = O.warp_frame(frames[p + 1], flows[0, 2 * p])
        warps[0, 2 * p + 1] = O.warp_frame(frames[p], flows[0, 2 * p + 1])
    np.savez_compressed(os.path.join(GOLD, "scene1_crop96.npz"), frames=frames, flows=flows, warps=warps)
    print("scene1_crop96.npz", frames.shape, flows.shape, warps.shape)
    return frames, flows, warps


def gen_model(frames, flows, warps):
    import fisr_oracle as O
    import c_oracle as C
    from fisr_amd.weights import synthetic_weights
    W = synthetic_weights(2020)
    blob = C.pack_blob(W)
    rng = np.random.default_rng(77)
    x = rng.random((1, 32, 64, 29)).astype(np.float32)
    x[..., 9:17] = (x[..., 9:17] - 0.5) * 0.4
    taps = {}
    p1, p2, p3 = O.model(x, W, np.float64, taps)
    c1, c2, c3 = C.forward(x, blob, True)
    assert np.abs(p3 - c3).max() < 1e-12
    np.savez_compressed(os.path.join(GOLD, "model_32x64.npz"), x=x, l1=p1, l2=p2, l3=p3,
                        dec3=taps["FISRnet/level_3/dec_out"].astype(np.float32))
    print("model_32x64.npz", p3.shape, float(p3.mean()))
    # cfg1: three windows of the 96x96 crop
    fl = O.merge_seq_dim(flows)           # [1,96,96,16]
    wp = O.merge_seq_dim(warps / np.float32(255.))  # utils.py:51 (/255 in float32)
    outs, ins = [], []
    for s in range(3):
        img9 = np.concatenate([frames[s], frames[s + 1], frames[s + 2]], axis=2)
        inp = O.assemble_input(img9, fl[0, :, :, 4 * s:4 * s + 8], wp[0, :, :, 6 * s:6 * s + 12])
        ins.append(inp.astype(np.float32))
        outs.append(C.forward(inp.astype(np.float32), blob, True)[2][0])
    np.savez_compressed(os.path.join(GOLD, "model_96.npz"), inp=np.concatenate(ins).astype(np.float32),
                        l3=np.stack(outs).astype(np.float32))
    print("model_96.npz", np.stack(outs).shape)


if __name__ == "__main__":
    os.makedirs(GOLD, exist_ok=True)
    gen_ref_utils()
    fr, fl, wp = gen_scene_crop()
    gen_model(fr, fl, wp)
