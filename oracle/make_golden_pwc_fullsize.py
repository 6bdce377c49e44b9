"""Full-size oracle fixture of the optical-flow path (run ONCE in the build container; ~20-40 minutes of CPU, ~25 GB):
one 1080x1920 frame pair through `compute_flow_pair` of oracle/pwcnet_oracle.py in float64 -- YUV->RGB, x2 up-resize,
PWC-Net-large at 2176x3840 in both directions (levels of 544x960 .. 34x60: every dilated context layer sees sub-images of
many 8x32 tiles), x4 bilinear, anti-aliased down-resize -- committed on a sparse grid (every 8th LR pixel) together with
the refined level-2 .. level-6 flows of direction a->b on the same kind of grid:
    tests/golden/pwc_flow_1080p_sparse.npz
Inputs are regenerated from the seed by tests/tests_support.make_flow_frames; a CRC of the frames is stored so that a
drifting generator is noticed.  Weights: the seeded stand-in set (the PWC-Net checkpoint is not in the reference tree)."""
import os
import sys
import time
import zlib

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, os.path.join(ROOT, "tests"))

import pwcnet_oracle as P
from tests_support import make_flow_frames

SEED, STRIDE, H, W_, GAIN = 77, 8, 1080, 1920, 3.0
fa, fb = make_flow_frames(SEED, H, W_)
Wt = P.synthetic_weights(595000, flow_gain=GAIN)
torch.set_num_threads(os.cpu_count())
t0 = time.time()
imgs = [np.array(P.resize_up2_skimage(P.yuv2rgb(np.asarray(f, np.float32))), dtype=np.uint8) for f in (fa, fb)]
out, pyr_ab = [], None
for d in range(2):                                   # one direction at a time: half the peak memory
    pair = np.stack([imgs[d], imgs[1 - d]])[None].astype(np.float32) / 255.0
    Hh, Ww = pair.shape[2:4]
    pair = np.pad(pair, [(0, 0), (0, 0), (0, (-Hh) % 64), (0, (-Ww) % 64), (0, 0)])
    with torch.no_grad():
        flow, pyr = P.nn(torch.from_numpy(pair).double(), Wt)
    if d == 0:
        pyr_ab = [p_[0].permute(1, 2, 0).numpy() for p_ in pyr]          # flow6 .. flow2, [h_l, w_l, 2]
    out.append(P.resize_down2_skimage_aa(flow.numpy()[:, :Hh, :Ww])[0] / 2.0)
    print(f"direction {d}: {time.time() - t0:.0f} s", flush=True)
flows = np.stack(out)                                # [2, 1080, 1920, 2] LR pixels
np.savez_compressed(os.path.join(ROOT, "tests", "golden", "pwc_flow_1080p_sparse.npz"),
                    seed=np.int64(SEED), stride=np.int64(STRIDE), flow_gain=np.float64(GAIN),
                    frames_crc=np.int64(zlib.crc32(fa.tobytes() + fb.tobytes())),
                    flow_sparse=flows[:, ::STRIDE, ::STRIDE, :].astype(np.float32),
                    flow_absmax=np.float64(np.abs(flows).max()),
                    **{f"flow{6 - k}_ab_sparse": p_[::(2 if k > 2 else 1), ::(2 if k > 2 else 1)].astype(np.float32) for k, p_ in enumerate(pyr_ab)})
print("done", flows.shape, "max |flow|", np.abs(flows).max(), f"{time.time() - t0:.0f} s")
