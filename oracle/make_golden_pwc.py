"""Fixtures that pin the two scikit-image resizes of the reference's flow script
(FISR_tfoptflow/FISR_for_video_pwcnet_predict_from_img_test.py:129-130, :139) against a real scikit-image.
Run with the anaconda interpreter of this image (scikit-image 0.18.3; the reference pins 0.16.2, same `resize`
semantics for these arguments):      /opt/conda/bin/python3.9 oracle/make_golden_pwc.py
-> tests/golden/pwc_resize.npz  (inputs + skimage outputs; data only)."""
import os

import numpy as np
import skimage
from skimage.transform import resize

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
rng = np.random.default_rng(42)
rgb = (rng.random((12, 20, 3)) * 255).astype(np.float32).astype(np.float64)     # YUV2RGB output: float 0..255
up = resize(rgb, (24, 40))                                                       # script :129-130
flow = rng.standard_normal((2, 24, 40, 2)).astype(np.float32) * 6
down = resize(flow, (2, 12, 20, 2), anti_aliasing=True)                          # script :139 (before the /scale)
np.savez_compressed(os.path.join(ROOT, "tests", "golden", "pwc_resize.npz"), rgb=rgb, up=up, flow=flow, down=down,
                    skimage_version=skimage.__version__)
print("skimage", skimage.__version__, up.shape, down.shape, down.dtype)
