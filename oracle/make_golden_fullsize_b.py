"""Second full-size oracle fixture (run in the build container; a few minutes of CPU): one reference tile
544x992x29 of a SECOND synthetic scene through the C oracle in float64 with the SURVEY-8d weight set
(`weights.spec_weights(2020)`), committed on a sparse grid as tests/golden/model_544x992_sparse_spec.npz."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, os.path.join(ROOT, "tests"))

import c_oracle as C
from fisr_amd.weights import spec_weights
from tests_support import make_full_size_input

SEED, STRIDE = 9191, 16
x = make_full_size_input(SEED, 544, 992)
blob = C.pack_blob(spec_weights(2020))
t = time.time()
l3 = C.forward(x, blob, double=True)[2]
print("oracle fp64 544x992 (spec weights):", time.time() - t, "s")
np.savez_compressed(os.path.join(ROOT, "tests", "golden", "model_544x992_sparse_spec.npz"),
                    seed=np.int64(SEED), stride=np.int64(STRIDE), weights="spec_weights(2020)",
                    l3_sparse=l3[0, ::STRIDE, ::STRIDE, :])
print(l3.shape, l3[0, ::STRIDE, ::STRIDE, :].shape, float(l3.mean()), float(l3.std()))
