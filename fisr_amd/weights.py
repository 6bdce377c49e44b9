"""FISRnet weight container: variable names, shapes, synthetic weights, file I/O.

The 276 tensors (138 convs x {w, b}) are keyed by the TF variable names the
reference creates (`ops.py:7-9` under the scopes of `FISRnet.py:73-173`, top
scope 'FISRnet' from `FISRnet.py:750`), so a checkpoint written by the
reference's `tf.train.Saver` (`FISRnet.py:585, 751-753`) maps 1:1.

  name  = 'FISRnet/level_{1,2,3}/<block path>/{w,b}'
  w     = [3, 3, Cin, Cout] float32 (HWIO, as `tf.nn.conv2d` takes it, ops.py:10)
  b     = [Cout] float32

Container formats accepted by `load_weights`:
  * `.npz`  - numpy archive keyed by the TF variable names ('/' kept).
  * TF checkpoint-V2 bundle prefix (`FISRnet-<step>.index` + `.data-00000-of-00001`)
    via `fisr_amd.tf_bundle` (format restated from TF upstream; no checkpoint is
    in the reference tree).
"""
from __future__ import annotations

import os
import re
from collections import OrderedDict

import numpy as np

CH = 64          # FISRnet.py:74
IN_CH = 29       # 9 (3 YUV frames) + 8 (4 flows) + 12 (4 warped frames); FISRnet.py:843
LEVELS = ("level_1", "level_2", "level_3")


def level_in_channels(level: str) -> int:
    """level_1 sees the 29-ch input; level_2/3 see input ++ previous prediction (9)
    (FISRnet.py:84, 113-116, 144-147)."""
    return IN_CH if level == "level_1" else IN_CH + 9


def conv_specs_for_level(level: str):
    """[(relative name, Cin, Cout)] in graph-construction order (ops.py:48-76,
    FISRnet.py:83-106)."""
    cin = level_in_channels(level)
    ch = CH
    out = []

    def rb(prefix, c, idx):
        out.append((f"{prefix}/res_block/{idx}/conv/0", c, c))
        out.append((f"{prefix}/res_block/{idx}/conv/1", c, c))

    # encoder: Enc_level_res (ops.py:48-55)
    for li, (c1, c) in enumerate(((cin, ch), (ch, ch * 2), (ch * 2, ch * 4))):
        p = f"enc/level_{li}"
        out.append((f"{p}/conv/0", c1, c))
        rb(p, c, 0)
        rb(p, c, 1)
    # bottleneck (ops.py:59-63)
    out.append(("bottleneck/conv/0", ch * 4, ch * 8))
    rb("bottleneck", ch * 8, 0)
    # decoder: Dec_level_res (ops.py:67-76), called level_2 -> level_0 (FISRnet.py:91-93)
    for li, (c1, c) in ((2, (ch * 8, ch * 4)), (1, (ch * 4, ch * 2)), (0, (ch * 2, ch))):
        p = f"dec/level_{li}"
        out.append((f"{p}/resize", c1, c))
        out.append((f"{p}/conv/0", c * 2, c))
        rb(p, c, 0)
        rb(p, c, 1)
    # heads (FISRnet.py:95-106)
    for head, cout in (("FI-SR", 6), ("SR", 3)):
        out.append((f"{head}/conv/0", ch, ch))
        rb(head, ch, 0)
        out.append((f"{head}/conv/1", ch, ch * 4))
        out.append((f"{head}/conv/2", ch, cout))
    return out


def conv_specs():
    """All 138 convs: [(full conv name, Cin, Cout)], level_1 .. level_3."""
    specs = []
    for lv in LEVELS:
        for rel, ci, co in conv_specs_for_level(lv):
            specs.append((f"FISRnet/{lv}/{rel}", ci, co))
    return specs


def variable_shapes() -> "OrderedDict[str, tuple]":
    d = OrderedDict()
    for name, ci, co in conv_specs():
        d[name + "/w"] = (3, 3, ci, co)
        d[name + "/b"] = (co,)
    return d


def num_parameters() -> int:
    return sum(int(np.prod(s)) for s in variable_shapes().values())


def synthetic_weights(seed: int = 2020, gain: float = 1.0, rb_damp: float = 0.3,
                      head_scale: float = 1.0) -> "OrderedDict[str, np.ndarray]":
    """Seeded stand-in weights (no checkpoint ships with the reference).

    Scaled so activations stay O(1) through 46 convs per level: conv kernels
    ~ N(0, gain^2 * 2/(9*Cin)) (He), the second conv of every residual block damped
    by `rb_damp` (0.3: the identity path dominates; 0.6 lets the decoder activations
    grow to ~1.5 rms, a harsher set for the reduced-precision engines), `conv/2` heads
    centred on 0.5 so that predictions land inside [0,1] like real YUV frames.
    Generated in `conv_specs()` order from one `default_rng(seed)` stream.
    """
    rng = np.random.default_rng(seed)
    out = OrderedDict()
    for name, ci, co in conv_specs():
        std = gain * np.sqrt(2.0 / (9.0 * ci))
        if re.search(r"res_block/\d/conv/1$", name):
            std *= rb_damp
        b_mean, b_std = 0.0, 0.01
        if name.endswith("conv/2"):
            std = head_scale * gain * 0.25 * np.sqrt(1.0 / (9.0 * ci))
            b_mean = 0.5
        out[name + "/w"] = (rng.standard_normal((3, 3, ci, co)) * std).astype(np.float32)
        out[name + "/b"] = (b_mean + rng.standard_normal((co,)) * b_std).astype(np.float32)
    return out


def spec_weights(seed: int = 2020, head_gain: float = 4.0) -> "OrderedDict[str, np.ndarray]":
    """The weight set SURVEY.md 8d specifies: every conv `w ~ N(0, (0.8*sqrt(2/(9*(Ci+Co))))^2)` (Glorot-normal
    x 0.8, the reference's own initialiser family, ops.py:8), `b ~ N(0, 0.01^2)`, NO damping of the residual
    branches, `default_rng(seed)` in `conv_specs()` order.  As the survey allows ("rescale the three conv/2
    heads if not O(1)"), only the conv/2 heads are touched: weights x head_gain and bias + 0.5, so that the
    predictions spread over [0,1] (std ~0.15) instead of sitting at 0 +- 0.03 where the clip would hide errors."""
    rng = np.random.default_rng(seed)
    out = OrderedDict()
    for name, ci, co in conv_specs():
        w = rng.standard_normal((3, 3, ci, co)) * (0.8 * np.sqrt(2.0 / (9.0 * (ci + co))))
        b = rng.standard_normal((co,)) * 0.01
        if name.endswith("conv/2"):
            w = w * head_gain
            b = b + 0.5
        out[name + "/w"] = w.astype(np.float32)
        out[name + "/b"] = b.astype(np.float32)
    return out


# The weight sets the parity tests run on (name -> factory): the benign default, the survey's spec, a harsher one.
WEIGHT_SETS = {
    "default": lambda: synthetic_weights(2020),
    "survey_spec": lambda: spec_weights(2020),
    "harsh": lambda: synthetic_weights(7, 1.0, rb_damp=0.6, head_scale=0.4),
}


def xavier_weights(seed: int = 0) -> "OrderedDict[str, np.ndarray]":
    """The reference's initialiser for a run from scratch (ops.py:8-9): `w` = tf.contrib.layers.xavier_initializer(
    uniform=False) = variance_scaling_initializer(factor=1, mode='FAN_AVG', uniform=False), i.e. a normal truncated at two
    standard deviations with stddev = sqrt(1.3 / ((fan_in + fan_out) / 2)), fan = 9 * channels; `b` = zeros."""
    rng = np.random.default_rng(seed)
    out = OrderedDict()
    for name, ci, co in conv_specs():
        std = np.sqrt(1.3 * 2.0 / (9.0 * (ci + co)))
        w = rng.standard_normal((3, 3, ci, co))
        bad = np.abs(w) > 2.0
        while bad.any():                                     # tf.truncated_normal re-draws the samples beyond 2 sigma
            w[bad] = rng.standard_normal(int(bad.sum()))
            bad = np.abs(w) > 2.0
        out[name + "/w"] = (w * std).astype(np.float32)
        out[name + "/b"] = np.zeros(co, np.float32)
    return out


def check_complete(weights) -> None:
    """Raise KeyError/ValueError unless every one of the 276 tensors is present with
    the reference's shape (what `saver.restore` would enforce, FISRnet.py:1108)."""
    for name, shape in variable_shapes().items():
        if name not in weights:
            raise KeyError(f"missing variable {name}")
        if tuple(weights[name].shape) != tuple(shape):
            raise ValueError(f"{name}: shape {tuple(weights[name].shape)} != {shape}")


def save_npz(path: str, weights) -> None:
    np.savez(path, **{k: np.asarray(v, np.float32) for k, v in weights.items()})


def load_npz(path: str):
    with np.load(path) as z:
        return OrderedDict((k, np.asarray(z[k], np.float32)) for k in z.files)


def find_checkpoint(checkpoint_dir: str, model_dir: str):
    """Locate weights the way `FISRnet.load` does (FISRnet.py:1101-1115):
    `<checkpoint_dir>/<model_dir>/checkpoint` names the latest bundle prefix; the
    training step is the last integer in the file name.  Also accepts
    `<checkpoint_dir>/<model_dir>/FISRnet-<step>.npz` (this build's container).
    Returns (path_or_prefix, kind, step) or (None, None, 0)."""
    d = os.path.join(checkpoint_dir, model_dir)
    if not os.path.isdir(d):
        return None, None, 0
    state = os.path.join(d, "checkpoint")
    if os.path.isfile(state):
        with open(state) as f:
            m = re.search(r'model_checkpoint_path:\s*"([^"]+)"', f.read())
        if m:
            name = os.path.basename(m.group(1))
            prefix = os.path.join(d, name)
            step = int(next(re.finditer(r"(\d+)(?!.*\d)", name)).group(0))
            if os.path.isfile(prefix + ".index"):
                return prefix, "tf_bundle", step
            if os.path.isfile(prefix + ".npz"):
                return prefix + ".npz", "npz", step
    # No usable `checkpoint` state file.  The reference's load() returns False here (FISRnet.py:1113-1115);
    # as a convenience the newest container in the directory is used instead -- by training step (natural
    # order: FISRnet-10 after FISRnet-9), and loudly.
    def step_of(stem):
        m = list(re.finditer(r"(\d+)(?!.*\d)", stem))
        return int(m[0].group(0)) if m else 0

    for suffix, kind in ((".npz", "npz"), (".index", "tf_bundle")):
        cands = sorted((f for f in os.listdir(d) if f.endswith(suffix)),
                       key=lambda f: (step_of(f[:-len(suffix)]), f))
        if cands:
            name = cands[-1]
            stem = name[:-len(suffix)]
            print(f" [!] no 'checkpoint' state file in {d}: falling back to the highest step found, {name}")
            return os.path.join(d, name if kind == "npz" else stem), kind, step_of(stem)
    return None, None, 0


ADAM_SLOTS = ("/Adam", "/Adam_1")          # tf.train.AdamOptimizer's slot variables m and v of `<var>` (FISRnet.py:490-491)


def load_optimizer_state(path_or_prefix: str, kind: str | None = None):
    """The Adam state a training checkpoint holds next to the weights -- `<var>/Adam`, `<var>/Adam_1` for all 276
    variables and the scalars `beta1_power`, `beta2_power` (the reference creates its tf.train.Saver after build_model,
    FISRnet.py:585, so these are saved and restored with the weights).  None when the file holds weights only."""
    if kind is None:
        kind = "npz" if path_or_prefix.endswith(".npz") else "tf_bundle"
    if kind == "npz":
        with np.load(path_or_prefix) as z:
            names = set(z.files)
            want = [v + s for v in variable_shapes() for s in ADAM_SLOTS]
            if not all(n in names for n in want) or "beta1_power" not in names or "beta2_power" not in names:
                return None
            out = OrderedDict((n, np.asarray(z[n], np.float32)) for n in want)
            out["beta1_power"], out["beta2_power"] = np.float64(z["beta1_power"]), np.float64(z["beta2_power"])
            return out
    from . import tf_bundle
    try:
        w = tf_bundle.read_bundle(path_or_prefix)
    except (OSError, ValueError, KeyError):
        return None
    want = [v + s for v in variable_shapes() for s in ADAM_SLOTS]
    if not all(n in w for n in want) or "beta1_power" not in w or "beta2_power" not in w:
        return None
    out = OrderedDict((n, np.asarray(w[n], np.float32)) for n in want)
    out["beta1_power"], out["beta2_power"] = np.float64(w["beta1_power"]), np.float64(w["beta2_power"])
    return out


def load_weights(path_or_prefix: str, kind: str | None = None):
    if kind is None:
        kind = "npz" if path_or_prefix.endswith(".npz") else "tf_bundle"
    if kind == "npz":
        w = load_npz(path_or_prefix)
    else:
        from . import tf_bundle
        w = tf_bundle.read_bundle(path_or_prefix, name_filter="FISRnet")
    # a training checkpoint also holds Adam slots etc. (FISRnet.py:490-491); keep ours only
    shapes = variable_shapes()
    w = OrderedDict((k, np.ascontiguousarray(w[k], np.float32)) for k in shapes if k in w)
    check_complete(w)
    return w
