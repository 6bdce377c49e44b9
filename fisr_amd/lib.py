"""ctypes binding of libfisr_hip.so (include/fisr.h) and its in-tree build.

The product path has NO fallback: if the HIP library is missing or a call fails, an
exception is raised (FisrError) -- nothing here ever routes through `oracle/` or a
PyTorch/CPU implementation.
"""
from __future__ import annotations

import ctypes
import os
import subprocess
from ctypes import POINTER, c_char_p, c_double, c_float, c_int, c_int64, c_size_t, c_uint8, c_void_p

_PKG = os.path.dirname(os.path.abspath(__file__))
_ROOT = os.path.dirname(_PKG)
SO_PATH = os.environ.get("FISR_HIP_SO") or os.path.join(_PKG, "libfisr_hip.so")   # override: A/B kernel builds
CSRC = os.path.join(_PKG, "csrc")

PREC_F32, PREC_F16, PREC_BF16X3, PREC_F16F8, PREC_F32W, PREC_MIXED, PREC_F16R, PREC_MIXEDR, PREC_F32W4, PREC_F16F8R = 0, 1, 2, 3, 4, 5, 6, 7, 8, 9
CONV_RELU_IN, CONV_RELU_OUT, CONV_D2S, CONV_UP2_IN = 1, 2, 4, 8

EXPORTS = [
    "fisr_version", "fisr_create", "fisr_destroy", "fisr_last_error", "fisr_set_weight",
    "fisr_finalize_weights", "fisr_num_variables_set", "fisr_workspace_bytes", "fisr_forward", "fisr_forward_frames",
    "fisr_profile_enable", "fisr_profile_reset", "fisr_profile_read", "fisr_warp", "fisr_pack_input",
    "fisr_unpack_output", "fisr_stitch", "fisr_sse_vs_u8", "fisr_ssim_u8", "fisr_op_conv3x3", "fisr_op_conv3x3_pool", "fisr_op_maxpool2", "fisr_op_prep_level_input",
    "fisr_op_upsample2", "fisr_bench_conv",
    "fisr_comm_unique_id", "fisr_comm_init", "fisr_comm_rank", "fisr_comm_size", "fisr_comm_allgather",
    "fisr_comm_sendrecv", "fisr_comm_destroy",
    "fisr_pwc_create", "fisr_pwc_destroy", "fisr_pwc_last_error", "fisr_pwc_num_variables", "fisr_pwc_variable",
    "fisr_pwc_set_weight", "fisr_pwc_finalize", "fisr_pwc_finalize_precision", "fisr_pwc_flow_stack_workspace_bytes", "fisr_pwc_flow_stack",
    "fisr_pwc_flow_workspace_bytes", "fisr_pwc_flow_pair",
    "fisr_pwc_nn_workspace_bytes", "fisr_pwc_nn", "fisr_pwc_prep", "fisr_pwc_flow_out",
    "fisr_pwc_op_conv", "fisr_pwc_op_deconv", "fisr_pwc_op_costvol", "fisr_pwc_op_warp",
    "fisr_train_packed_bytes", "fisr_train_pack", "fisr_train_wino_bytes", "fisr_train_pack_wino", "fisr_train_pack_all", "fisr_train_conv3x3", "fisr_train_wgrad", "fisr_train_bgrad",
    "fisr_train_relu_bwd", "fisr_train_axpy", "fisr_train_maxpool2_bwd", "fisr_train_upsample2_bwd", "fisr_train_s2d",
    "fisr_train_copy_channels", "fisr_train_loss", "fisr_train_adam",
]
COMM_ID_BYTES = 128


class FisrError(RuntimeError):
    pass


def hipcc_path() -> str:
    for p in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if p and (os.path.isabs(p) and os.path.isfile(p) or not os.path.isabs(p)):
            return p
    return "hipcc"


def sources():
    out = []
    for d, _, files in sorted(os.walk(CSRC)):
        out += [os.path.join(d, f) for f in sorted(files)]
    return out + [os.path.join(_ROOT, "include", "fisr.h")]


def source_hash() -> str:
    """sha256 (16 hex digits) over csrc/ and include/fisr.h: compiled into the library (`fisr_version()`), printed by bench.py."""
    import hashlib
    h = hashlib.sha256()
    for p in sources():
        h.update(os.path.relpath(p, _ROOT).encode() + b"\0")
        with open(p, "rb") as f:
            h.update(f.read())
    return h.hexdigest()[:16]


def needs_build() -> bool:
    if not os.path.isfile(SO_PATH):
        return True
    t = os.path.getmtime(SO_PATH)
    return any(os.path.getmtime(s) > t for s in sources())


def build(force: bool = False, verbose: bool = False, diag: bool = False, defines=(), out: str | None = None) -> str:
    """hipcc --offload-arch=gfx950 -> fisr_amd/libfisr_hip.so (in-tree; cross-compiles without a GPU).
    diag=True: the diagnostics build (-DFISR_DIAG: environment switches, superseded kernels, ablations, traces) ->
    build_ab/libfisr_hip_diag.so, loaded through FISR_HIP_SO by the scripts under scripts/; never the product."""
    target = out or (os.path.join(_ROOT, "build_ab", "libfisr_hip_diag.so") if diag else SO_PATH)
    if not diag and not out and not force and not needs_build():
        return SO_PATH
    os.makedirs(os.path.dirname(target), exist_ok=True)
    cmd = [hipcc_path(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-fvisibility=hidden",
           "-Wl,--version-script=" + os.path.join(CSRC, "exports.map"),
           f'-DFISR_SRC_HASH="{source_hash()}"'] + (["-DFISR_DIAG"] if diag else []) + [f"-D{d}" for d in defines] + \
          ["-o", target, os.path.join(CSRC, "fisr_api.hip")]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return target


MAX_SRC_ITEMS = 16          # FISR_MAX_SRC_ITEMS (include/fisr.h)


class SrcItem(ctypes.Structure):
    """fisr_src_item (include/fisr.h): one tile of one window for fisr_forward_frames."""
    _fields_ = [("frames", c_void_p * 3), ("flows", c_void_p * 4), ("warps", c_void_p * 4), ("y0", c_int), ("x0", c_int)]


_lib = None


def lib():
    """Load the library (never builds implicitly on a GPU box: the .so ships in-tree)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.isfile(SO_PATH):
        raise FisrError(f"{SO_PATH} not found: run `python -c 'import __graft_entry__ as g; g.build()'` "
                        "(hipcc --offload-arch=gfx950). There is no CPU fallback.")
    L = ctypes.CDLL(SO_PATH)
    vp = c_void_p
    L.fisr_version.restype = c_char_p
    L.fisr_create.argtypes = [POINTER(vp), c_int]
    L.fisr_destroy.argtypes = [vp]
    L.fisr_destroy.restype = None
    L.fisr_last_error.argtypes = [vp]
    L.fisr_last_error.restype = c_char_p
    L.fisr_set_weight.argtypes = [vp, c_char_p, POINTER(c_float), POINTER(c_int64), c_int]
    L.fisr_finalize_weights.argtypes = [vp, c_int]
    L.fisr_num_variables_set.argtypes = [vp]
    L.fisr_workspace_bytes.argtypes = [vp, c_int, c_int, c_int]
    L.fisr_workspace_bytes.restype = c_size_t
    L.fisr_forward.argtypes = [vp, vp, c_int, c_int, c_int, vp, vp, vp, vp, c_size_t, vp]
    L.fisr_forward_frames.argtypes = [vp, POINTER(SrcItem), c_int, c_int, c_int, c_int, c_int, vp, vp, vp, vp, c_size_t, vp]
    L.fisr_profile_enable.argtypes = [vp, c_int]
    L.fisr_profile_reset.argtypes = [vp]
    L.fisr_profile_read.argtypes = [vp, c_int, POINTER(c_char_p), POINTER(c_double), POINTER(c_int64),
                                    POINTER(c_double), POINTER(c_double)]
    L.fisr_warp.argtypes = [vp, vp, c_float, c_int, c_int, c_int, vp, vp]
    L.fisr_pack_input.argtypes = [POINTER(vp), POINTER(vp), POINTER(vp), c_int, c_int, c_int, c_int, vp, vp]
    L.fisr_unpack_output.argtypes = [vp, c_int, c_int, vp, vp, vp]
    L.fisr_stitch.argtypes = [vp, c_int, c_int, c_int, c_int, c_int, c_int, vp, c_int, c_int, c_int, c_int, vp]
    L.fisr_sse_vs_u8.argtypes = [vp, vp, c_size_t, POINTER(c_double), vp]
    L.fisr_op_conv3x3.argtypes = [vp, c_int, vp, c_int, POINTER(c_float), POINTER(c_float), c_int, vp, vp,
                                  c_int, c_int, c_int, c_int, c_int, c_int, vp]
    L.fisr_op_conv3x3_pool.argtypes = [vp, c_int, vp, c_int, POINTER(c_float), POINTER(c_float), c_int, vp, vp, vp,
                                       c_int, c_int, c_int, c_int, c_int, vp]
    L.fisr_op_maxpool2.argtypes = [vp, vp, c_int, c_int, c_int, c_int, c_int, vp]
    L.fisr_op_upsample2.argtypes = [vp, vp, c_int, c_int, c_int, c_int, c_int, vp]
    L.fisr_op_prep_level_input.argtypes = [vp, vp, vp, c_int, c_int, c_int, c_int, c_int, vp]
    L.fisr_ssim_u8.argtypes = [vp, vp, c_int, c_int, c_int, c_int, POINTER(c_double), vp]
    L.fisr_bench_conv.argtypes = [c_int] * 9 + [POINTER(c_double)]
    L.fisr_comm_unique_id.argtypes = [vp]
    L.fisr_comm_init.argtypes = [POINTER(vp), vp, c_int, c_int, c_int]
    L.fisr_comm_rank.argtypes = [vp]
    L.fisr_comm_size.argtypes = [vp]
    L.fisr_comm_allgather.argtypes = [vp, vp, vp, c_size_t, vp]
    L.fisr_comm_sendrecv.argtypes = [vp, vp, vp, c_size_t, c_int, vp]
    L.fisr_comm_destroy.argtypes = [vp]
    L.fisr_comm_destroy.restype = None
    L.fisr_pwc_create.argtypes = [POINTER(vp), c_int]
    L.fisr_pwc_destroy.argtypes = [vp]
    L.fisr_pwc_destroy.restype = None
    L.fisr_pwc_last_error.argtypes = [vp]
    L.fisr_pwc_last_error.restype = c_char_p
    L.fisr_pwc_variable.argtypes = [c_int, POINTER(c_char_p), POINTER(c_int64)]
    L.fisr_pwc_set_weight.argtypes = [vp, c_char_p, POINTER(c_float), POINTER(c_int64), c_int]
    L.fisr_pwc_finalize.argtypes = [vp]
    L.fisr_pwc_flow_workspace_bytes.argtypes = [vp, c_int, c_int]
    L.fisr_pwc_flow_workspace_bytes.restype = c_size_t
    L.fisr_pwc_flow_pair.argtypes = [vp, vp, vp, c_int, c_int, vp, vp, vp, c_size_t, vp]
    L.fisr_pwc_nn_workspace_bytes.argtypes = [vp, c_int, c_int]
    L.fisr_pwc_nn_workspace_bytes.restype = c_size_t
    L.fisr_pwc_nn.argtypes = [vp, vp, c_int, c_int, vp, POINTER(vp), vp, c_size_t, vp]
    L.fisr_pwc_prep.argtypes = [vp, c_int, c_int, vp, c_int, c_int, vp]
    L.fisr_pwc_flow_out.argtypes = [vp, c_int, c_int, vp, c_int, c_int, vp]
    L.fisr_pwc_op_conv.argtypes = [vp, c_int, c_int, c_int, POINTER(c_float), POINTER(c_float), c_int, c_int, POINTER(c_int), vp, c_int,
                                   c_int, c_int, vp, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_float, c_int, c_int, vp]
    L.fisr_pwc_op_deconv.argtypes = [vp, c_int, c_int, c_int, c_int, POINTER(c_float), POINTER(c_float), c_int, POINTER(c_int), vp, c_int,
                                     c_int, c_int, c_int, c_int, c_int, vp]
    L.fisr_pwc_op_costvol.argtypes = [vp, vp, c_int, vp, c_int, c_int, c_int, c_int, c_int, c_int, vp]
    L.fisr_pwc_op_warp.argtypes = [vp, c_int, vp, c_int, c_int, c_float, vp, c_int, c_int, c_int, c_int, vp]
    L.fisr_pwc_finalize_precision.argtypes = [vp, c_int]
    L.fisr_pwc_flow_stack_workspace_bytes.argtypes = [vp, c_int, c_int, c_int]
    L.fisr_pwc_flow_stack_workspace_bytes.restype = c_size_t
    L.fisr_pwc_flow_stack.argtypes = [vp, POINTER(vp), c_int, c_int, c_int, vp, vp, c_size_t, vp]
    L.fisr_train_packed_bytes.argtypes = [c_int, c_int, c_int]
    L.fisr_train_packed_bytes.restype = c_size_t
    L.fisr_train_pack.argtypes = [vp, c_int, c_int, c_int, vp, vp]
    L.fisr_train_pack_all.argtypes = [vp, c_int, vp]
    L.fisr_train_wino_bytes.argtypes = [c_int, c_int, c_int]
    L.fisr_train_wino_bytes.restype = c_size_t
    L.fisr_train_pack_wino.argtypes = [vp, c_int, c_int, c_int, vp, vp]
    L.fisr_train_conv3x3.argtypes = [vp, c_int, vp, c_int, vp, vp, c_int, vp, vp, c_int, c_int, c_int, c_int,
                                     c_int, c_int, c_int, c_int, vp, vp]
    L.fisr_train_wgrad.argtypes = [vp, c_int, vp, c_int, c_int, vp, c_int, vp, vp, c_int, c_int, c_int, c_int, c_int, vp]
    L.fisr_train_bgrad.argtypes = [vp, c_int, c_size_t, vp, c_int, vp]
    L.fisr_train_relu_bwd.argtypes = [vp, vp, vp, c_size_t, vp]
    L.fisr_train_axpy.argtypes = [vp, c_float, vp, c_size_t, vp]
    L.fisr_train_maxpool2_bwd.argtypes = [vp, vp, vp, c_int, c_int, c_int, c_int, vp]
    L.fisr_train_upsample2_bwd.argtypes = [vp, vp, c_int, c_int, c_int, c_int, vp]
    L.fisr_train_s2d.argtypes = [vp, vp, c_int, c_int, c_int, c_int, vp]
    L.fisr_train_copy_channels.argtypes = [vp, c_int, c_int, vp, c_int, c_int, c_int, c_size_t, c_int, vp]
    L.fisr_train_loss.argtypes = [POINTER(vp), vp, POINTER(vp), vp, c_size_t, POINTER(c_float), vp]
    L.fisr_train_adam.argtypes = [vp, vp, vp, vp, c_size_t, c_float, c_float, c_float, c_float, vp]
    _lib = L
    return L


def check(rc: int, ctx=None):
    if rc < 0:
        msg = lib().fisr_last_error(ctx).decode("utf-8", "replace")
        raise FisrError(f"libfisr_hip error {rc}: {msg}")
    return rc
