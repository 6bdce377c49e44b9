"""ctypes loader for the C oracle (oracle/fisr_oracle.c) -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.
"""
from __future__ import annotations

import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# FISR_ORACLE_SO: another build of the same source (oracle/Makefile `asan`: AddressSanitizer + UBSan, `make -C oracle asan-test`)
_SO = os.environ.get("FISR_ORACLE_SO") and os.path.abspath(os.environ["FISR_ORACLE_SO"]) or os.path.join(_HERE, "_build", "libfisr_oracle.so")
_lib = None


def build(force: bool = False) -> str:
    src = os.path.join(_HERE, "fisr_oracle.c")
    if os.environ.get("FISR_ORACLE_SO"):
        return _SO                          # (an explicitly chosen build is never rebuilt from here)
    if force or not os.path.isfile(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-B" if force else "-s"],
                              stdout=subprocess.DEVNULL)
    return _SO


def usable_cores() -> int:
    """Cores this process may actually use: scheduler affinity capped by the cgroup CPU quota.  OpenMP's default is one thread per
    LOGICAL cpu of the host -- 128 threads on the GPU box, whose container has a quota of 16 cores: the checker then spends its
    time in the scheduler (the ten-scene fixture of tests/test_gpu_harness.py took 209 s of the GPU suite that way)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            with open(path) as f:
                parts = f.read().split()
            if path.endswith("cpu.max"):
                if parts[0] != "max":
                    n = min(n, max(1, int(int(parts[0]) / int(parts[1]))))
            else:
                q = int(parts[0])
                if q > 0:
                    with open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as f2:
                        n = min(n, max(1, int(q / int(f2.read()))))
            break
        except (OSError, ValueError, IndexError):
            continue
    return max(1, n)


def lib():
    global _lib
    if _lib is None:
        if not os.path.isfile(_SO):
            build()
        _lib = ctypes.CDLL(_SO)
        fp = ctypes.POINTER(ctypes.c_float)
        dp = ctypes.POINTER(ctypes.c_double)
        _lib.fisr_oracle_forward_f32.argtypes = [fp, ctypes.c_int, ctypes.c_int, ctypes.c_int, fp, fp, fp, fp]
        _lib.fisr_oracle_forward_f64.argtypes = [fp, ctypes.c_int, ctypes.c_int, ctypes.c_int, fp, dp, dp, dp]
        _lib.fisr_oracle_conv3x3_f32.argtypes = [fp] + [ctypes.c_int] * 4 + [fp, fp, ctypes.c_int, ctypes.c_int, fp]
        _lib.fisr_oracle_conv3x3_f64.argtypes = [dp] + [ctypes.c_int] * 4 + [fp, fp, ctypes.c_int, ctypes.c_int, dp]
        _lib.fisr_oracle_set_threads.argtypes = [ctypes.c_int]
        _lib.fisr_oracle_set_threads(usable_cores())       # (not OpenMP's one-per-logical-cpu default: see usable_cores)
    return _lib


def pack_blob(weights) -> np.ndarray:
    """Weight blob in conv_specs() order: w (HWIO) then b per conv, float32."""
    from fisr_amd.weights import conv_specs
    parts = []
    for name, _, _ in conv_specs():
        parts.append(np.ascontiguousarray(weights[name + "/w"], np.float32).ravel())
        parts.append(np.ascontiguousarray(weights[name + "/b"], np.float32).ravel())
    return np.concatenate(parts)


def _ptr(a, t):
    return a.ctypes.data_as(ctypes.POINTER(t))


def forward(inp, blob, double: bool = True, threads: int = 0):
    """inp [N,H,W,29] -> (pred_l1, pred_l2, pred_l3) via FISRnet.py:73-173 restated in C."""
    L = lib()
    if threads:
        L.fisr_oracle_set_threads(threads)
    inp = np.ascontiguousarray(inp, np.float32)
    n, h, w, c = inp.shape
    assert c == 29
    dt, ct = (np.float64, ctypes.c_double) if double else (np.float32, ctypes.c_float)
    l1 = np.empty((n, h // 2, w // 2, 9), dt)
    l2 = np.empty((n, h, w, 9), dt)
    l3 = np.empty((n, 2 * h, 2 * w, 9), dt)
    fn = L.fisr_oracle_forward_f64 if double else L.fisr_oracle_forward_f32
    rc = fn(_ptr(inp, ctypes.c_float), n, h, w, _ptr(blob, ctypes.c_float),
            _ptr(l1, ct), _ptr(l2, ct), _ptr(l3, ct))
    if rc != 0:
        raise ValueError("fisr_oracle_forward: bad arguments")
    return l1, l2, l3


def conv3x3(x, w, b, relu_in=False, double=True):
    L = lib()
    dt, ct = (np.float64, ctypes.c_double) if double else (np.float32, ctypes.c_float)
    x = np.ascontiguousarray(x, dt)
    n, h, wd, ci = x.shape
    co = w.shape[3]
    w = np.ascontiguousarray(w, np.float32)
    b = np.ascontiguousarray(b, np.float32)
    y = np.empty((n, h, wd, co), dt)
    fn = L.fisr_oracle_conv3x3_f64 if double else L.fisr_oracle_conv3x3_f32
    fn(_ptr(x, ct), n, h, wd, ci, _ptr(w, ctypes.c_float), _ptr(b, ctypes.c_float), co, int(relu_in), _ptr(y, ct))
    return y
