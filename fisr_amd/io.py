"""On-disk formats either side of the hot path (host-side I/O, no arithmetic of the network).

  .flo   FISR's 5-D flow container: reader utils.py:57-74 (`read_flo_file_5dim`), writer
         FISR_tfoptflow/FISR_for_video_pwcnet_predict_from_img_test.py:57-81 (`write_flow`).
  warp   the reference stores warped frames in a MATLAB v7.3 (HDF5) `.mat` written by
         hdf5storage (warp script :131-136) and reads it with h5py (utils.py:45-54).  h5py is used
         when importable, otherwise fisr_amd/hdf5_min.py (the subset of the HDF5 file format those
         files use); a `.npy` holding the same [N,N_seq,H,W,3] float32 0..255 array is accepted
         everywhere a `.mat` is.
  PNG    frames are YUV packed in 8-bit RGB PNGs (README.md:38-55), read with PIL.
"""
from __future__ import annotations

import os

import numpy as np

FLO_MAGIC = np.float32(202021.25)


def read_flo_file_5dim(filename: str) -> np.ndarray:
    """-> float32 [N, N_seq, h, w, 2].  Unlike the reference (which prints and returns None on a
    bad magic number, utils.py:62-63) this raises ValueError."""
    with open(filename, "rb") as f:
        magic = np.fromfile(f, np.float32, count=1)
        if magic.size != 1 or magic[0] != FLO_MAGIC:
            raise ValueError(f"{filename}: magic number incorrect, invalid .flo file")
        hdr = np.fromfile(f, np.int32, count=4)
        if hdr.size != 4 or (hdr <= 0).any():
            raise ValueError(f"{filename}: bad header {hdr}")
        n, s, h, w = (int(v) for v in hdr)
        data = np.fromfile(f, np.float32, count=n * s * h * w * 2)
    if data.size != n * s * h * w * 2:
        raise ValueError(f"{filename}: truncated, expected {n * s * h * w * 2} floats, got {data.size}")
    return data.reshape(n, s, h, w, 2)


def write_flow(flow: np.ndarray, filename: str) -> None:
    flow = np.ascontiguousarray(flow, np.float32)
    if flow.ndim != 5 or flow.shape[4] != 2:
        raise ValueError("flow must be [N, N_seq, h, w, 2]")
    with open(filename, "wb") as f:
        np.array([FLO_MAGIC], np.float32).tofile(f)
        np.array(flow.shape[:4], np.int32).tofile(f)
        flow.tofile(f)


def read_warp_file(filename: str, key: str = "pred") -> np.ndarray:
    """-> float32 [N, N_seq, H, W, 3] in 0..255 (NOT yet divided by 255; utils.py:51 divides,
    here the pack kernel does, FISRnet.py:839-840)."""
    ext = os.path.splitext(filename)[1].lower()
    if ext == ".npy":
        a = np.load(filename)
    elif ext == ".npz":
        a = np.load(filename)[key]
    else:
        try:
            import h5py
        except ImportError:
            from . import hdf5_min                  # the subset of HDF5 those files use, restated
            a = hdf5_min.read_dataset(filename, key)
        else:
            with h5py.File(filename, "r") as f:
                a = np.array(f[key], dtype=np.float32)
        a = np.transpose(a, (4, 3, 2, 1, 0))            # utils.py:52 (MATLAB dims are reversed on disk)
    a = np.asarray(a, np.float32)
    if a.ndim != 5 or a.shape[4] != 3:
        raise ValueError(f"{filename}: expected [N,N_seq,H,W,3], got {a.shape}")
    return a


def write_warp_file(filename: str, pred: np.ndarray) -> None:
    pred = np.ascontiguousarray(pred, np.float32)
    ext = os.path.splitext(filename)[1].lower()
    if ext == ".npy":
        np.save(filename, pred)
        return
    disk = np.ascontiguousarray(np.transpose(pred, (4, 3, 2, 1, 0)))     # hdf5storage reverses the dimension order
    try:
        import h5py
    except ImportError:
        from . import hdf5_min
        hdf5_min.write_dataset(filename, "pred", disk, chunks=hdf5_min.auto_chunks(disk.shape, 4),
                               compress=7, shuffle=True, checksum=True, matlab=True)
        return
    with h5py.File(filename, "w") as f:
        f.create_dataset("pred", data=disk)


def read_png(path: str) -> np.ndarray:
    from PIL import Image
    a = np.array(Image.open(path))
    if a.ndim != 3 or a.shape[2] != 3 or a.dtype != np.uint8:
        raise ValueError(f"{path}: expected an 8-bit 3-channel PNG (YUV packed as RGB)")
    return a


def write_png(path: str, a: np.ndarray) -> None:
    from PIL import Image
    Image.fromarray(np.ascontiguousarray(a, np.uint8)).save(path)


def merge_seq_dim(data: np.ndarray) -> np.ndarray:
    """utils.py:78-83: [N,S,H,W,C] -> [N,H,W,S*C]."""
    sz = data.shape
    return np.transpose(data, (0, 2, 3, 1, 4)).reshape(sz[0], sz[2], sz[3], sz[1] * sz[4])
