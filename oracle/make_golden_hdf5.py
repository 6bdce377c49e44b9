"""Writes tests/golden/hdf5/*.mat with a REAL HDF5 library -- TEST INFRASTRUCTURE ONLY.

Run with the anaconda interpreter that happens to be in this image (the project's own python has no
h5py):      /opt/conda/bin/python3.9 oracle/make_golden_hdf5.py

The files pin fisr_amd/hdf5_min.py's reader against h5py 3.3.0 / libhdf5 1.10.6.  `expect.npz` holds
the arrays that must come back.  Shapes follow the reference's warp container: a [N,N_seq,H,W,3]
float32 array stored with reversed dimension order under the key 'pred' (warp script :131-136,
utils.py:45-54).
"""
import os
import sys

import h5py
import numpy as np

OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden", "hdf5")


def main():
    os.makedirs(OUT, exist_ok=True)
    rng = np.random.default_rng(11)
    warp = (rng.random((1, 8, 12, 20, 3)) * 255).astype(np.float32)           # [N,N_seq,H,W,3]
    disk = np.ascontiguousarray(np.transpose(warp, (4, 3, 2, 1, 0)))              # what hdf5storage stores
    big = (rng.random((3, 24, 20, 8, 2)) * 255).astype(np.float32)
    with h5py.File(os.path.join(OUT, "contig.mat"), "w") as f:
        f.create_dataset("pred", data=disk)
    with h5py.File(os.path.join(OUT, "chunked.mat"), "w", userblock_size=512) as f:
        d = f.create_dataset("pred", data=disk, chunks=(1, 7, 5, 3, 1), compression="gzip", compression_opts=7,
                             shuffle=True, fletcher32=True)
        d.attrs["MATLAB_class"] = np.bytes_("single")
    with open(os.path.join(OUT, "chunked.mat"), "r+b") as f:                      # the MATLAB 7.3 user block
        head = b"MATLAB 7.3 MAT-file, Platform: posix, Created on: (fixture) HDF5 schema 1.00 ."
        f.write(head.ljust(116, b" ") + b"\0" * 8 + b"\x00\x02" + b"IM")
    with h5py.File(os.path.join(OUT, "auto.mat"), "w") as f:
        f.create_dataset("pred", data=disk, compression="gzip", shuffle=True, fletcher32=True)
        f.create_dataset("other", data=np.arange(10, dtype=np.int16))
        f.create_group("grp").create_dataset("x", data=np.arange(6, dtype=np.float64).reshape(2, 3))
        f.create_dataset("be", data=np.arange(5, dtype=">u4"))
    with h5py.File(os.path.join(OUT, "latest.mat"), "w", libver="latest") as f:
        f.create_dataset("pred", data=disk)
        f.create_dataset("c", data=disk, chunks=disk.shape, compression="gzip")
    with h5py.File(os.path.join(OUT, "manychunks.mat"), "w") as f:               # 3*12*10*4*2 = 2880 chunks: 2-level B-tree
        f.create_dataset("pred", data=big, chunks=(1, 2, 2, 2, 1), shuffle=True, compression="gzip", compression_opts=9)
    np.savez_compressed(os.path.join(OUT, "expect.npz"), warp=warp, big=big)
    print("wrote", sorted(os.listdir(OUT)), "with h5py", h5py.__version__, "hdf5", h5py.version.hdf5_version)


if __name__ == "__main__":
    sys.exit(main())
