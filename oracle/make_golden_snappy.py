"""Writes tests/golden/snappy_blocks.npz with a REAL snappy library -- TEST INFRASTRUCTURE ONLY.

Run with the anaconda interpreter of this image (libsnappy 1.1.8 lives in /opt/conda/lib; the project's own
python has no snappy binding):      /opt/conda/bin/python3.9 oracle/make_golden_snappy.py

TensorFlow's checkpoint index (an SSTable) may snappy-compress its blocks (tensorflow/core/lib/io/format.cc);
the fixtures pin fisr_amd/tf_bundle.py's decompressor: literals of every length class, 1-/2-byte-offset
copies, overlapping copies (run-length), and a block that looks like a real index block.
"""
import ctypes
import os

import numpy as np

OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden", "snappy_blocks.npz")


def main():
    lib = ctypes.CDLL("/opt/conda/lib/libsnappy.so.1")
    lib.snappy_max_compressed_length.restype = ctypes.c_size_t
    lib.snappy_max_compressed_length.argtypes = [ctypes.c_size_t]
    lib.snappy_compress.argtypes = [ctypes.c_char_p, ctypes.c_size_t, ctypes.c_char_p, ctypes.POINTER(ctypes.c_size_t)]

    def compress(raw: bytes) -> bytes:
        n = ctypes.c_size_t(lib.snappy_max_compressed_length(len(raw)))
        buf = ctypes.create_string_buffer(n.value)
        assert lib.snappy_compress(raw, len(raw), buf, ctypes.byref(n)) == 0
        return buf.raw[:n.value]

    rng = np.random.default_rng(3)
    names = [f"FISRnet/level_{l}/{part}/level_{k}/rb/{r}/conv/{c}/{v}".encode()
             for l in (1, 2, 3) for part in ("enc", "dec") for k in range(3) for r in range(2) for c in range(2) for v in ("w", "b")]
    index_like = b"".join(bytes([0, len(n), 12]) + n + rng.integers(0, 256, 12, dtype=np.uint8).tobytes() for n in names)
    cases = {
        "empty": b"",
        "one": b"x",
        "short_literal": b"hello snappy",
        "literal_61": bytes(range(61)),                      # length needs one extra byte
        "literal_300": rng.integers(0, 256, 300, dtype=np.uint8).tobytes(),
        "literal_70000": rng.integers(0, 256, 70000, dtype=np.uint8).tobytes(),   # length needs three bytes, 2 blocks
        "run": b"a" * 1000,                                  # overlapping copy, offset 1
        "period7": b"abcdefg" * 500,
        "far_copy": rng.integers(0, 256, 3000, dtype=np.uint8).tobytes() * 3,     # 2-byte offsets
        "index_like": index_like,
        "text": (b"the quick brown fox jumps over the lazy dog; " * 40) + bytes(range(256)) * 4,
    }
    out = {}
    for k, raw in cases.items():
        out[k + "_raw"] = np.frombuffer(raw, np.uint8)
        out[k + "_snappy"] = np.frombuffer(compress(raw), np.uint8)
        print(k, len(raw), "->", len(out[k + "_snappy"]))
    np.savez_compressed(OUT, **out)


if __name__ == "__main__":
    main()
