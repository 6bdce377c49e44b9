"""Hand-assembles a TensorFlow checkpoint-V2 index (a LevelDB-format SSTable) + data file from the FORMAT SPECIFICATION -- TEST
INFRASTRUCTURE ONLY; writes tests/golden/leveldb_handmade.{index,data-00000-of-00001,npz}.

Why: fisr_amd/tf_bundle.py's table reader was only ever checked against its own writer ("LevelDB table block /
footer / restart framing ... against bytes written by an independent implementation").  No TensorFlow / LevelDB exists in this image,
so this script IS that implementation: it shares no code with fisr_amd/ (own bit-wise crc32c, own varints, own protobuf bytes) and
follows leveldb/doc/table_format.md + table/{block_builder,format,table_builder}.cc and tensorflow/core/util/tensor_bundle/
tensor_bundle.cc (BundleWriter::Finish) as published, INCLUDING what the reader's own writer never produces:
  * data blocks with the default restart interval 16 -> entries with a non-zero `shared` prefix length, several restart points;
  * many small data blocks (block size 256) and an index block whose keys are SHORTENED separators (FindShortestSeparator /
    FindShortSuccessor), i.e. not equal to any key in the table;
  * one data block stored snappy-compressed (type byte 1, compressed by the real libsnappy 1.1.8 of /opt/conda when present);
  * an empty metaindex block and the 48-byte footer with zero padding of the two varint64 BlockHandles up to 40 bytes.
Run:  python oracle/make_golden_leveldb.py
"""
import ctypes
import os
import struct

import numpy as np

OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden", "leveldb_handmade")
MAGIC = 0xDB4775248B80FB57


def crc32c_bitwise(data: bytes) -> int:
    """CRC-32C (Castagnoli), reflected polynomial 0x82F63B78, one bit at a time."""
    c = 0xFFFFFFFF
    for byte in data:
        c ^= byte
        for _ in range(8):
            c = (c >> 1) ^ (0x82F63B78 & -(c & 1))
    return c ^ 0xFFFFFFFF


assert crc32c_bitwise(b"123456789") == 0xE3069283          # the check value of the CRC catalogue


def masked(c: int) -> int:                                  # leveldb util/crc32c.h Mask()
    return (((c >> 15) | (c << 17)) + 0xA282EAD8) & 0xFFFFFFFF


def varint(v: int) -> bytes:
    out = b""
    while v >= 0x80:
        out += bytes([(v & 0x7F) | 0x80])
        v >>= 7
    return out + bytes([v])


def build_block(items, restart_interval):
    """block_builder.cc: entries (shared, non_shared, value_len varint32 | key delta | value), then fixed32 restart offsets and
    their count.  `shared` is the common prefix with the PREVIOUS key, 0 at a restart point."""
    buf, restarts, last, counter = b"", [], b"", restart_interval
    for key, val in items:
        if counter >= restart_interval:
            restarts.append(len(buf))
            counter, shared = 0, 0
        else:
            shared = 0
            while shared < min(len(last), len(key)) and last[shared] == key[shared]:
                shared += 1
        buf += varint(shared) + varint(len(key) - shared) + varint(len(val)) + key[shared:] + val
        last, counter = key, counter + 1
    if not restarts:
        restarts = [0]
    for r in restarts:
        buf += struct.pack("<I", r)
    return buf + struct.pack("<I", len(restarts))


def shortest_separator(start: bytes, limit: bytes) -> bytes:
    """comparator.cc BytewiseComparatorImpl::FindShortestSeparator"""
    n = min(len(start), len(limit))
    d = 0
    while d < n and start[d] == limit[d]:
        d += 1
    if d >= n:
        return start                                        # one is a prefix of the other: unchanged
    b = start[d]
    if b < 0xFF and b + 1 < limit[d]:
        return start[:d] + bytes([b + 1])
    return start


def short_successor(key: bytes) -> bytes:
    for i, b in enumerate(key):
        if b != 0xFF:
            return key[:i] + bytes([b + 1])
    return key


def snappy_compress(raw: bytes):
    try:
        lib = ctypes.CDLL("/opt/conda/lib/libsnappy.so.1")
    except OSError:
        return None
    lib.snappy_max_compressed_length.restype = ctypes.c_size_t
    lib.snappy_max_compressed_length.argtypes = [ctypes.c_size_t]
    lib.snappy_compress.argtypes = [ctypes.c_char_p, ctypes.c_size_t, ctypes.c_char_p, ctypes.POINTER(ctypes.c_size_t)]
    n = ctypes.c_size_t(lib.snappy_max_compressed_length(len(raw)))
    buf = ctypes.create_string_buffer(n.value)
    assert lib.snappy_compress(raw, len(raw), buf, ctypes.byref(n)) == 0
    return buf.raw[:n.value]


def pb_varint_field(field, v):
    return varint(field << 3) + varint(v)


def pb_bytes_field(field, b):
    return varint((field << 3) | 2) + varint(len(b)) + b


def entry_proto(shape, offset, size, crc):
    """BundleEntryProto (tensor_bundle.proto): dtype = 1 (DT_FLOAT = 1), shape = 2 (TensorShapeProto: repeated Dim dim = 2 {size = 1}),
    shard_id = 3 (0: proto3 omits zeros), offset = 4, size = 5, crc32c = 6 (fixed32)."""
    dims = b"".join(pb_bytes_field(2, pb_varint_field(1, d)) for d in shape)
    out = pb_varint_field(1, 1) + pb_bytes_field(2, dims)
    if offset:
        out += pb_varint_field(4, offset)
    out += pb_varint_field(5, size)
    return out + varint((6 << 3) | 5) + struct.pack("<I", crc)


def main():
    rng = np.random.default_rng(20200406)
    names = []
    for lvl in (1, 2, 3):
        for part, k in (("enc/level_0", 64), ("enc/level_1", 128), ("dec/level_0", 64)):
            for rb in (0, 1):
                for cv in (0, 1):
                    names.append((f"FISRnet/level_{lvl}/{part}/res_block/{rb}/conv/{cv}", k))
    names += [("FISRnet/level_3/SR/conv/2", 3), ("FISRnet/level_3/FI-SR/conv/2", 6)]
    tensors = {}
    for base, co in names:
        tensors[base + "/w"] = rng.standard_normal((3, 3, 2, min(co, 5))).astype(np.float32)
        tensors[base + "/b"] = rng.standard_normal((min(co, 5),)).astype(np.float32)
        # what a training checkpoint also holds (FISRnet.py:232, 490-491): Adam slots under longer names -- skipped by readers
        tensors[base + "/w/Adam"] = np.zeros((3, 3, 2, min(co, 5)), np.float32)
    keys = sorted(tensors)                                  # BundleWriter keeps a std::map: bytewise order
    data, entries, off = b"", [], 0
    for k in keys:
        raw = tensors[k].tobytes()
        entries.append((k.encode(), entry_proto(tensors[k].shape, off, len(raw), masked(crc32c_bitwise(raw)))))
        data += raw
        off += len(raw)
    # BundleHeaderProto {num_shards = 1 (field 1), endianness = 2 (LITTLE = 0: omitted), version = 3 {producer = 1}}
    header = pb_varint_field(1, 1) + pb_bytes_field(3, pb_varint_field(1, 1))
    items = [(b"", header)] + entries

    # ---- table_builder.cc: cut a data block when its size estimate reaches block_size; index entry = separator -> BlockHandle
    file_ = b""
    index_items = []
    block, pending = [], None
    n_blocks = 0
    compressed_one = False

    def flush(block, next_key):
        nonlocal file_, n_blocks, compressed_one
        raw = build_block(block, 16)
        body, ctype = raw, 0
        if n_blocks == 2:                                   # the third data block goes out snappy-compressed, as format.cc may do
            c = snappy_compress(raw)
            if c is not None:
                body, ctype, compressed_one = c, 1, True
        handle = varint(len(file_)) + varint(len(body))
        trailer = bytes([ctype]) + struct.pack("<I", masked(crc32c_bitwise(body + bytes([ctype]))))
        file_ += body + trailer
        last = block[-1][0]
        sep = shortest_separator(last, next_key) if next_key is not None else short_successor(last)
        index_items.append((sep, handle))
        n_blocks += 1

    size = 0
    for i, (k, v) in enumerate(items):
        block.append((k, v))
        size += len(k) + len(v) + 3
        if size >= 256:
            flush(block, items[i + 1][0] if i + 1 < len(items) else None)
            block, size = [], 0
    if block:
        flush(block, None)
    meta = build_block([], 16)
    meta_handle = varint(len(file_)) + varint(len(meta))
    file_ += meta + bytes([0]) + struct.pack("<I", masked(crc32c_bitwise(meta + bytes([0]))))
    index = build_block(index_items, 1)
    index_handle = varint(len(file_)) + varint(len(index))
    file_ += index + bytes([0]) + struct.pack("<I", masked(crc32c_bitwise(index + bytes([0]))))
    footer = meta_handle + index_handle
    footer += bytes(40 - len(footer)) + struct.pack("<Q", MAGIC)
    assert len(footer) == 48
    file_ += footer

    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    with open(OUT + ".index", "wb") as f:
        f.write(file_)
    with open(OUT + ".data-00000-of-00001", "wb") as f:
        f.write(data)
    np.savez_compressed(OUT + ".npz", **{k.replace("/", "|"): v for k, v in tensors.items()})
    print(f"{len(items)} entries in {n_blocks} data blocks (one snappy-compressed: {compressed_one}), index {len(index)} B, "
          f"file {len(file_)} B, data {len(data)} B; index keys: {[s.decode(errors='replace')[-28:] for s, _ in index_items[:4]]} ...")


if __name__ == "__main__":
    main()
