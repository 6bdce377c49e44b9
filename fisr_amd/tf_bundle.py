"""Reader (and a small writer) for TensorFlow checkpoint-V2 "tensor bundles", so that the
reference's `checkpoint_dir/FISRnet_exp1/FISRnet-<step>.{index,data-00000-of-00001}` written by
`tf.train.Saver` (FISRnet.py:585, 1092-1099) loads unchanged (weight seam, FISRnet.py:1101-1115).

Format (restated from TensorFlow 1.13 upstream, tensorflow/core/util/tensor_bundle/ and
tensorflow/core/lib/io/{table,format,block}.cc -- not in the reference tree, and no checkpoint
ships with it, so the TABLE layout is PARITY UNPINNED against a real file (it is round-tripped
against the writer below, which follows the same specification).  The layers underneath are
pinned: crc32c by its check value, the snappy decompressor against blocks compressed by the real
libsnappy 1.1.8 (tests/golden/snappy_blocks.npz, oracle/make_golden_snappy.py), the protobuf wire
coding of BundleEntryProto / TensorShapeProto against google.protobuf in both directions):

  <prefix>.index   an SSTable in LevelDB table format: data blocks of prefix-compressed
                   (shared, non_shared, value_len varint32; key delta; value) entries with a
                   restart array, each block followed by a 1-byte compression type (0 none,
                   1 snappy) and a masked crc32c; an index block; a 48-byte footer holding the
                   metaindex/index BlockHandles and the magic 0xdb4775248b80fb57.
                   key ""   -> BundleHeaderProto {num_shards, endianness, version}
                   key name -> BundleEntryProto {dtype=1, shape=2, shard_id=3, offset=4, size=5,
                                                  crc32c=6 (fixed32)}
  <prefix>.data-SSSSS-of-NNNNN   raw little-endian tensor bytes at [offset, offset+size).
"""
from __future__ import annotations

import os
import struct
from collections import OrderedDict

import numpy as np

TABLE_MAGIC = 0xDB4775248B80FB57
DT_FLOAT, DT_DOUBLE, DT_INT32, DT_INT64 = 1, 2, 3, 9
_DTYPES = {DT_FLOAT: np.float32, DT_DOUBLE: np.float64, DT_INT32: np.int32, DT_INT64: np.int64}

# ------------------------------------------------------------------ crc32c (Castagnoli)
_CRC_TABLE = None


def _crc_table():
    global _CRC_TABLE
    if _CRC_TABLE is None:
        poly = 0x82F63B78
        t = []
        for i in range(256):
            c = i
            for _ in range(8):
                c = (c >> 1) ^ poly if c & 1 else c >> 1
            t.append(c)
        _CRC_TABLE = t
    return _CRC_TABLE


def _crc32c_bytewise(data, c: int) -> int:
    """the table-driven register update over `data`, from register state c to register state (no init / final xor here)"""
    t = _crc_table()
    for b in data:
        c = t[(c ^ b) & 0xFF] ^ (c >> 8)
    return c


_CRC_LANES = 1 << 16           # the big-buffer path runs this many register states in lock-step through numpy
_CRC_BIG = 1 << 20             # ... for buffers of at least this many bytes (a checkpoint's tensors: 580 MB with the Adam slots)


def _crc32c_lanes(buf, c: int) -> int:
    """The same register update for a big buffer.  The update is linear over GF(2) in (state, data) jointly, so the buffer is cut
    into _CRC_LANES chunks of L bytes that advance side by side (one numpy gather per byte position: L steps instead of
    len(buf)), lane 0 starting from the incoming state and the others from 0; the lanes are then folded left to right,
    state <- Z^L(state) ^ lane_k, with Z^L (L zero bytes pushed through the register) as four 256-entry tables.  Bytes behind the
    last whole chunk go through the byte-wise loop."""
    import numpy as np
    a = np.frombuffer(buf, dtype=np.uint8)
    K = _CRC_LANES
    L = a.size // K
    tab = np.asarray(_crc_table(), dtype=np.uint32)
    state = np.zeros(K, dtype=np.uint32)
    state[0] = c
    cols = a[:K * L].reshape(K, L)
    step = max(1, (32 << 20) // (4 * K))                    # transpose a slab of byte positions at a time: <= 32 MB of uint32 (+ 8 MB of bytes)
    for j0 in range(0, L, step):
        slab = np.ascontiguousarray(cols[:, j0:j0 + step].T).astype(np.uint32)
        for row in slab:
            state = tab[(state ^ row) & 0xFF] ^ (state >> 8)
    # Z^L on the 32 basis vectors, all at once
    z = (np.uint32(1) << np.arange(32, dtype=np.uint32)).astype(np.uint32)
    for _ in range(L):
        z = tab[z & 0xFF] ^ (z >> 8)
    zt = []
    for byte in range(4):                                   # zt[byte][v] = Z^L(v << 8 * byte)
        col = [0] * 256
        for v in range(256):
            acc = 0
            for bit in range(8):
                if v >> bit & 1:
                    acc ^= int(z[8 * byte + bit])
            col[v] = acc
        zt.append(col)
    s = int(state[0])
    for k in state[1:].tolist():
        s = zt[0][s & 0xFF] ^ zt[1][(s >> 8) & 0xFF] ^ zt[2][(s >> 16) & 0xFF] ^ zt[3][s >> 24] ^ k
    return _crc32c_bytewise(a[K * L:].tobytes(), s)


def crc32c(data: bytes, crc: int = 0) -> int:
    c = crc ^ 0xFFFFFFFF
    c = _crc32c_lanes(data, c) if len(data) >= _CRC_BIG else _crc32c_bytewise(data, c)
    return c ^ 0xFFFFFFFF


def mask_crc(c: int) -> int:
    return (((c >> 15) | (c << 17)) + 0xA282EAD8) & 0xFFFFFFFF


# ------------------------------------------------------------------ varints / protobuf
def _get_varint(buf, pos):
    r, shift = 0, 0
    while True:
        b = buf[pos]
        pos += 1
        r |= (b & 0x7F) << shift
        if not b & 0x80:
            return r, pos
        shift += 7


def _put_varint(v: int) -> bytes:
    out = bytearray()
    while True:
        b = v & 0x7F
        v >>= 7
        if v:
            out.append(b | 0x80)
        else:
            out.append(b)
            return bytes(out)


def _parse_proto(buf):
    """-> list of (field, wire_type, value) with value int (varint/fixed) or bytes."""
    out, pos = [], 0
    while pos < len(buf):
        tag, pos = _get_varint(buf, pos)
        f, wt = tag >> 3, tag & 7
        if wt == 0:
            v, pos = _get_varint(buf, pos)
        elif wt == 1:
            v = struct.unpack_from("<Q", buf, pos)[0]; pos += 8
        elif wt == 2:
            n, pos = _get_varint(buf, pos)
            v = bytes(buf[pos:pos + n]); pos += n
        elif wt == 5:
            v = struct.unpack_from("<I", buf, pos)[0]; pos += 4
        else:
            raise ValueError(f"unsupported protobuf wire type {wt}")
        out.append((f, wt, v))
    return out


def _parse_entry(buf):
    e = dict(dtype=0, shape=(), shard_id=0, offset=0, size=0, crc32c=0, sliced=False)
    for f, wt, v in _parse_proto(buf):
        if f == 1:
            e["dtype"] = v
        elif f == 2:
            dims = []
            for f2, _, v2 in _parse_proto(v):
                if f2 == 2:  # TensorShapeProto.dim
                    size = 0
                    for f3, _, v3 in _parse_proto(v2):
                        if f3 == 1:
                            size = v3 - (1 << 64) if v3 >= (1 << 63) else v3
                    dims.append(size)
            e["shape"] = tuple(dims)
        elif f == 3:
            e["shard_id"] = v
        elif f == 4:
            e["offset"] = v
        elif f == 5:
            e["size"] = v
        elif f == 6:
            e["crc32c"] = v
        elif f == 7:
            e["sliced"] = True
    return e


# ------------------------------------------------------------------ snappy (decompress only)
def _snappy_decompress(src: bytes) -> bytes:
    try:
        return _snappy_decompress_unchecked(src)
    except IndexError as e:
        raise ValueError("snappy: truncated or corrupt stream") from e


def _snappy_decompress_unchecked(src: bytes) -> bytes:
    n, pos = _get_varint(src, 0)
    out = bytearray()
    while pos < len(src):
        tag = src[pos]; pos += 1
        kind = tag & 3
        if kind == 0:
            ln = tag >> 2
            if ln >= 60:
                nb = ln - 59
                ln = int.from_bytes(src[pos:pos + nb], "little"); pos += nb
            ln += 1
            out += src[pos:pos + ln]; pos += ln
            continue
        if kind == 1:
            ln = ((tag >> 2) & 7) + 4
            off = ((tag >> 5) << 8) | src[pos]; pos += 1
        elif kind == 2:
            ln = (tag >> 2) + 1
            off = src[pos] | (src[pos + 1] << 8); pos += 2
        else:
            ln = (tag >> 2) + 1
            off = int.from_bytes(src[pos:pos + 4], "little"); pos += 4
        for _ in range(ln):
            out.append(out[-off])
    if len(out) != n:
        raise ValueError("snappy: length mismatch")
    return bytes(out)


# ------------------------------------------------------------------ table reading
def _read_block(f, offset, size, verify=True):
    f.seek(offset)
    raw = f.read(size + 5)
    if len(raw) != size + 5:
        raise ValueError("truncated table block")
    body, ctype = raw[:size], raw[size]
    if verify:
        stored = struct.unpack_from("<I", raw, size + 1)[0]
        if mask_crc(crc32c(raw[:size + 1])) != stored:
            raise ValueError("table block checksum mismatch")
    if ctype == 1:
        body = _snappy_decompress(body)
    elif ctype != 0:
        raise ValueError(f"unknown block compression {ctype}")
    return body


def _iter_block(body):
    nrestart = struct.unpack_from("<I", body, len(body) - 4)[0]
    end = len(body) - 4 - 4 * nrestart
    pos, key = 0, b""
    while pos < end:
        shared, pos = _get_varint(body, pos)
        non_shared, pos = _get_varint(body, pos)
        vlen, pos = _get_varint(body, pos)
        key = key[:shared] + bytes(body[pos:pos + non_shared]); pos += non_shared
        val = bytes(body[pos:pos + vlen]); pos += vlen
        yield key, val


def read_index(index_path: str, verify: bool = True):
    """-> (header dict, OrderedDict name -> entry dict)."""
    with open(index_path, "rb") as f:
        f.seek(0, os.SEEK_END)
        fsize = f.tell()
        if fsize < 48:
            raise ValueError(f"{index_path}: too small to be a table")
        f.seek(fsize - 48)
        footer = f.read(48)
        if struct.unpack_from("<Q", footer, 40)[0] != TABLE_MAGIC:
            raise ValueError(f"{index_path}: bad table magic (not a TF checkpoint-V2 index)")
        pos = 0
        _, pos = _get_varint(footer, pos)       # metaindex handle
        _, pos = _get_varint(footer, pos)
        ioff, pos = _get_varint(footer, pos)
        isize, pos = _get_varint(footer, pos)
        index_block = _read_block(f, ioff, isize, verify)
        header, entries = None, OrderedDict()
        for _, handle in _iter_block(index_block):
            boff, p2 = _get_varint(handle, 0)
            bsize, _ = _get_varint(handle, p2)
            for key, val in _iter_block(_read_block(f, boff, bsize, verify)):
                if key == b"":
                    header = {fld: v for fld, _, v in _parse_proto(val)}
                else:
                    entries[key.decode("utf-8")] = _parse_entry(val)
    if header is None:
        raise ValueError(f"{index_path}: no bundle header")
    if header.get(2, 0) != 0:
        raise ValueError("big-endian bundles are not supported")
    return header, entries


def read_bundle(prefix: str, name_filter: str | None = None, verify_crc: bool = False):
    """Read all (matching) float tensors of a checkpoint-V2 bundle -> OrderedDict name -> ndarray."""
    header, entries = read_index(prefix + ".index")
    num_shards = header.get(1, 1)
    out = OrderedDict()
    files = {}
    try:
        for name, e in entries.items():
            if name_filter and name_filter not in name:
                continue
            if e["sliced"]:
                raise ValueError(f"{name}: partitioned variables are not supported")
            if e["dtype"] not in _DTYPES:
                continue
            sid = e["shard_id"]
            if sid not in files:
                files[sid] = open(f"{prefix}.data-{sid:05d}-of-{num_shards:05d}", "rb")
            fh = files[sid]
            fh.seek(e["offset"])
            raw = fh.read(e["size"])
            if len(raw) != e["size"]:
                raise ValueError(f"{name}: data shard truncated")
            if verify_crc and mask_crc(crc32c(raw)) != e["crc32c"]:
                raise ValueError(f"{name}: tensor checksum mismatch")
            out[name] = np.frombuffer(raw, _DTYPES[e["dtype"]]).reshape(e["shape"]).copy()
    finally:
        for fh in files.values():
            fh.close()
    return out


# ------------------------------------------------------------------ writer (tests / export)
def _block_bytes(items, restart_interval=16):
    body = bytearray()
    restarts = []
    last = b""
    for i, (k, v) in enumerate(items):
        if i % restart_interval == 0:
            restarts.append(len(body))
            shared = 0
        else:
            shared = 0
            while shared < min(len(last), len(k)) and last[shared] == k[shared]:
                shared += 1
        body += _put_varint(shared) + _put_varint(len(k) - shared) + _put_varint(len(v)) + k[shared:] + v
        last = k
    if not restarts:
        restarts = [0]
    for r in restarts:
        body += struct.pack("<I", r)
    body += struct.pack("<I", len(restarts))
    return bytes(body)


def _entry_proto(dtype, shape, offset, size, crc):
    dims = b"".join(b"\x12" + _put_varint(len(d)) + d for d in (b"\x08" + _put_varint(int(s)) for s in shape))
    out = b"\x08" + _put_varint(dtype) + b"\x12" + _put_varint(len(dims)) + dims
    if offset:
        out += b"\x20" + _put_varint(offset)
    out += b"\x28" + _put_varint(size) + b"\x35" + struct.pack("<I", crc)
    return out


def write_bundle(prefix: str, tensors, block_size: int = 4096, with_crc: bool = True) -> None:
    """Write `tensors` (name -> ndarray: float32, or an integer type for counters) as a one-shard checkpoint-V2 bundle, plus the
    `checkpoint` state file tf.train.get_checkpoint_state reads (FISRnet.py:1106)."""
    names = sorted(tensors)
    entries = []
    with open(prefix + ".data-00000-of-00001", "wb") as f:
        off = 0
        for n in names:
            a = np.asarray(tensors[n])
            # integer tensors keep their type (the reference's global step `Variable`, FISRnet.py:232, is a DT_INT32 scalar)
            dt, npdt = (DT_INT32, np.int32) if a.dtype.kind in "iu" and a.dtype.itemsize <= 4 else \
                       (DT_INT64, np.int64) if a.dtype.kind in "iu" else (DT_FLOAT, np.float32)
            raw = np.ascontiguousarray(a, npdt).tobytes()
            f.write(raw)
            crc = mask_crc(crc32c(raw)) if with_crc else 0
            entries.append((n.encode(), _entry_proto(dt, np.shape(a), off, len(raw), crc)))
            off += len(raw)
    header = b"\x08\x01" + b"\x1a\x02\x08\x01"       # num_shards=1, version{producer=1}
    items = [(b"", header)] + entries
    blocks, cur, cur_size = [], [], 0
    for k, v in items:
        cur.append((k, v)); cur_size += len(k) + len(v) + 3
        if cur_size >= block_size:
            blocks.append(cur); cur, cur_size = [], 0
    if cur:
        blocks.append(cur)
    with open(prefix + ".index", "wb") as f:
        handles = []

        def emit(body):
            o = f.tell()
            trailer = b"\x00"
            f.write(body + trailer + struct.pack("<I", mask_crc(crc32c(body + trailer))))
            return o, len(body)

        for blk in blocks:
            o, s = emit(_block_bytes(blk))
            handles.append((blk[-1][0], _put_varint(o) + _put_varint(s)))
        mo, ms = emit(_block_bytes([]))
        io_, is_ = emit(_block_bytes(handles, restart_interval=1))
        footer = _put_varint(mo) + _put_varint(ms) + _put_varint(io_) + _put_varint(is_)
        footer += b"\x00" * (40 - len(footer)) + struct.pack("<Q", TABLE_MAGIC)
        f.write(footer)
    with open(os.path.join(os.path.dirname(prefix), "checkpoint"), "w") as f:
        base = os.path.basename(prefix)
        f.write(f'model_checkpoint_path: "{base}"\nall_model_checkpoint_paths: "{base}"\n')
