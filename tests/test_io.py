"""CPU tests of the on-disk formats and the CLI surface (host logic, no GPU)."""
import os

import numpy as np
import pytest

from fisr_amd import io as fio
from fisr_amd import main as fmain
from fisr_amd import splitfmt, tf_bundle, weights


def test_flo_roundtrip_and_reference_bytes(tmp_path, gold_dir):
    g = np.load(os.path.join(gold_dir, "ref_utils.npz"))
    p = str(tmp_path / "a.flo")
    fio.write_flow(g["flo_in"], p)
    with open(p, "rb") as f:
        assert np.array_equal(np.frombuffer(f.read(), np.uint8), g["flo_bytes"])   # same bytes as the oracle writer
    assert np.array_equal(fio.read_flo_file_5dim(p), g["flo_read_by_ref"])        # == what the reference reader returns
    with open(p, "r+b") as f:
        f.write(b"\x01\x02\x03\x04")
    with pytest.raises(ValueError):
        fio.read_flo_file_5dim(p)
    fio.write_flow(g["flo_in"], p)
    with open(p, "r+b") as f:
        f.truncate(40)
    with pytest.raises(ValueError):
        fio.read_flo_file_5dim(p)
    with pytest.raises(ValueError):
        fio.write_flow(np.zeros((2, 3, 4)), p)


def test_warp_file_npy_and_merge(tmp_path, gold_dir):
    g = np.load(os.path.join(gold_dir, "ref_utils.npz"))
    a = np.random.default_rng(0).random((1, 2, 5, 6, 3)).astype(np.float32) * 255
    p = str(tmp_path / "w.npy")
    fio.write_warp_file(p, a)
    assert np.array_equal(fio.read_warp_file(p), a)
    with pytest.raises(ValueError):
        np.save(p, a[..., :2]); fio.read_warp_file(p)
    assert np.array_equal(fio.merge_seq_dim(g["seq_in"]), g["merge_seq"])
    # .mat (HDF5) goes through h5py when importable, else through the restated subset
    pm = str(tmp_path / "w.mat")
    fio.write_warp_file(pm, a)
    assert np.array_equal(fio.read_warp_file(pm), a)
    with pytest.raises((ValueError, OSError)):
        open(pm, "wb").write(b"not hdf5" * 100); fio.read_warp_file(pm)


HDF5_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "hdf5")
CONDA_PY = "/opt/conda/bin/python3.9"          # an unrelated anaconda tree in this image that has h5py


def test_hdf5_reader_vs_files_written_by_h5py():
    """tests/golden/hdf5/*.mat come from h5py 3.3.0 / libhdf5 1.10.6 (oracle/make_golden_hdf5.py)."""
    from fisr_amd import hdf5_min as H
    e = np.load(os.path.join(HDF5_DIR, "expect.npz"))
    disk = np.transpose(e["warp"], (4, 3, 2, 1, 0))
    for name, key, exp in (("contig", "pred", disk), ("chunked", "pred", disk), ("auto", "pred", disk),
                           ("auto", "other", np.arange(10, dtype=np.int16)), ("auto", "grp/x", np.arange(6.).reshape(2, 3)),
                           ("auto", "be", np.arange(5, dtype=np.uint32)), ("latest", "pred", disk), ("latest", "c", disk),
                           ("manychunks", "pred", e["big"])):
        got = H.read_dataset(os.path.join(HDF5_DIR, name + ".mat"), key)
        assert got.dtype == exp.dtype and got.shape == exp.shape and np.array_equal(got, exp), (name, key)
    assert H.list_names(os.path.join(HDF5_DIR, "auto.mat")) == ["be", "grp", "other", "pred"]
    with pytest.raises(KeyError):
        H.read_dataset(os.path.join(HDF5_DIR, "auto.mat"), "nope")
    with pytest.raises(H.Hdf5Error):
        H.read_dataset(os.path.join(HDF5_DIR, "auto.mat"), "grp")           # a group is not a dataset
    # the reference's reader (utils.py:45-54) on the hdf5storage-style fixture, through the io layer
    import fisr_amd.hdf5_min  # noqa: F401
    got = H.read_dataset(os.path.join(HDF5_DIR, "chunked.mat"), "pred")
    assert np.array_equal(np.transpose(got, (4, 3, 2, 1, 0)), e["warp"])
    # a corrupted chunk is caught by the fletcher32 filter
    raw = bytearray(open(os.path.join(HDF5_DIR, "chunked.mat"), "rb").read())
    raw[-40] ^= 0x10
    import tempfile
    with tempfile.NamedTemporaryFile(suffix=".mat") as t:
        t.write(bytes(raw)); t.flush()
        with pytest.raises((H.Hdf5Error, Exception)):
            H.read_dataset(t.name, "pred")


def test_hdf5_reader_on_a_real_matlab_file():
    """scipy ships a MATLAB-written v7.3 file in its test data."""
    from fisr_amd import hdf5_min as H
    scipy_io = pytest.importorskip("scipy.io")
    p = os.path.join(os.path.dirname(scipy_io.__file__), "matlab", "tests", "data", "testhdf5_7.4_GLNX86.mat")
    if not os.path.isfile(p):
        pytest.skip("scipy test data not installed")
    assert H.list_names(p) == ["testdouble"]
    a = H.read_dataset(p, "testdouble")
    assert a.dtype == np.float64 and a.ndim == 2 and a.size == 9          # MATLAB: 0:pi/4:2*pi
    np.testing.assert_allclose(a.ravel(), np.arange(9) * np.pi / 4, rtol=0, atol=1e-15)


def test_hdf5_writer_roundtrip_and_fletcher(tmp_path):
    from fisr_amd import hdf5_min as H
    rng = np.random.default_rng(0)
    for dt in (np.float32, np.float64, np.uint8, np.int16):
        a = (rng.random((3, 17, 9, 4, 2)) * 255).astype(dt)
        for kw in (dict(), dict(chunks=(1, 5, 9, 2, 2)), dict(chunks=(3, 4, 4, 3, 1), compress=7, shuffle=True, checksum=True),
                   dict(chunks=(1, 1, 2, 1, 1), shuffle=True), dict(chunks=(2, 17, 9, 4, 2), compress=1, matlab=False)):
            p = str(tmp_path / "t.mat")
            H.write_dataset(p, "pred", a, **kw)
            b = H.read_dataset(p, "pred")
            assert b.dtype == a.dtype and np.array_equal(a, b), (dt, kw)
    # H5_checksum_fletcher32 restated literally (16-bit big-endian words, 360-word folding) vs the vectorised one
    def ref(data):
        s1 = s2 = 0
        i, ln = 0, len(data) // 2
        while ln:
            t = min(360, ln); ln -= t
            for _ in range(t):
                s1 += (data[i] << 8) | data[i + 1]; i += 2; s2 += s1
            s1 = (s1 & 0xffff) + (s1 >> 16); s2 = (s2 & 0xffff) + (s2 >> 16)
        if len(data) % 2:
            s1 += data[-1] << 8; s2 += s1
            s1 = (s1 & 0xffff) + (s1 >> 16); s2 = (s2 & 0xffff) + (s2 >> 16)
        s1 = (s1 & 0xffff) + (s1 >> 16); s2 = (s2 & 0xffff) + (s2 >> 16)
        return (s2 << 16) | s1
    for n in (0, 1, 2, 3, 7, 720, 721, 5000, 9999):
        x = bytes(rng.integers(0, 256, n, dtype=np.uint8))
        assert H.fletcher32(x) == ref(x), n
    assert H.fletcher32(b"\xff\xff" * 3) == ref(b"\xff\xff" * 3) == 0xFFFFFFFF and H.fletcher32(b"\0" * 10) == 0
    assert H.auto_chunks((3, 1920, 1080, 8, 1), 4) and np.prod(H.auto_chunks((3, 1920, 1080, 8, 1), 4)) * 4 <= 1 << 20


@pytest.mark.skipif(not os.path.exists(CONDA_PY), reason="no anaconda h5py in this image")
def test_hdf5_writer_is_readable_by_libhdf5(tmp_path):
    """Files from the restated writer are opened by the real library (h5py 3.3.0 in /opt/conda)."""
    import subprocess
    a = (np.random.default_rng(3).random((1, 4, 24, 40, 3)) * 255).astype(np.float32)
    fio_h5py = None
    try:
        import h5py as fio_h5py  # noqa: F401
    except ImportError:
        pass
    if fio_h5py is not None:
        pytest.skip("io layer already uses h5py here")
    fio.write_warp_file(str(tmp_path / "w.mat"), a)
    np.save(str(tmp_path / "a.npy"), a)
    code = ("import h5py, numpy as np, sys\n"
            "a = np.load(sys.argv[2])\n"
            "with h5py.File(sys.argv[1], 'r') as f:\n"
            "    d = f['pred']\n"
            "    b = np.transpose(np.array(d, dtype=np.float32), (4, 3, 2, 1, 0))   # utils.py:45-54\n"
            "    assert (a == b).all() and d.compression == 'gzip' and d.shuffle and d.fletcher32\n"
            "    assert d.attrs['MATLAB_class'] == b'single' and f.userblock_size == 512\n"
            "print('OK')\n")
    r = subprocess.run([CONDA_PY, "-W", "ignore", "-c", code, str(tmp_path / "w.mat"), str(tmp_path / "a.npy")],
                       capture_output=True, text=True, cwd=str(tmp_path), timeout=120)
    assert r.returncode == 0 and "OK" in r.stdout, r.stderr[-2000:]


def test_tf_bundle_roundtrip(tmp_path, syn_weights):
    d = tmp_path / "checkpoint_dir" / "FISRnet_exp1"
    d.mkdir(parents=True)
    prefix = str(d / "FISRnet-122000")
    names = list(syn_weights)[:40]                      # > one 4 KiB index block
    small = {k: syn_weights[k] for k in names if syn_weights[k].size < 40000}
    small["FISRnet/level_1/enc/level_0/conv/0/w/Adam"] = np.zeros((3, 3, 29, 64), np.float32)   # optimizer slot
    small["beta1_power"] = np.float32(0.9).reshape(())
    tf_bundle.write_bundle(prefix, small)
    hdr, ent = tf_bundle.read_index(prefix + ".index")
    assert hdr[1] == 1 and set(ent) == set(small)
    back = tf_bundle.read_bundle(prefix, name_filter="FISRnet", verify_crc=True)
    assert "beta1_power" not in back
    for k in small:
        if "FISRnet" in k:
            assert np.array_equal(back[k], small[k]) and back[k].shape == np.shape(small[k])
    path, kind, step = weights.find_checkpoint(str(tmp_path / "checkpoint_dir"), "FISRnet_exp1")
    assert (kind, step) == ("tf_bundle", 122000) and path == prefix
    with pytest.raises(KeyError):
        weights.load_weights(path, kind)                # incomplete checkpoint -> loud failure
    # corruption is detected
    with open(prefix + ".index", "r+b") as f:
        f.seek(10); f.write(b"\xff")
    with pytest.raises(ValueError):
        tf_bundle.read_index(prefix + ".index")
    assert tf_bundle.crc32c(b"123456789") == 0xE3069283   # CRC-32C check value


def test_full_bundle_loads_unchanged(tmp_path, syn_weights):
    d = tmp_path / "ck" / "FISRnet_exp1"
    d.mkdir(parents=True)
    tiny = {k: (v if v.size < 5000 else v) for k, v in syn_weights.items()}
    tf_bundle.write_bundle(str(d / "FISRnet-5"), tiny, with_crc=False)
    path, kind, step = weights.find_checkpoint(str(tmp_path / "ck"), "FISRnet_exp1")
    w = weights.load_weights(path, kind)
    assert len(w) == 276 and step == 5
    assert all(np.array_equal(w[k], syn_weights[k]) for k in w)


def test_cli_flags_match_reference_defaults(tmp_path):
    os.chdir(tmp_path)
    a = fmain.parse_args([])
    assert (a.phase, a.scale_factor, a.exp_num, a.test_patch, a.test_input_size) == ("FISR_for_video", 2, 1, (2, 2), (1080, 1920))
    assert a.checkpoint_dir == "./checkpoint_dir" and a.frame_num == 5 and a.FISR_test_patch == (2, 2)
    a = fmain.parse_args(["--phase", "test", "--test_patch", "(1,1)", "--test_input_size", "96,96", "--precision", "fp32"])
    assert a.test_patch == (1, 1) and a.test_input_size == (96, 96)
    for d in ("checkpoint_dir", "text_dir", "logdir", "test_img_dir"):
        assert os.path.isdir(d)                               # check_args creates them (main.py:108-121)
    with pytest.raises(SystemExit):
        fmain.parse_args(["--test_patch", "2"])
    # --phase train: the reference's hyper-parameters (main.py:64-85) ...
    t = fmain.parse_args(["--phase", "train"])
    assert (t.epoch, t.freq_display, t.init_lr, t.lr_type, t.lr_stair_decay_points, t.lr_decreasing_factor) == (100, 100, 1e-4, "stair_decay", [80, 90], 0.1)
    assert (t.batch_size, t.val_batch_size, t.val_data_size) == (8, 2, 320)
    assert (t.recn_lambda, t.tm1_lambda, t.tm2_lambda, t.tmm_lambda, t.td_lambda, t.ss2_lambda) == (1.0, 1.0, 0.1, 1.0, 0.1, 1.0)
    assert t.train_data_path == "./data/train/LR_LFR/LR_Surfing_SlamDunk_5seq.mat"
    # ... and, like the reference, it starts by reading the pre-made training files (FISRnet.py:177-209)
    with pytest.raises(FileNotFoundError):
        fmain.main(["--phase", "train"])


def test_split_format_roundtrip():
    x = np.random.default_rng(1).standard_normal((2, 3, 4, 32)).astype(np.float32) * 3
    s = splitfmt.to_split(x)
    assert s.shape == (2, 3, 4, 2, 2, 16) and s.dtype == np.uint16
    r = splitfmt.from_split(s)
    assert np.abs(r - x).max() <= np.abs(x).max() * 2.0 ** -17
    assert np.array_equal(splitfmt.split_round(r), r)         # idempotent
    assert np.array_equal(splitfmt.to_split(np.zeros((1, 16), np.float32)), np.zeros((1, 1, 2, 16), np.uint16))


def test_f16f8_record_layout_and_roundtrip():
    """The f16f8 record as the kernels read it (conv3x3.h, r05): per 16 channels 32 B of fp16 h, then per 8 channels {8 x l8 | 8 x h8} --
    both fp8 parts of a channel in one 16-byte unit; x ~ h + l8 * 2^-14 (>= 14 significant bits), h8 = fp8(h) keeps h's sign."""
    rng = np.random.default_rng(2)
    x = (rng.standard_normal((3, 5, 32)) * np.array([1e-3, 1.0, 30.0])[:, None, None]).astype(np.float32)
    s = splitfmt.to_fsplit(x)
    assert s.shape == (3, 5, 2, 64) and s.dtype == np.uint8
    r = splitfmt.from_fsplit(s)
    assert np.abs(r - x).max() <= np.abs(x).max() * 2.0 ** -14
    assert np.array_equal(splitfmt.from_fsplit(splitfmt.to_fsplit(r)), r)                      # idempotent
    h = s[..., :32].copy().view(np.float16).astype(np.float32)                                # [3, 5, 2, 16]
    g = x.reshape(3, 5, 2, 16)
    assert np.array_equal(h, g.astype(np.float16).astype(np.float32))
    for half in range(2):                                                                      # channels 8 half .. 8 half + 7
        l8 = splitfmt.fp8_e4m3_decode(s[..., 32 + 16 * half:40 + 16 * half])
        h8 = splitfmt.fp8_e4m3_decode(s[..., 40 + 16 * half:48 + 16 * half])
        hh = h[..., 8 * half:8 * half + 8]
        assert np.array_equal(h8, splitfmt.fp8_e4m3_decode(splitfmt.fp8_e4m3_encode(np.clip(hh, -448, 448))))
        assert np.array_equal(np.signbit(h8), np.signbit(hh))                                  # relu-on-load masks l8 / h8 by h8's sign
        assert np.abs(hh + l8 * 2.0 ** -14 - g[..., 8 * half:8 * half + 8]).max() <= np.abs(g).max() * 2.0 ** -14
    assert not splitfmt.to_fsplit(np.zeros((1, 16), np.float32)).any()
    big = splitfmt.from_fsplit(splitfmt.to_fsplit(np.full((1, 16), 1e6, np.float32)))         # saturates instead of overflowing
    assert np.isfinite(big).all() and big.max() <= 65504 + 448 * 2.0 ** -14


def test_snappy_decompressor_vs_real_snappy(gold_dir):
    """tests/golden/snappy_blocks.npz was compressed by libsnappy 1.1.8 (oracle/make_golden_snappy.py)."""
    g = np.load(os.path.join(gold_dir, "snappy_blocks.npz"))
    names = sorted(set(n.rsplit("_", 1)[0] for n in g.files))
    assert len(names) >= 10
    for k in names:
        assert tf_bundle._snappy_decompress(g[k + "_snappy"].tobytes()) == g[k + "_raw"].tobytes(), k
    with pytest.raises(ValueError):
        tf_bundle._snappy_decompress(g["text_snappy"].tobytes()[:-5])          # truncated stream


def test_bundle_protos_vs_real_protobuf():
    """The hand-rolled BundleEntryProto / TensorShapeProto wire coding against google.protobuf (messages declared
    here from TensorFlow's tensor_bundle.proto / tensor_shape.proto field numbers)."""
    pytest.importorskip("google.protobuf")
    from google.protobuf import descriptor_pb2, descriptor_pool, message_factory
    fd = descriptor_pb2.FileDescriptorProto(name="fisr_bundle_test.proto", package="t", syntax="proto3")
    F = descriptor_pb2.FieldDescriptorProto
    shape = fd.message_type.add(name="TensorShapeProto")
    dim = shape.nested_type.add(name="Dim")
    dim.field.add(name="size", number=1, type=F.TYPE_INT64, label=F.LABEL_OPTIONAL)
    dim.field.add(name="name", number=2, type=F.TYPE_STRING, label=F.LABEL_OPTIONAL)
    shape.field.add(name="dim", number=2, type=F.TYPE_MESSAGE, label=F.LABEL_REPEATED, type_name=".t.TensorShapeProto.Dim")
    shape.field.add(name="unknown_rank", number=3, type=F.TYPE_BOOL, label=F.LABEL_OPTIONAL)
    ent = fd.message_type.add(name="BundleEntryProto")
    ent.field.add(name="dtype", number=1, type=F.TYPE_INT32, label=F.LABEL_OPTIONAL)
    ent.field.add(name="shape", number=2, type=F.TYPE_MESSAGE, label=F.LABEL_OPTIONAL, type_name=".t.TensorShapeProto")
    ent.field.add(name="shard_id", number=3, type=F.TYPE_INT32, label=F.LABEL_OPTIONAL)
    ent.field.add(name="offset", number=4, type=F.TYPE_INT64, label=F.LABEL_OPTIONAL)
    ent.field.add(name="size", number=5, type=F.TYPE_INT64, label=F.LABEL_OPTIONAL)
    ent.field.add(name="crc32c", number=6, type=F.TYPE_FIXED32, label=F.LABEL_OPTIONAL)
    pool = descriptor_pool.DescriptorPool()
    pool.Add(fd)
    try:
        Entry = message_factory.GetMessageClass(pool.FindMessageTypeByName("t.BundleEntryProto"))
    except AttributeError:                      # older protobuf
        Entry = message_factory.MessageFactory(pool).GetPrototype(pool.FindMessageTypeByName("t.BundleEntryProto"))
    for shape_, off, size, crc in (((3, 3, 64, 128), 0, 294912, 0xDEADBEEF), ((512,), 193264640, 2048, 1), ((), 5, 4, 0xFFFFFFFF),
                                   ((3, 3, 1024, 512), (1 << 33) + 7, 18874368, 123456789)):
        m = Entry(dtype=1, shard_id=0, offset=off, size=size, crc32c=crc)
        for d in shape_:
            m.shape.dim.add(size=d)
        if not shape_:
            m.shape.SetInParent()
        wire = m.SerializeToString()
        e = tf_bundle._parse_entry(wire)                       # real encoder -> my decoder
        assert (e["dtype"], e["shape"], e["offset"], e["size"], e["crc32c"]) == (1, tuple(shape_), off, size, crc)
        back = Entry()
        back.ParseFromString(tf_bundle._entry_proto(1, shape_, off, size, crc))   # my encoder -> real decoder
        assert back.dtype == 1 and tuple(d.size for d in back.shape.dim) == tuple(shape_)
        assert (back.offset, back.size, back.crc32c) == (off, size, crc)


def test_table_reader_vs_hand_assembled_leveldb_table(gold_dir):
    """tests/golden/leveldb_handmade.{index,data-00000-of-00001} were assembled from the LevelDB table-format / tensor_bundle
    specification by oracle/make_golden_leveldb.py, which shares no code with fisr_amd/tf_bundle.py (bit-wise crc32c, own varints,
    own protobuf bytes) and produces what the reader's own writer never does: data blocks with restart interval 16 (entries with
    a non-zero shared-prefix length), 29 small blocks under an index block of SHORTENED separator keys, one block compressed by the
    real libsnappy, an empty metaindex block, the padded 48-byte footer.  Every tensor must come back bit-exact with the block
    and tensor checksums verified -- the pin of the table framing that round 3 left checked only against its own writer (real
    TensorFlow bytes remain absent from this image)."""
    from fisr_amd import tf_bundle
    prefix = os.path.join(gold_dir, "leveldb_handmade")
    g = np.load(prefix + ".npz")
    exp = {k.replace("|", "/"): g[k] for k in g.files}
    header, entries = tf_bundle.read_index(prefix + ".index", verify=True)
    assert header.get("num_shards", 1) == 1 and len(entries) == len(exp) == 114
    got = tf_bundle.read_bundle(prefix, verify_crc=True)
    assert sorted(got) == sorted(exp)
    for k, v in exp.items():
        assert got[k].dtype == np.float32 and got[k].shape == v.shape and np.array_equal(got[k], v), k
    only = tf_bundle.read_bundle(prefix, name_filter="FISRnet/level_3/SR")
    assert sorted(only) == ["FISRnet/level_3/SR/conv/2/b", "FISRnet/level_3/SR/conv/2/w", "FISRnet/level_3/SR/conv/2/w/Adam"]
    # a flipped byte inside a data block must be caught by the block checksum
    raw = bytearray(open(prefix + ".index", "rb").read())
    raw[100] ^= 0x40
    import tempfile
    with tempfile.TemporaryDirectory() as d:
        with open(os.path.join(d, "bad.index"), "wb") as f:
            f.write(bytes(raw))
        with pytest.raises(ValueError):
            tf_bundle.read_index(os.path.join(d, "bad.index"), verify=True)


def test_crc32c_lockstep_path_equals_the_bytewise_register():
    """tf_bundle.crc32c runs buffers >= 1 MiB as 65 536 register states in lock-step and folds them with Z^L (a checkpoint with its
    Adam slots is 580 MB: 46 s byte by byte in Python); same value as the byte-wise table loop, incl. a running crc and a tail
    that is not a whole number of lanes; the check value of the polynomial pins both."""
    from fisr_amd import tf_bundle as t
    assert t.crc32c(b"123456789") == 0xE3069283
    rng = np.random.default_rng(5)
    for n, seed in ((t._CRC_BIG, 0), (t._CRC_BIG + 77_777, 0xDEADBEEF), (3 * t._CRC_BIG + 1, 1)):
        b = rng.integers(0, 256, n, dtype=np.uint8).tobytes()
        assert t.crc32c(b, seed) == t._crc32c_bytewise(b, seed ^ 0xFFFFFFFF) ^ 0xFFFFFFFF
    b = rng.integers(0, 256, t._CRC_BIG + 5, dtype=np.uint8).tobytes()
    assert t.crc32c(b[400_000:], t.crc32c(b[:400_000])) == t.crc32c(b)          # (running crc across the two paths)
