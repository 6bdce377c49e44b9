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
    try:
        import h5py  # noqa: F401
    except ImportError:
        with pytest.raises(ImportError):
            fio.read_warp_file(str(tmp_path / "x.mat"))


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
    assert fmain.main(["--phase", "train"]) == 2


def test_split_format_roundtrip():
    x = np.random.default_rng(1).standard_normal((2, 3, 4, 32)).astype(np.float32) * 3
    s = splitfmt.to_split(x)
    assert s.shape == (2, 3, 4, 2, 2, 16) and s.dtype == np.uint16
    r = splitfmt.from_split(s)
    assert np.abs(r - x).max() <= np.abs(x).max() * 2.0 ** -17
    assert np.array_equal(splitfmt.split_round(r), r)         # idempotent
    assert np.array_equal(splitfmt.to_split(np.zeros((1, 16), np.float32)), np.zeros((1, 1, 2, 16), np.uint16))
