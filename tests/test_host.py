"""CPU tests of the host side: weight container, tiling, and that the C-ABI library builds,
loads and exports every symbol include/fisr.h declares (no compute without a GPU)."""
import ctypes
import os
import re

import numpy as np
import pytest

from fisr_amd import lib, tiling, weights

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_variable_inventory():
    specs = weights.conv_specs()
    assert len(specs) == 138
    shapes = weights.variable_shapes()
    assert len(shapes) == 276
    assert weights.num_parameters() == 48316251
    assert shapes["FISRnet/level_1/enc/level_0/conv/0/w"] == (3, 3, 29, 64)
    assert shapes["FISRnet/level_2/enc/level_0/conv/0/w"] == (3, 3, 38, 64)
    assert shapes["FISRnet/level_3/dec/level_0/resize/w"] == (3, 3, 128, 64)
    assert shapes["FISRnet/level_3/dec/level_2/conv/0/w"] == (3, 3, 512, 256)
    assert shapes["FISRnet/level_3/FI-SR/conv/1/b"] == (256,)
    assert shapes["FISRnet/level_3/SR/conv/2/w"] == (3, 3, 64, 3)
    assert shapes["FISRnet/level_3/FI-SR/conv/2/w"] == (3, 3, 64, 6)


def test_npz_roundtrip_and_checkpoint_lookup(tmp_path, syn_weights):
    d = tmp_path / "ckpt" / "FISRnet_exp1"
    d.mkdir(parents=True)
    assert weights.find_checkpoint(str(tmp_path / "ckpt"), "FISRnet_exp1") == (None, None, 0)
    small = {k: v for k, v in list(syn_weights.items())[:4]}
    weights.save_npz(str(d / "FISRnet-122000.npz"), small)
    path, kind, step = weights.find_checkpoint(str(tmp_path / "ckpt"), "FISRnet_exp1")
    assert kind == "npz" and step == 122000
    back = weights.load_npz(path)
    assert all(np.array_equal(back[k], small[k]) for k in small)
    with pytest.raises(KeyError):
        weights.load_weights(path)            # incomplete container must fail like saver.restore
    bad = dict(syn_weights)
    bad["FISRnet/level_1/SR/conv/2/b"] = np.zeros(4, np.float32)
    with pytest.raises(ValueError):
        weights.check_complete(bad)


def test_tiles_default_1080p():
    assert tiling.crop_hw(1080, 1920, (2, 2)) == (1024, 1920)
    assert tiling.crop_hw(1080, 1920, (1, 1)) == (1056, 1920)
    tiles = tiling.plan_tiles(1024, 1920, (2, 2))
    assert [(t.h_lo, t.h_hi, t.w_lo, t.w_hi) for t in tiles] == [
        (0, 544, 0, 992), (0, 544, 928, 1920), (480, 1024, 0, 992), (480, 1024, 928, 1920)]
    assert all(t.in_h == 544 and t.in_w == 992 and t.out_h == 1024 and t.out_w == 1920 for t in tiles)
    assert [(t.src_y, t.src_x) for t in tiles] == [(0, 0), (0, 64), (64, 0), (64, 64)]
    with pytest.raises(ValueError):
        tiling.plan_tiles(1080, 1920, (2, 2))


def test_tiles_vs_reference_fixture(gold_dir):
    g = np.load(os.path.join(gold_dir, "ref_utils.npz"))
    for row in g["tiling"]:
        h, w, nh, nw, p, hl, hh, wl, wh, ah, aw, th, tw = (int(v) for v in row)
        t = tiling.plan_tiles(h, w, (nh, nw))[p]
        assert (t.h_lo, t.h_hi, t.w_lo, t.w_hi, t.out_h, t.out_w) == (hl, hh, wl, wh, th, tw)
        assert tiling.get_hw_boundary(32, h, w, p // nw, h // nh, p % nw, w // nw) == (hl, hh, wl, wh, ah, aw)
    # which HR pixels survive: compare with the reference's trim on an index image
    for p in range(4):
        t = tiling.plan_tiles(128, 192, (2, 2))[p]
        idx = np.arange(t.in_h * 2 * t.in_w * 2, dtype=np.int64).reshape(t.in_h * 2, t.in_w * 2)
        mine = idx[t.src_y:t.src_y + t.out_h, t.src_x:t.src_x + t.out_w]
        assert np.array_equal(mine, g[f"trim_{p}"][0, :, :, 0])


def test_library_builds_loads_and_exports_header_symbols():
    lib.build()
    L = lib.lib()
    header = open(os.path.join(ROOT, "include", "fisr.h")).read()
    declared = sorted(set(re.findall(r"\b(fisr_[a-z0-9_]+)\s*\(", header)))
    assert len(declared) >= 18
    for name in declared:
        assert hasattr(L, name), f"{name} declared in include/fisr.h but not exported"
    assert sorted(lib.EXPORTS) == declared
    assert b"gfx950" in L.fisr_version()


def test_product_never_imports_oracle():
    """The product path must not route through the oracle (or any CPU fallback)."""
    pkg = os.path.join(ROOT, "fisr_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".h", ".hip", ".cpp")):
                src = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(import|from)\s+(oracle|fisr_oracle|c_oracle)", src, re.M), f
                assert "libfisr_oracle" not in src, f


def test_no_gpu_fails_loudly():
    torch = pytest.importorskip("torch")
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    ctx = ctypes.c_void_p()
    rc = lib.lib().fisr_create(ctypes.byref(ctx), 0)
    assert rc < 0 and b"no HIP device" in lib.lib().fisr_last_error(None)
    from fisr_amd.fisrnet import FISRnet, FisrError
    with pytest.raises(FisrError):
        FISRnet()


def test_header_is_plain_c(tmp_path):
    """include/fisr.h must be consumable by a C compiler (the drop-in boundary is a C ABI)."""
    import shutil
    import subprocess
    gcc = shutil.which("gcc")
    if not gcc:
        pytest.skip("no gcc")
    src = tmp_path / "t.c"
    src.write_text('#include "fisr.h"\nint main(void) { fisr_ctx* c = 0; (void)c; return FISR_OK; }\n')
    subprocess.check_call([gcc, "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-fsyntax-only",
                           "-I", os.path.join(ROOT, "include"), str(src)])
