"""CPU tests of the host side: weight container, tiling, and that the C-ABI library builds,
loads and exports every symbol include/fisr.h declares (no compute without a GPU)."""
import ctypes
import os
import re
import subprocess

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
    # --pad_mode: the frame is padded up instead, 1080 -> 1088 rows, and 2 x 2 tiles of 576 x 992 cover it
    assert tiling.pad_hw(1080, 1920, (2, 2)) == (1088, 1920) and tiling.pad_hw(1080, 1920, (1, 1)) == (1088, 1920)
    assert tiling.pad_hw(1024, 1920, (2, 2)) == (1024, 1920)
    assert [(t.in_h, t.in_w) for t in tiling.plan_tiles(1088, 1920, (2, 2))] == [(576, 992)] * 4
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
    # the dynamic surface of the binary IS the header: -fvisibility=hidden + FISR_API, no C++ kernel stubs beside the C entries
    assert len(re.findall(r"^FISR_API ", header, re.M)) == len(declared)
    nm = subprocess.run(["nm", "-D", "--defined-only", lib.SO_PATH], capture_output=True, text=True, check=True).stdout
    exported = sorted(l.split()[-1] for l in nm.splitlines() if l.split()[-2] in "TWBDRV")
    assert exported == declared, sorted(set(exported) ^ set(declared))


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


def test_pack_input_shape_contract():
    """pack_input indexes all 11 tensors with frame 0's row stride, so mismatched inputs must
    be cropped to a common size or rejected -- never read with the wrong stride (reference: independent
    [:h,:w] slices, FISRnet.py:828-843)."""
    torch = pytest.importorskip("torch")
    from fisr_amd.fisrnet import fit_pack_inputs
    z = lambda h, w, c, dt=torch.float32: torch.zeros((h, w, c), dtype=dt)
    fr = [z(40, 64, 3, torch.uint8) for _ in range(3)]
    fl = [z(40, 64, 2) for _ in range(4)]
    wp = [z(40, 64, 3) for _ in range(4)]
    a, b, c, h0, w0 = fit_pack_inputs(fr, fl, wp, 32, 64)
    assert (h0, w0) == (40, 64) and all(t.shape[:2] == (40, 64) for t in a + b + c)
    # larger flow / warp arrays (different padding of the .flo / .mat) are cropped to frame 0's size
    fl2 = [z(48, 80, 2) for _ in range(4)]
    fl2[0][:40, :64, 0] = 7.0
    _, b2, _, _, _ = fit_pack_inputs(fr, fl2, wp, 32, 64)
    assert all(t.shape == (40, 64, 2) and t.is_contiguous() for t in b2) and float(b2[0][39, 63, 0]) == 7.0
    with pytest.raises(ValueError, match="smaller"):          # a warp smaller than the crop
        fit_pack_inputs(fr, fl, [z(24, 64, 3)] + wp[1:], 32, 64)
    with pytest.raises(ValueError, match="common size"):      # covers the crop but not frame 0
        fit_pack_inputs(fr, fl, [z(36, 64, 3)] + wp[1:], 32, 64)
    with pytest.raises(ValueError, match=r"\[h,w,2\]"):        # wrong channel count
        fit_pack_inputs(fr, [z(40, 64, 3)] + fl[1:], wp, 32, 64)
    with pytest.raises(ValueError, match="exceeds"):
        fit_pack_inputs(fr, fl, wp, 64, 64)
    with pytest.raises(ValueError):                           # frames 1-2 must match frame 0 as well
        fit_pack_inputs([fr[0], z(32, 64, 3, torch.uint8), fr[2]], fl, wp, 32, 64)


def test_checkpoint_fallback_sorts_by_step(tmp_path, syn_weights, capsys):
    """Without a `checkpoint` state file the highest training step wins (natural order: 10 after 9)."""
    d = tmp_path / "ckpt" / "FISRnet_exp1"
    d.mkdir(parents=True)
    small = {k: v for k, v in list(syn_weights.items())[:2]}
    for step in (9, 10, 2):
        weights.save_npz(str(d / f"FISRnet-{step}.npz"), small)
    path, kind, step = weights.find_checkpoint(str(tmp_path / "ckpt"), "FISRnet_exp1")
    assert (os.path.basename(path), kind, step) == ("FISRnet-10.npz", "npz", 10)
    assert "falling back" in capsys.readouterr().out
    # with a state file the named checkpoint wins, silently
    (d / "checkpoint").write_text('model_checkpoint_path: "FISRnet-9"\n')
    path, kind, step = weights.find_checkpoint(str(tmp_path / "ckpt"), "FISRnet_exp1")
    assert (os.path.basename(path), step) == ("FISRnet-9.npz", 9)
    assert "falling back" not in capsys.readouterr().out


def test_one_default_precision_everywhere():
    from fisr_amd import fisrnet, main
    assert fisrnet.default_args().precision == fisrnet.DEFAULT_PRECISION == "fp32"
    assert main.parse_args(["--phase", "test"]).precision == fisrnet.DEFAULT_PRECISION
    src = open(os.path.join(ROOT, "bench.py")).read()
    assert 'ap.add_argument("--precision", default="fp32"' in src


def test_python_mirror_of_the_header_enums():
    """fisr_amd/lib.py repeats include/fisr.h's precision and conv-flag values (ctypes has no header): they must not drift."""
    import re
    from fisr_amd import lib as flib
    header = open(os.path.join(ROOT, "include", "fisr.h")).read()
    enums = {m.group(1): int(m.group(2)) for m in re.finditer(r"\b(FISR_(?:PREC|CONV)_[A-Z0-9_]+)\s*=\s*(\d+)", header)}
    assert len(enums) >= 13, enums
    for name, value in enums.items():
        mirror = name[len("FISR_"):]
        assert getattr(flib, mirror) == value, (name, value, getattr(flib, mirror, None))


def test_xavier_initialiser_and_lr_schedule():
    """ops.py:8-9: xavier_initializer(uniform=False) = truncated normal, stddev sqrt(1.3 * 2 / (fan_in + fan_out)), zero
    biases; FISRnet.py:227-245: tf.train.piecewise_constant switches one step AFTER a boundary."""
    import types
    from fisr_amd import train_harness, weights
    W = weights.xavier_weights(3)
    weights.check_complete(W)
    w = W["FISRnet/level_3/enc/level_1/res_block/0/conv/0/w"]            # 128 -> 128
    std = np.sqrt(1.3 * 2.0 / (9 * 256))
    assert np.abs(w).max() <= 2.0 * std + 1e-7                           # truncated at two sigma
    assert abs(w.std() / (std * 0.8796) - 1) < 0.02                      # std of a unit normal truncated at 2: 0.8796
    assert not W["FISRnet/level_1/SR/conv/2/b"].any()
    a = types.SimpleNamespace(lr_type="stair_decay", lr_stair_decay_points=[2, 3], init_lr=1e-4, lr_decreasing_factor=0.1, epoch=4,
                              lr_linear_decay_point=2)
    lr = lambda step: train_harness.learning_rate(a, 0, step, 10)
    assert lr(20) == 1e-4 and abs(lr(21) - 1e-5) < 1e-12 and abs(lr(30) - 1e-5) < 1e-12 and abs(lr(31) - 1e-6) < 1e-13


def test_optimizer_state_lookup(tmp_path, syn_weights):
    """A weights-only container has no optimizer state; one with `<var>/Adam`, `<var>/Adam_1` and the beta powers has."""
    from fisr_amd import weights
    p0 = str(tmp_path / "a.npz")
    weights.save_npz(p0, syn_weights)
    assert weights.load_optimizer_state(p0) is None
    full = dict(syn_weights)
    for k, v in syn_weights.items():
        full[k + "/Adam"] = np.full_like(v, 0.5)
        full[k + "/Adam_1"] = np.full_like(v, 0.25)
    full["beta1_power"], full["beta2_power"] = np.float64(0.9 ** 3), np.float64(0.999 ** 3)
    p1 = str(tmp_path / "b.npz")
    np.savez(p1, **full)
    st = weights.load_optimizer_state(p1)
    assert st is not None and len(st) == 2 * 276 + 2 and float(st["beta2_power"]) == 0.999 ** 3
    W = weights.load_weights(p1)                                          # the slots do not disturb the weight loader
    assert len(W) == 276


def test_check_published_compares_with_the_readme_figures(capsys):
    """main.py --phase test --check_published: the four averages against README.md:97 within BASELINE.json's tolerance."""
    from fisr_amd import main as fmain
    assert fmain.check_published(dict(FISR_PSNR=37.862, SR_PSNR=48.05, FISR_SSIM=0.97431, SR_SSIM=0.9918))
    assert not fmain.check_published(dict(FISR_PSNR=37.80, SR_PSNR=48.07, FISR_SSIM=0.9743, SR_SSIM=0.9921))
    out = capsys.readouterr().out
    assert out.count("published check:") == 10 and "OUTSIDE" in out and "README.md:97" in out
    assert fmain.parse_args(["--phase", "test", "--check_published"]).check_published


def test_code_object_has_no_store_data_hazard():
    """r04: a 16-byte buffer store reads its data VGPRs after it has issued; a vector instruction that overwrites one of them in the
    very next slot wins that race now and then on gfx950 -- also when the store carries a register soffset, the case LLVM's hazard
    recogniser exempts.  The GENERAL instantiation of the F(4x4) kernel was wrong in 0.02-0.15 % of its outputs that way (a
    different set on every run) until its stores became asm with the wait state attached.  This scans the device code of the
    BUILT library for such pairs (scripts/isa_store_hazard.py: unbundle, llvm-objdump, scan): every kernel, every store of more
    than 8 bytes."""
    import importlib.util
    import shutil
    if not os.path.isfile("/opt/rocm/lib/llvm/bin/llvm-objdump") and not shutil.which("llvm-objdump"):
        pytest.skip("no llvm-objdump")
    lib.build()
    spec = importlib.util.spec_from_file_location("isa_store_hazard", os.path.join(ROOT, "scripts", "isa_store_hazard.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    text = m.disassemble(lib.SO_PATH)
    assert text.count("buffer_store_dwordx4") > 100 and "conv3x3_wf4_kernel" in text
    found = m.scan(text)
    assert not found, found[:5]
    # the scanner does find the pattern (a synthetic listing)
    bad = "_Zfoo:\n\tbuffer_store_dwordx4 v[4:7], v1, s[0:3], s9 offen\n\tv_mul_f32_e32 v6, v2, v3\n\ts_endpgm\n"
    ok = "_Zfoo:\n\tbuffer_store_dwordx4 v[4:7], v1, s[0:3], s9 offen\n\ts_nop 0\n\tv_mul_f32_e32 v6, v2, v3\n\ts_endpgm\n"
    assert len(m.scan(bad)) == 1 and not m.scan(ok)


def _kernel_notes():
    """(mangled name -> {vgpr, spill, scratch}) of every kernel in the built library's gfx950 code object (llvm-readelf --notes)."""
    import re
    import shutil
    import subprocess
    import tempfile
    L = "/opt/rocm/lib/llvm/bin"
    if not os.path.isfile(os.path.join(L, "llvm-readelf")) or not os.path.isfile(os.path.join(L, "clang-offload-bundler")):
        pytest.skip("no llvm-readelf / clang-offload-bundler")
    lib.build()
    d = tempfile.mkdtemp()
    try:
        subprocess.run([f"{L}/llvm-objcopy", "-O", "binary", "--only-section=.hip_fatbin", lib.SO_PATH, f"{d}/fat.bin"], check=True)
        subprocess.run([f"{L}/clang-offload-bundler", "--unbundle", "--type=o", f"--input={d}/fat.bin",
                        "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", f"--output={d}/dev.co"], check=True)
        txt = subprocess.run([f"{L}/llvm-readelf", "--notes", f"{d}/dev.co"], check=True, capture_output=True, text=True).stdout
    finally:
        shutil.rmtree(d, ignore_errors=True)
    out = {}
    for m in re.finditer(r"\.name:\s+(\S+)(.*?)(?=\.name:\s+_Z|\Z)", txt, re.S):
        g = lambda k: int((re.search(k + r":\s+(\d+)", m.group(2)) or [0, "-1"])[1])
        out[m.group(1)] = {"vgpr": g(r"\.vgpr_count"), "spill": g(r"\.vgpr_spill_count"), "scratch": g(r"\.private_segment_fixed_size")}
    return out


def test_register_budget_of_the_hot_kernels():
    """The conv kernels live at the edge of their register files (every attempt to add live state to the F(4x4)
    kernel or to the f16f8 direct kernel was paid in spills -- +30 % time at 74 spilled registers), and a change elsewhere in a
    shared header can tip them over without any test turning red.  This pins what the SHIPPED code object has: spill counts of the
    F(4x4) instantiations as measured with the numbers in DESIGN (a regression shows up here, an improvement means: lower the bound),
    no spills at all in the other product kernels of the inference path, and the occupancy the launch bounds ask for."""
    k = _kernel_notes()
    assert len(k) > 100
    wf4 = {n: v for n, v in k.items() if "conv3x3_wf4_kernel" in n}
    assert len(wf4) == 7
    for n, v in wf4.items():
        general = n.endswith("Lb1EEEvNS_8ConvArgsEi")          # <.., GENERAL = true> (r06: SHARE is a template parameter in -DFISR_F4_SHARE builds only)
        assert v["vgpr"] <= 256 and v["spill"] <= (48 if general else 46), (n, v)       # 1 workgroup of 8 waves per CU = 2 waves per SIMD
    clean = [n for n, v in wf4.items() if v["spill"] == 0]
    assert any("ILb0ELb1ELb0ELb0ELb0EEE" in n for n in clean), clean                          # the plain residual instantiation stays spill-free
    # hygiene (r06): what only diagnostics builds carry is not in the shipped code object
    assert not any("head_conv_strip_kernel" in n or "conv3x3_wino8b_kernel" in n or "conv3x3_wino8_kernel" in n or "conv3x3_wf4x" in n for n in k), \
        [n for n in k if "strip" in n or "wino8b" in n]
    for n, v in k.items():
        if "conv3x3_dma_f16_kernel" in n or "conv3x3_wino8p_kernel" in n or "head_conv_f32_kernel" in n or "prep_level_frames_kernel" in n:
            assert v["spill"] == 0 and v["scratch"] == 0, (n, v)
        if "conv3x3_dma_f16_kernel" in n:
            assert v["vgpr"] <= 256, (n, v)                                               # 2 workgroups of 4 waves per CU
    # the split-format direct kernels at MR = 2 (two workgroups of 4 waves per CU): no spills, at most 256 registers
    mr2 = {n: v for n, v in k.items() if "conv3x3_mfma_kernel" in n and n.endswith("Li2EEEvNS_8ConvArgsE") and ("bsplit" in n or "fsplit" in n)}
    assert len(mr2) >= 8
    for n, v in mr2.items():
        assert v["spill"] == 0 and v["vgpr"] <= 256, (n, v)


def test_bench_counter_fields_need_the_same_launch_population():
    """bench.py takes a counter-derived field from profiles/pmc_traffic.json only when the table's entry is THIS engine's launch
    population: dispatches = a whole number of forward passes (read off the engine's dominant kernel) x this run's launches per step.
    A kernel name shared with other engines of the same PMC run (r03: maxpool2, the F(2x2) kernel) fails that and is dropped."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bench.py"))
    b = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(b)
    assert b._pmc_passes({"dispatches": 188}, 47.0) == 4                      # the fp32 engine's dominant kernel: four passes
    assert b._pmc_passes({"dispatches": 169}, 4.0) == 0                       # 42.25: another engine's launches are in there
    assert b._pmc_passes({}, 47.0) == 0 and b._pmc_passes({"dispatches": 10}, 0) == 0
    assert b._pmc_same_population({"dispatches": 132}, 4, 33.0)               # residual instantiation: 33 per step x 4
    assert not b._pmc_same_population({"dispatches": 58}, 4, 1.0)             # maxpool2 of all engines against the one launch left here
    assert not b._pmc_same_population({"dispatches": 132}, 0, 33.0)
    # (keys are prefixes of the demangled names, without the closing bracket: in -DFISR_F4_SHARE builds a SHARE parameter sits behind these)
    assert b._pmc_key("conv3x3_wf4<f32w4,relu_in,nores>") == "conv3x3_wf4_kernel<true, false, false, false, false"
    assert b._pmc_key("conv3x3_wf4<f32w4,plain,res+pool>") == "conv3x3_wf4_kernel<false, true, true, false, false"
    assert "conv3x3_wf4_kernel<true, false, false, false, false, false>".startswith(b._pmc_key("conv3x3_wf4<f32w4,relu_in,nores>"))
    assert b._pmc_key("conv3x3_dma_fs<f16f8,tw32,relu_in,nores>") == "conv3x3_dma_fs_kernel<32, true, false, false>"
    assert b._pmc_key("conv3x3_dma_fs<f16f8,tw64,plain,res+pool>") == "conv3x3_dma_fs_kernel<64, false, true, true>"
    assert "conv3x3_dma_f16_kernel<false, 2>".startswith(b._pmc_key("conv3x3_dma<f16>"))
    # an engine profiled alone has its own table (r05: `mixed` shares its kernel names with f16f8 / fp16)
    assert b._pmc_file("no_such_engine").endswith(os.path.join("profiles", "pmc_traffic.json"))


def test_prepare_paths_are_refused_before_anything_is_overwritten(tmp_path):
    """(a) `--prepare_ss 2` derives the stride-2 names by 'ss1' -> 'ss2' -- names without 'ss1' would make it overwrite the
    stride-1 files with [N, 4, ...] data: refused before a frame is read; (b) `--phase test` in auto mode with neither pre-made file
    and no PWC-Net weights names the missing FILES (not the estimator); (c) only one of the two files present is its own error.
    All three are decided on the host before the GPU is touched, so a stand-in `net` will do."""
    from types import SimpleNamespace
    from fisr_amd import harness
    flo, mat = str(tmp_path / "flow.flo"), str(tmp_path / "warp.mat")
    args = SimpleNamespace(prepare="auto", pwc_ckpt=str(tmp_path / "nope" / "pwcnet.ckpt"), synthetic_weights=None)
    net = SimpleNamespace(_finalized=True, args=args, test_data_path=str(tmp_path), test_label_path=str(tmp_path), test_flow_data_path=flo,
                          test_warped_data_path=mat, test_input_size=(96, 96), test_patch=(1, 1), device="cuda:0")
    with pytest.raises(ValueError, match="do not contain 'ss1'"):
        harness.prepare_scene_set(net, args, ss=2)
    with pytest.raises(FileNotFoundError, match="flow.flo.*warp.mat.*no PWC-Net weights"):
        harness.run_test(net)
    open(flo, "wb").close()
    with pytest.raises(FileNotFoundError, match="only one of the pre-made files"):
        harness.run_test(net)
    assert not harness._pwc_available(args)
    assert harness._pwc_available(SimpleNamespace(synthetic_weights=3, pwc_ckpt=None))


def test_committed_pmc_tables_hold_no_impossible_clock():
    """review r05, item 6: profiles/pmc_traffic*.json derive a shader clock per kernel; rows whose cycles and durations came from
    different passes once read 3.4 - 5.7 GHz.  scripts/summarize_prof.py now nulls such rows: every clock in a committed table lies in
    0.5 - 2.5 GHz or is null (with the reason beside it), and a null clock takes the cycle-derived columns with it."""
    import glob
    import json
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    files = sorted(glob.glob(os.path.join(root, "profiles", "pmc_traffic*.json")))
    assert files
    for f in files:
        for k, e in json.load(open(f)).items():
            if k == "_meta" or "gui_active_cycles_per_launch" not in e:
                continue
            clk = e.get("shader_clock_mhz")
            if clk is None:
                assert e.get("cycle_columns_nulled") and e.get("mfma_busy_frac") is None, (f, k)
            else:
                assert 500.0 <= clk <= 2500.0, (f, k, clk)
