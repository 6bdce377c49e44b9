"""fisr_forward_frames (include/fisr.h; r04): the graph seam with input assembly, tile cut and level inputs folded into one kernel per
level (FISRnet.py:828-843, :853-857, :81,112-113,144; SURVEY 2.2 "pack -> prep as one kernel").  The contract is bit-identity with
the three-step path fisr_pack_input -> tile slices -> fisr_forward, which itself is checked against the oracle in test_gpu_parity.py
(pack_input bit-exact vs oracle.assemble_input, the forward vs the fp64 oracle) -- so these tests compare the two HIP paths
value for value, on every engine, and pin the level-3 input of the fused kernel against the oracle's assembly directly."""
import ctypes

import numpy as np
import pytest

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu

import fisr_oracle as O  # noqa: E402
from fisr_amd import lib as flib  # noqa: E402
from fisr_amd import tiling  # noqa: E402
from fisr_amd.fisrnet import FISRnet  # noqa: E402


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch.device("cuda:0")


def _sources(seed, h0, w0, windows=1):
    rng = np.random.default_rng(seed)
    out = []
    for _ in range(windows):
        frames = [torch.from_numpy(rng.integers(0, 256, (h0, w0, 3)).astype(np.uint8)).cuda() for _ in range(3)]
        flows = [torch.from_numpy((rng.standard_normal((h0, w0, 2)) * 60).astype(np.float32)).cuda() for _ in range(4)]
        warps = [torch.from_numpy((rng.random((h0, w0, 3)) * 300 - 20).astype(np.float32)).cuda() for _ in range(4)]
        out.append((frames, flows, warps))
    return out


@pytest.mark.parametrize("precision", ["fp32", "fp32d", "bf16x3", "f16f8", "fp16", "mixed"])
def test_forward_frames_bit_identical_to_pack_slice_forward(dev, syn_weights, precision):
    """3 windows x 2x2 tiles of a 128x192 crop out of 135x200 frames (crop and row pitch differ), every engine: the stitched
    prediction of the fused path equals the one of pack_input -> forward_tiled bit for bit."""
    net = FISRnet(device="cuda:0", precision=precision)
    net.set_weights(syn_weights)
    try:
        h, w = 128, 192
        wins = _sources(31, 135, 200, windows=3)
        inp = torch.cat([net.pack_input(fr, fl, wp, h, w) for fr, fl, wp in wins], dim=0)
        ref = net.forward_tiled(inp, (2, 2))
        got = net.forward_tiled_frames(wins, h, w, (2, 2))
        torch.cuda.synchronize()
        assert got.shape == ref.shape == (3, 2 * h, 2 * w, 9)
        assert torch.equal(got, ref)
        assert float(ref.abs().max()) > 1e-3
    finally:
        net.close()


def test_forward_frames_more_items_than_one_call_takes(dev, syn_weights):
    """5 windows x 4 tiles = 20 items > FISR_MAX_SRC_ITEMS: the Python surface splits them over two calls; results unchanged."""
    net = FISRnet(device="cuda:0", precision="f16f8")
    net.set_weights(syn_weights)
    try:
        h, w = 64, 128
        wins = _sources(32, 64, 128, windows=5)
        inp = torch.cat([net.pack_input(fr, fl, wp, h, w) for fr, fl, wp in wins], dim=0)
        assert torch.equal(net.forward_tiled_frames(wins, h, w, (2, 2)), net.forward_tiled(inp, (2, 2)))
    finally:
        net.close()


def _item(fr, fl, wp, y0, x0):
    it = flib.SrcItem()
    for j in range(3):
        it.frames[j] = fr[j].data_ptr()
    for j in range(4):
        it.flows[j] = fl[j].data_ptr()
        it.warps[j] = wp[j].data_ptr()
    it.y0, it.x0 = y0, x0
    return it


def test_forward_frames_level_outputs_and_argument_checks(dev, syn_weights):
    """Through the C-ABI directly: out_l2 / out_l1 of an off-origin rectangle equal fisr_forward's on the packed slice (levels 1 and 2
    read every 4th / 2nd source pixel of the rectangle); rectangles outside the frame, too many items, sizes that are not multiples
    of 32 and misaligned flow planes are refused with FISR_EINVAL and a message."""
    net = FISRnet(device="cuda:0", precision="fp32")
    net.set_weights(syn_weights)
    L = flib.lib()
    try:
        h0, w0, h, w, y0, x0 = 100, 170, 64, 96, 33, 71
        (fr, fl, wp), = _sources(33, h0, w0)
        packed = net.pack_input(fr, fl, wp, h0, w0)                     # the whole frame, then the rectangle
        l1, l2, l3 = net.model(packed[:, y0:y0 + h, x0:x0 + w, :].contiguous())
        o3 = torch.empty_like(l3); o2 = torch.empty_like(l2); o1 = torch.empty_like(l1)
        ws = net._workspace(1, h, w)
        st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
        items = (flib.SrcItem * 1)(_item(fr, fl, wp, y0, x0))
        vp = lambda t: ctypes.c_void_p(t.data_ptr())
        call = lambda its, n, hh, ww: L.fisr_forward_frames(net._ctx, its, n, h0, w0, hh, ww, vp(o3), vp(o2), vp(o1), vp(ws), ws.numel(), st)
        assert call(items, 1, h, w) == 0
        torch.cuda.synchronize()
        assert torch.equal(o3, l3) and torch.equal(o2, l2) and torch.equal(o1, l1)
        # refusals
        bad = (flib.SrcItem * 1)(_item(fr, fl, wp, h0 - h + 1, 0))
        assert call(bad, 1, h, w) == -1 and b"inside" in L.fisr_last_error(net._ctx)
        many = (flib.SrcItem * (flib.MAX_SRC_ITEMS + 1))(*[_item(fr, fl, wp, 0, 0)] * (flib.MAX_SRC_ITEMS + 1))
        assert call(many, flib.MAX_SRC_ITEMS + 1, h, w) == -1 and b"FISR_MAX_SRC_ITEMS" in L.fisr_last_error(net._ctx)
        assert call(items, 1, 48, 96) == -1 and b"multiples of 32" in L.fisr_last_error(net._ctx)
        mis = _item(fr, fl, wp, 0, 0)
        mis.flows[2] = fl[2].data_ptr() + 4
        assert call((flib.SrcItem * 1)(mis), 1, h, w) == -1 and b"aligned" in L.fisr_last_error(net._ctx)
    finally:
        net.close()


def test_tile_rectangles_are_the_reference_plan():
    """The rectangles handed to fisr_forward_frames are tiling.plan_tiles' (pinned to the reference's get_HW_boundary in
    tests/test_host.py): the default plan on the 1024 x 1920 crop."""
    plan = tiling.plan_tiles(1024, 1920, (2, 2), 2)
    assert [(t.h_lo, t.h_hi, t.w_lo, t.w_hi) for t in plan] == [(0, 544, 0, 992), (0, 544, 928, 1920), (480, 1024, 0, 992), (480, 1024, 928, 1920)]


def test_level3_input_of_the_fused_kernel_vs_oracle_assembly(dev, syn_weights):
    """The oracle's own input assembly (oracle.assemble_input = FISRnet.py:828-843) on the rectangle, pushed through the
    oracle-checked fisr_forward, against the fused call -- ties the fused path to the oracle without the pack_input kernel."""
    net = FISRnet(device="cuda:0", precision="fp32d")
    net.set_weights(syn_weights)
    try:
        h0, w0, h, w, y0, x0 = 70, 140, 64, 128, 5, 9
        (fr, fl, wp), = _sources(34, h0, w0)
        img9 = np.concatenate([f.cpu().numpy() for f in fr], axis=2)[y0:y0 + h, x0:x0 + w]
        fl8 = np.concatenate([f.cpu().numpy() for f in fl], axis=2)[y0:y0 + h, x0:x0 + w]
        wp12 = (np.concatenate([f.cpu().numpy() for f in wp], axis=2) / np.float32(255.))[y0:y0 + h, x0:x0 + w]
        exp_in = torch.from_numpy(O.assemble_input(img9, fl8, wp12).astype(np.float32)).cuda()
        _, _, ref = net.model(exp_in, want_all=False)
        o3 = torch.empty_like(ref)
        ws = net._workspace(1, h, w)
        items = (flib.SrcItem * 1)(_item(fr, fl, wp, y0, x0))
        rc = flib.lib().fisr_forward_frames(net._ctx, items, 1, h0, w0, h, w, ctypes.c_void_p(o3.data_ptr()), None, None,
                                            ctypes.c_void_p(ws.data_ptr()), ws.numel(), ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
        assert rc == 0
        torch.cuda.synchronize()
        assert torch.equal(o3, ref)
    finally:
        net.close()


@pytest.mark.parametrize("precision", ["f16f8", "fp32", "mixed"])
def test_forward_frames_full_size_stack_bit_identical(dev, syn_weights, precision):
    """cfg2's own shape: 3 windows of 1080x1920 frames cropped to 1024x1920, 2x2 tiles of 544x992 -> the 12 items of one call (what
    bench.py's step runs) against pack_input -> 12-tile forward, bit for bit -- on f16f8 and, r06, on the engines that carry the
    numbers (fp32: the headline; mixed: cfg5's); and unpack_output writing into caller-owned slots.  With
    test_gpu_parity.py::test_full_size_batch_of_twelve_equals_single_tiles (12-tile forward == single tiles == the oracle's sparse
    grid on tiles 0 and 11) this ties the schedule the bench times to the oracle."""
    net = FISRnet(device="cuda:0", precision=precision)
    net.set_weights(syn_weights)
    try:
        h, w = 1024, 1920
        wins = _sources(35, 1080, 1920, windows=3)
        inp = torch.cat([net.pack_input(fr, fl, wp, h, w) for fr, fl, wp in wins], dim=0)
        ref = net.forward_tiled(inp, (2, 2))
        del inp
        got = net.forward_tiled_frames(wins, h, w, (2, 2))
        torch.cuda.synchronize()
        assert torch.equal(got, ref)
        yuv0, rgb0 = net.unpack_output(ref[0])
        slot = torch.zeros((2, 2 * h, 2 * w, 9), dtype=torch.uint8, device="cuda")
        rgbb = torch.zeros((3, 2 * h, 2 * w, 3), dtype=torch.uint8, device="cuda")
        y1, r1 = net.unpack_output(got[0], out_yuv=slot[1], out_rgb=rgbb)
        assert y1.data_ptr() == slot[1].data_ptr() and torch.equal(slot[1], yuv0) and torch.equal(rgbb, rgb0) and not bool(slot[0].any())
        with pytest.raises(ValueError):
            net.unpack_output(got[0], out_yuv=slot[:, :, :, :3])
    finally:
        net.close()


@pytest.mark.parametrize("precision", ["fp32", "mixed"])
def test_forward_is_hip_graph_capturable(dev, syn_weights, precision):
    """fisr_forward only enqueues on the caller's stream -- no allocation, no synchronisation, no host read-back (the arena is the
    caller's, kernel attributes are set on the first call): a forward can be captured into a HIP graph and replayed, with the
    eager result bit for bit.  (scripts/probes/hipgraph_probe.py, late r04: the replay of the 12-tile forward takes what the eager
    stream takes, 112.5 / 64.6 ms -- the step is not launch-bound; this pins the property, not a speed-up.)"""
    net = FISRnet(device="cuda:0", precision=precision)
    net.set_weights(syn_weights)
    try:
        g = torch.Generator(device="cuda").manual_seed(3)
        x = torch.rand((2, 64, 96, 29), device="cuda", generator=g)
        ref = net.model(x, want_all=False)[-1].clone()
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            net.model(x, want_all=False)
        torch.cuda.current_stream().wait_stream(s)
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr):
            out = net.model(x, want_all=False)[-1]
        out.zero_()
        gr.replay()
        torch.cuda.synchronize()
        assert torch.equal(out, ref)
        x.copy_(torch.rand((2, 64, 96, 29), device="cuda", generator=g))      # the graph reads the same buffers: new input, new result
        ref2 = net.model(x, want_all=False)[-1].clone()
        gr.replay()
        torch.cuda.synchronize()
        assert torch.equal(out, ref2) and not torch.equal(ref2, ref)
    finally:
        net.close()
