"""GPU parity on MORE THAN ONE weight set (run with `-m gpu`): the whole forward of every shipped engine against
the fp64 C oracle with the PSNR protocol of SURVEY.md 8c-ii, on
  * "survey_spec": the undamped Glorot-0.8 set SURVEY.md 8d specifies (only the conv/2 heads rescaled),
  * "harsh":       He-initialised, residual branches damped by 0.6 only: decoder activations ~1.5 rms, predictions
                   swinging beyond [0,1] -- rounding errors are NOT attenuated by a dominant identity path,
in addition to the "default" set every other test uses.  No checkpoint ships with the reference, so these stand in
for "any checkpoint": the tolerance that matters is north_star's +-0.02 dB / 1e-3 SSIM.
"""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")

import c_oracle as C          # noqa: E402
import fisr_oracle as O       # noqa: E402
from fisr_amd import weights  # noqa: E402
from fisr_amd.fisrnet import FISRnet  # noqa: E402

# engine -> (max |err| relative to the largest |oracle value| of the level, allowed PSNR shift in dB)
ENGINES = {"fp32d": (3e-5, 2e-4), "fp32": (1e-4, 5e-4), "fp32w4": (1e-4, 5e-4), "bf16x3": (5e-4, 0.004), "f16f8": (2e-3, 0.01),
           "mixed": (1e-2, 0.008)}     # (levels 1 and 2 of the mixed engine are plain fp16: the relative bound is fp16's)


def _psnr_shift(hip, oracle, rng):
    out = []
    for sl, db in ((slice(0, 3), 37.86), (slice(3, 6), 48.07), (slice(6, 9), 37.86)):
        sigma = 10 ** (-db / 20)
        o = np.clip(oracle[..., sl], 0, 1)
        gt = o + rng.standard_normal(o.shape) * sigma
        out.append(abs(O.compute_psnr(gt, np.clip(hip[..., sl], 0, 1)) - O.compute_psnr(gt, o)))
    return max(out)


@pytest.fixture(scope="module", params=["survey_spec", "harsh"])
def wset(request):
    W = weights.WEIGHT_SETS[request.param]()
    C.build()
    return request.param, W, C.pack_blob(W)


@pytest.mark.parametrize("engine", list(ENGINES))
def test_forward_other_weight_sets_vs_oracle(wset, engine):
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    name, W, blob = wset
    from tests_support import make_full_size_input
    net = FISRnet(device="cuda:0", precision=engine)
    net.set_weights(W)
    rel_tol, db_tol = ENGINES[engine]
    rng = np.random.default_rng(31)
    try:
        for seed, (n, h, w) in ((11, (1, 64, 96)), (12, (2, 96, 160))):
            x = make_full_size_input(seed, h, w, n)
            ref = C.forward(x, blob, True)
            outs = net.model(torch.from_numpy(x).cuda())
            torch.cuda.synchronize()
            for lvl, got, exp in zip(("pred_l1", "pred_l2", "pred_l3"), outs, ref):
                g = got.cpu().numpy().astype(np.float64)
                scale = max(1.0, float(np.abs(exp).max()))
                err = float(np.abs(g - exp).max())
                shift = _psnr_shift(g, exp, rng)
                ssim = min(O.ssim_pil(O.quantize_u8(np.clip(g[b, ..., 3 * f:3 * f + 3], 0, 1)),
                                      O.quantize_u8(np.clip(exp[b, ..., 3 * f:3 * f + 3], 0, 1)))
                           for b in range(n) for f in range(3)) if min(g.shape[1:3]) >= 7 else 1.0
                print(f"{name} {engine} {lvl} n{n} {h}x{w}: max|err| {err:.3e} (scale {scale:.2f}) dPSNR {shift:.2e} dB  SSIM(hip,oracle) {ssim:.6f}")
                assert err <= rel_tol * scale, (name, engine, lvl, err, scale)
                # (the mixed engine runs levels 1 and 2 -- inputs of level 3, not outputs of the network -- in plain fp16)
                lvl_tol = 0.02 if engine == "mixed" and lvl != "pred_l3" else db_tol
                assert shift <= lvl_tol <= 0.02, (name, engine, lvl, shift)
                assert 1.0 - ssim <= 1e-3, (name, engine, lvl, ssim)
    finally:
        net.close()


def test_full_size_tile_second_scene_spec_weights(gold_dir):
    """A SECOND scene and weight set at full tile size: one 544x992x29 tile with spec_weights(2020) against the fp64
    oracle values committed on a sparse grid (tests/golden/model_544x992_sparse_spec.npz, oracle/make_golden_fullsize_b.py)."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    path = os.path.join(gold_dir, "model_544x992_sparse_spec.npz")
    if not os.path.isfile(path):
        pytest.skip("second sparse full-size fixture not generated")
    g = np.load(path)
    from tests_support import make_full_size_input
    x = torch.from_numpy(make_full_size_input(int(g["seed"]), 544, 992)).cuda()
    st = int(g["stride"])
    exp = g["l3_sparse"].astype(np.float64)
    W = weights.spec_weights(2020)
    rng = np.random.default_rng(5)
    for engine, (rel_tol, db_tol) in ENGINES.items():
        net = FISRnet(device="cuda:0", precision=engine)
        net.set_weights(W)
        _, _, l3 = net.model(x, want_all=False)
        got = l3[0, ::st, ::st, :].cpu().numpy().astype(np.float64)
        net.close()
        del l3
        torch.cuda.empty_cache()
        err = float(np.abs(got - exp).max())
        shift = _psnr_shift(got, exp, rng)
        print(f"spec weights, scene 2, {engine}: max|err| {err:.3e} dPSNR {shift:.2e} dB")
        assert err <= rel_tol * max(1.0, float(np.abs(exp).max())), (engine, err)
        assert shift <= db_tol, (engine, shift)


def test_full_size_ten_scenes_fast_engines_vs_fp32(syn_weights):
    """cfg3 of BASELINE.json at FULL size (the data set is absent: ten synthetic 1080x1920 scenes, crop 1024x1920, the
    reference's 2x2 tiles with 32-px halo, 3 windows each = 30 windows of 2048x3840x9), engine level (no PNG files, so
    the test stays in seconds): the exact-fp32 engine is the reference, pseudo ground truth = its prediction + Gaussian
    noise at the published operating point (README.md:97); the shipped fp32 (Winograd) engine and the split-precision
    engines must score within +-0.02 dB PSNR / 1e-3 SSIM of it on every window (PSNR: 3 YUV channels of a frame jointly,
    FISRnet.py:883-889; SSIM: the SSIM_PIL restatement on the uint8 frames, on the GPU)."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import math
    dev = torch.device("cuda:0")
    ref = FISRnet(device="cuda:0", precision="fp32d")
    ref.set_weights(syn_weights)
    engines = {}
    for p in ("fp32", "bf16x3", "f16f8", "mixed"):
        engines[p] = FISRnet(device="cuda:0", precision=p)
        engines[p].set_weights(syn_weights)
    gen = torch.Generator(device=dev).manual_seed(2024)
    worst = {p: [0.0, 0.0] for p in engines}
    sig = (10 ** (-37.86 / 20), 10 ** (-48.07 / 20), 10 ** (-37.86 / 20))
    for sc in range(10):
        # a 5-frame scene: smooth random content moving by a few pixels per frame, smooth flows, GPU-warped neighbours
        coarse = torch.rand((1, 3, 1080 // 8 + 8, 1920 // 8 + 8), generator=gen, device=dev)
        big = torch.nn.functional.interpolate(coarse, scale_factor=8, mode="bilinear", align_corners=False)[0].permute(1, 2, 0)
        frames = []
        for k in range(5):
            f = big[8 + 3 * k:8 + 3 * k + 1080, 16 + 5 * k:16 + 5 * k + 1920] + 0.02 * torch.randn((1080, 1920, 3), generator=gen, device=dev)
            frames.append((f.clamp(0, 1) * 255).to(torch.uint8).contiguous())
        fc = torch.randn((8, 2, 1080 // 32 + 2, 1920 // 32 + 2), generator=gen, device=dev) * 4
        flows = torch.nn.functional.interpolate(fc, scale_factor=32, mode="bilinear", align_corners=False)[:, :, :1080, :1920].permute(0, 2, 3, 1).contiguous()
        warps = []
        for p in range(4):
            warps.append(ref.warp(frames[p + 1], flows[2 * p]))
            warps.append(ref.warp(frames[p], flows[2 * p + 1]))
        for s in range(3):
            inp = ref.pack_input(frames[s:s + 3], list(flows[2 * s:2 * s + 4]), warps[2 * s:2 * s + 4], 1024, 1920)
            base = ref.forward_tiled(inp, (2, 2)).clamp(0, 1)
            noise = torch.randn(base.shape, generator=gen, device=dev)
            gt = base.clone()
            for f in range(3):
                gt[..., 3 * f:3 * f + 3] += sig[f] * noise[..., 3 * f:3 * f + 3]
            gt_u8 = (gt.clamp(0, 1) * 255).to(torch.uint8)
            gtf = gt_u8.float() / 255

            def score(pred):
                pred = pred.clamp(0, 1)
                yuv, _ = ref.unpack_output(pred, want_rgb=False)
                out = []
                for f in range(3):
                    mse = float(((pred[..., 3 * f:3 * f + 3] - gtf[..., 3 * f:3 * f + 3]).double() ** 2).mean())
                    out.append((10 * math.log10(1.0 / mse), ref.ssim_u8(yuv, gt_u8, coff=3 * f)))
                return out

            s_ref = score(base)
            for p, eng in engines.items():
                got = score(eng.forward_tiled(inp, (2, 2)))
                for (pa, sa), (pb, sb) in zip(got, s_ref):
                    worst[p][0] = max(worst[p][0], abs(pa - pb))
                    worst[p][1] = max(worst[p][1], abs(sa - sb))
            del base, gt, gtf, noise
        torch.cuda.empty_cache()
    for p, (dp, ds) in worst.items():
        print(f"10 scenes x 3 windows x 3 frames at 2048x3840, {p} vs exact fp32: max |dPSNR| {dp:.2e} dB, max |dSSIM| {ds:.2e}")
        assert dp <= 0.02 and ds <= 1e-3, (p, dp, ds)
    assert 36.5 < s_ref[0][0] < 39 and 46 < s_ref[1][0] < 50.5      # the ground truth sits at the published operating point
    ref.close()
    for e in engines.values():
        e.close()


def test_two_by_four_tile_plan_vs_the_reference_two_by_two_seams(syn_weights):
    """the 8-GPU tile plan 1 window x 2 x 4 tiles moves the seams away from where the reference's 2 x 2 plan
    (FISRnet.py:847-880, utils.py:118-159) has them, "so parity (+-0.02 dB) must be re-verified".  Measured here, on one full-size
    window (1024 x 1920 -> 2048 x 3840 x 9, shipped fp32 engine, all plans in one process, pseudo ground truth = a plan's own
    prediction + Gaussian noise at the published operating point, README.md:97):

      * what the reference's OWN seams cost on these weights: 2 x 2 plan against the untiled forward (--test_patch 1,1);
      * what the 2 x 4 plan costs against the 2 x 2 plan.

    With the synthetic (untrained) weights the network's coarse levels carry every tile's content to every pixel (all 3840
    columns of the two plans differ), so NEITHER comparison is inside 0.02 dB on the 48-dB SR frame: r04 measured 0.32 dB for
    2 x 4 vs 2 x 2.  That is a property of tiling with weights that were never trained to be local, not of the plan -- which is
    why the default 8-GPU topology is 2 windows x the reference's 2 x 2 seams (dist.TileTopology), bit-identical to the
    single-GPU result, and the 2 x 4 plan is opt-in.  What the test pins: the 2 x 4 plan's seams cost no more than twice what the
    reference's own seams cost on the same weights (+ the 0.02 dB of the tolerance), and the FI-SR frames (38-dB operating
    point) stay inside 0.05 dB.  With `checkpoint_dir/FISRnet_exp1` the +-0.02 dB question is one `--phase test
    --test_patch 2,4 --check_published` run."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import math
    dev = torch.device("cuda:0")
    net = FISRnet(device="cuda:0", precision="fp32")
    net.set_weights(syn_weights)
    gen = torch.Generator(device=dev).manual_seed(77)
    coarse = torch.rand((1, 3, 1080 // 8 + 8, 1920 // 8 + 8), generator=gen, device=dev)
    big = torch.nn.functional.interpolate(coarse, scale_factor=8, mode="bilinear", align_corners=False)[0].permute(1, 2, 0)
    frames = []
    for k in range(3):
        f = big[8 + 3 * k:8 + 3 * k + 1080, 16 + 5 * k:16 + 5 * k + 1920] + 0.02 * torch.randn((1080, 1920, 3), generator=gen, device=dev)
        frames.append((f.clamp(0, 1) * 255).to(torch.uint8).contiguous())
    fc = torch.randn((4, 2, 1080 // 32 + 2, 1920 // 32 + 2), generator=gen, device=dev) * 4
    flows = torch.nn.functional.interpolate(fc, scale_factor=32, mode="bilinear", align_corners=False)[:, :, :1080, :1920].permute(0, 2, 3, 1).contiguous()
    warps = [net.warp(frames[1], flows[0]), net.warp(frames[0], flows[1]), net.warp(frames[2], flows[2]), net.warp(frames[1], flows[3])]
    inp = net.pack_input(frames, list(flows), warps, 1024, 1920)
    try:
        p11 = net.forward_tiled(inp, (1, 1)).clamp(0, 1)
        p22 = net.forward_tiled(inp, (2, 2)).clamp(0, 1)
        p24 = net.forward_tiled(inp, (2, 4)).clamp(0, 1)
        torch.cuda.synchronize()
        assert p11.shape == p22.shape == p24.shape == (2048, 3840, 9)
        sig = (10 ** (-37.86 / 20), 10 ** (-48.07 / 20), 10 ** (-37.86 / 20))
        noise = torch.randn(p22.shape, generator=gen, device=dev)

        def shifts(ref, other):
            out = []
            for f in range(3):
                sl = slice(3 * f, 3 * f + 3)
                gt = ref[..., sl] + sig[f] * noise[..., sl]
                ps = [10 * math.log10(1.0 / float(((p[..., sl] - gt).double() ** 2).mean())) for p in (ref, other)]
                out.append(abs(ps[0] - ps[1]))
            return out
        own = shifts(p11, p22)            # the reference's seams vs no seams
        new = shifts(p22, p24)            # the 8-GPU plan's seams vs the reference's
        d = (p22 - p24).abs().amax(dim=2)
        print(f"dPSNR per frame (FI-SR, SR, FI-SR): 2x2 plan vs untiled {[round(v, 4) for v in own]} dB; 2x4 plan vs 2x2 plan {[round(v, 4) for v in new]} dB; "
              f"max |2x4 - 2x2| {float(d.max()):.3e}, columns that differ {int((d.amax(dim=0) > 0).sum())} of 3840")
        for f in range(3):
            assert new[f] <= 2.0 * own[f] + 0.02, (f, new, own)
        assert max(new[0], new[2]) <= 0.05 and float(d.max()) < 0.1
    finally:
        net.close()
