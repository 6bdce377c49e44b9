"""GPU parity of the training graph against oracle/fisr_train_oracle.py (torch autograd, fp64):
the backward ops one by one, then the whole four-pass loss with all 276 gradients, then one Adam step."""
import ctypes
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))


@pytest.fixture(scope="module")
def env():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from fisr_amd import lib
    return torch, lib.lib(), lib


def _dev(torch, a):
    return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to("cuda:0")


def _ptr(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else None


def _rel(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30))


@pytest.mark.parametrize("n,h,w,c0,c1,ci,co,cg,relu_in", [
    (1, 8, 32, 64, 0, 64, 64, 64, 0), (2, 13, 45, 32, 0, 29, 64, 64, 0), (1, 16, 40, 64, 64, 128, 64, 64, 1),
    (1, 24, 24, 64, 0, 64, 6, 16, 0), (3, 9, 33, 48, 0, 38, 96, 96, 1),
    # the small maps of the U-Net's lower levels (each picks another tile geometry of the weight-gradient kernel)
    (5, 3, 3, 64, 0, 64, 64, 64, 1), (3, 6, 6, 32, 32, 64, 32, 32, 0), (2, 12, 12, 64, 0, 64, 64, 64, 1), (1, 48, 48, 32, 0, 32, 64, 64, 0),
    (2, 5, 24, 32, 0, 32, 32, 32, 0),
    # the Winograd-domain kernel (W >= 24 with at most a quarter of the 32-pixel tile columns wasted): odd sizes, concat input, relu
    (2, 13, 30, 64, 0, 64, 64, 64, 1), (1, 7, 61, 32, 32, 64, 32, 32, 0), (2, 96, 96, 64, 0, 64, 64, 64, 1), (1, 3, 24, 32, 0, 32, 96, 96, 0),
    (2, 16, 8, 64, 0, 64, 32, 32, 1), (1, 30, 40, 32, 32, 64, 64, 64, 0), (3, 24, 24, 64, 0, 64, 64, 64, 1), (1, 17, 47, 64, 0, 40, 64, 64, 0),
    # the 3- / 6-channel heads (vector-ALU kernel: ragged widths, one-row images, relu on load, fewer real input channels than 64)
    (2, 20, 37, 64, 0, 64, 3, 16, 1), (3, 1, 9, 64, 0, 64, 6, 16, 0), (1, 33, 8, 64, 0, 40, 3, 16, 0), (40, 7, 19, 64, 0, 64, 6, 16, 1)])
def test_wgrad_bgrad_dgrad_vs_autograd(env, n, h, w, c0, c1, ci, co, cg, relu_in):
    torch, L, lib = env
    import torch.nn.functional as F
    r = np.random.default_rng(n * 1000 + h + w + ci + co)
    x = r.standard_normal((n, h, w, c0 + c1)).astype(np.float32)
    x[..., ci:] = 0 if ci < c0 + c1 else x[..., ci:]
    g = r.standard_normal((n, h, w, cg)).astype(np.float32)
    g[..., co:] = 0
    wt = (r.standard_normal((3, 3, ci, co)) * 0.1).astype(np.float32)
    # oracle
    xt = torch.from_numpy(x[..., :ci].astype(np.float64)).permute(0, 3, 1, 2).requires_grad_(True)
    wt_t = torch.from_numpy(wt.astype(np.float64)).permute(3, 2, 0, 1).requires_grad_(True)
    bt = torch.zeros(co, dtype=torch.float64, requires_grad=True)
    y = F.conv2d(F.relu(xt) if relu_in else xt, wt_t, bt, padding=1)
    y.backward(torch.from_numpy(g[..., :co].astype(np.float64)).permute(0, 3, 1, 2))
    dw_ref = wt_t.grad.permute(2, 3, 1, 0).numpy()
    db_ref = bt.grad.numpy()
    dx_ref = xt.grad.permute(0, 2, 3, 1).numpy()
    # GPU
    x0 = _dev(torch, x[..., :c0])
    x1 = _dev(torch, x[..., c0:]) if c1 else None
    gd = _dev(torch, g)
    dw = torch.zeros(3, 3, ci, co, device="cuda:0")
    db = torch.zeros(co, device="cuda:0")
    db2 = torch.zeros(co, device="cuda:0")
    assert L.fisr_train_wgrad(_ptr(x0), c0, _ptr(x1), c1, relu_in, _ptr(gd), cg, _ptr(dw), _ptr(db2), ci, co, n, h, w, None) == 0
    assert L.fisr_train_bgrad(_ptr(gd), cg, n * h * w, _ptr(db), co, None) == 0
    assert _rel(dw.cpu().numpy(), dw_ref) < 2e-5
    assert _rel(db.cpu().numpy(), db_ref) < 2e-5
    assert _rel(db2.cpu().numpy(), db_ref) < 2e-5          # the bias gradient fused into the weight-gradient kernel
    if cg % 16 == 0:      # data gradient = the forward conv with the transposed packing
        wd = _dev(torch, wt)
        pk = torch.empty(L.fisr_train_packed_bytes(ci, co, 1) // 4, device="cuda:0")
        assert L.fisr_train_pack(_ptr(wd), ci, co, 1, _ptr(pk), None) == 0
        zb = torch.zeros(1024, device="cuda:0")
        dx = torch.empty(n, h, w, c0 + c1, device="cuda:0")
        assert L.fisr_train_conv3x3(_ptr(gd), cg, None, 0, _ptr(pk), _ptr(zb), c0 + c1, None, _ptr(dx), n, h, w, 0, 0, 0, 0, 0, None, None) == 0
        if relu_in:
            xd = _dev(torch, x)
            assert L.fisr_train_relu_bwd(_ptr(dx), _ptr(xd), _ptr(dx), dx.numel(), None) == 0
        got = dx.cpu().numpy()
        assert _rel(got[..., :ci], dx_ref) < 2e-5
        assert np.all(got[..., ci:] == 0)


def test_elementwise_adjoints_vs_autograd(env):
    torch, L, lib = env
    import torch.nn.functional as F
    import torch_cpu as tw
    r = np.random.default_rng(5)
    n, h, w, c = 2, 6, 10, 16
    # legacy x2 bilinear
    g = r.standard_normal((n, 2 * h, 2 * w, c)).astype(np.float32)
    xt = torch.zeros(n, c, h, w, dtype=torch.float64, requires_grad=True)
    tw._up2(xt).backward(torch.from_numpy(g.astype(np.float64)).permute(0, 3, 1, 2))
    dx = torch.empty(n, h, w, c, device="cuda:0")
    assert L.fisr_train_upsample2_bwd(_ptr(_dev(torch, g)), _ptr(dx), n, h, w, c, None) == 0
    assert _rel(dx.cpu().numpy(), xt.grad.permute(0, 2, 3, 1).numpy()) < 1e-6
    # max pool
    x = r.standard_normal((n, 2 * h, 2 * w, c)).astype(np.float32)
    gp = r.standard_normal((n, h, w, c)).astype(np.float32)
    xt = torch.from_numpy(x.astype(np.float64)).permute(0, 3, 1, 2).requires_grad_(True)
    F.max_pool2d(xt, 2).backward(torch.from_numpy(gp.astype(np.float64)).permute(0, 3, 1, 2))
    dx = torch.empty(n, 2 * h, 2 * w, c, device="cuda:0")
    assert L.fisr_train_maxpool2_bwd(_ptr(_dev(torch, x)), _ptr(_dev(torch, gp)), _ptr(dx), n, 2 * h, 2 * w, c, None) == 0
    assert np.array_equal(dx.cpu().numpy(), xt.grad.permute(0, 2, 3, 1).numpy().astype(np.float32))
    # depth_to_space
    xt = torch.zeros(n, 4 * c, h, w, dtype=torch.float64, requires_grad=True)
    tw._d2s(xt).backward(torch.from_numpy(g.astype(np.float64)).permute(0, 3, 1, 2))
    out = torch.empty(n, h, w, 4 * c, device="cuda:0")
    assert L.fisr_train_s2d(_ptr(_dev(torch, g)), _ptr(out), n, h, w, c, None) == 0
    assert np.array_equal(out.cpu().numpy(), xt.grad.permute(0, 2, 3, 1).numpy().astype(np.float32))


@pytest.fixture(scope="module")
def small_case():
    import fisr_train_oracle as fo
    from fisr_amd import weights
    W = weights.synthetic_weights(2020)
    batch = fo.synthetic_batch(11, 1, 32, 32)
    loss, terms, grads = fo.loss_and_grads(W, batch)
    return W, batch, loss, terms, grads


def test_training_loss_and_all_gradients_vs_oracle(env, small_case):
    """The four weight-sharing passes, seven loss terms and every one of the 276 gradients (FISRnet.py:283-491)."""
    torch, L, lib = env
    from fisr_amd import train
    W, batch, loss_ref, terms_ref, grads_ref = small_case
    net = train.TrainNet(W)
    net.zero_grad()
    loss, terms = net.loss_and_grads(train.to_device_batch(batch))
    assert abs(loss - loss_ref) <= 2e-5 * abs(loss_ref)
    for k in ("recn", "tm", "tmm", "td", "recn_ss2", "td_ss2", "tm_ss2"):
        assert abs(terms[k] - terms_ref[k]) <= 5e-5 * abs(terms_ref[k]) + 1e-9, k
    got = net.grads_numpy()
    errs = sorted((_rel(got[k], ref), k) for k, ref in grads_ref.items())
    e = np.array([x[0] for x in errs])
    print("gradient errors: median %.2e, 95%% %.2e, worst %.2e (%s)" % (np.median(e), np.percentile(e, 95), e[-1], errs[-1][1]))
    # fp32 against fp64: a pre-activation within rounding of zero may land on the other side of a relu.  One such
    # flip (seen here: one channel of the 2x2 map of level_1/dec/level_2) moves that layer's gradients by ~1e-2 and
    # everything upstream of it by ~1e-3, while all other entries agree to 1e-8 -- so the bulk must be tight and the
    # tail bounded, not every tensor tight.
    assert np.median(e) < 5e-5
    assert np.percentile(e, 60) < 2e-4
    assert e[-1] < 5e-2, errs[-1]


def test_training_gradients_second_case_spec_weights_batch2_64px(env):
    """Another weight set (SURVEY 8d's undamped spec), batch 2, 64x64 patches: the same bounds hold, so the 1e-3 tail of
    the first case is a relu flip of that case, not an error of some layer."""
    torch, L, lib = env
    import fisr_train_oracle as fo
    from fisr_amd import train, weights
    W = weights.spec_weights(2020)
    batch = fo.synthetic_batch(5, 2, 64, 64)
    loss_ref, terms_ref, grads_ref = fo.loss_and_grads(W, batch)
    net = train.TrainNet(W)
    net.zero_grad()
    loss, terms = net.loss_and_grads(train.to_device_batch(batch))
    assert abs(loss - loss_ref) <= 5e-5 * abs(loss_ref)
    got = net.grads_numpy()
    e = np.array(sorted(_rel(got[k], ref) for k, ref in grads_ref.items()))
    print("gradient errors (spec weights, 2 x 64x64): median %.2e, 95%% %.2e, worst %.2e" % (np.median(e), np.percentile(e, 95), e[-1]))
    assert np.median(e) < 5e-5 and np.percentile(e, 60) < 2e-4 and e[-1] < 5e-2


def test_adam_step_vs_oracle(env, small_case):
    torch, L, lib = env
    import fisr_train_oracle as fo
    from fisr_amd import train
    W, batch, loss_ref, terms_ref, grads_ref = small_case
    net = train.TrainNet(W)
    net.train_step(train.to_device_batch(batch), lr=1e-4)
    Wn = {k: np.array(v, np.float64) for k, v in W.items()}
    m = {k: np.zeros_like(v) for k, v in Wn.items()}
    v = {k: np.zeros_like(x) for k, x in Wn.items()}
    fo.adam_step(Wn, grads_ref, m, v, 1, 1e-4)
    got = net.weights_numpy()
    # the first Adam step moves every weight by lr * sign(g) (up to eps): compare the updates, not the weights
    for k in Wn:
        du, dr = got[k].astype(np.float64) - W[k], Wn[k] - W[k]
        # (entries with |g| near eps / sqrt(1 - b2) = 3e-7 move by lr * g / (|g| + 3e-7): too sensitive to compare)
        big = np.abs(grads_ref[k]) > max(1e-3 * np.abs(grads_ref[k]).max(), 3e-5)
        # (a relu that flips between fp32 and fp64 changes a channel's entries wholesale: allow a percent of them)
        assert np.isclose(du[big], dr[big], rtol=5e-3, atol=2e-8).mean() > 0.99, k
    # and a second step keeps going down the same loss
    l2, _ = net.train_step(train.to_device_batch(batch), lr=1e-4)
    assert np.isfinite(l2)


def test_repeated_steps_reduce_the_loss(env):
    torch, L, lib = env
    import fisr_train_oracle as fo
    from fisr_amd import train, weights
    net = train.TrainNet(weights.synthetic_weights(2020))
    batch = train.to_device_batch(fo.synthetic_batch(3, 2, 32, 32))
    losses = [net.train_step(batch, lr=1e-4)[0] for _ in range(8)]
    print("losses", [round(v, 4) for v in losses])
    assert all(np.isfinite(losses))
    assert losses[-1] < 0.9 * losses[0]


def test_resume_restores_the_adam_state(env):
    """A checkpoint carries Adam's moments and step (the reference's Saver stores the slots and beta powers, FISRnet.py:585):
    2 steps + save + load + 1 step must equal 3 steps in a row; without the state the third update differs."""
    torch, L, lib = env
    import fisr_train_oracle as fo
    from fisr_amd import train, weights
    W = weights.synthetic_weights(2020)
    batch = train.to_device_batch(fo.synthetic_batch(3, 2, 32, 32))
    a = train.TrainNet(W)
    for _ in range(3):
        a.train_step(batch, lr=1e-4)
    b = train.TrainNet(W)
    for _ in range(2):
        b.train_step(batch, lr=1e-4)
    Wb, Sb = b.weights_numpy(), b.optimizer_state_numpy()
    c = train.TrainNet(Wb)
    c.load_optimizer_state(Sb)
    assert c.step_count == 2
    c.train_step(batch, lr=1e-4)
    cold = train.TrainNet(Wb)                                      # what the advisor's finding describes: zero moments
    cold.train_step(batch, lr=1e-4)
    Wa, Wc, Wcold = a.weights_numpy(), c.weights_numpy(), cold.weights_numpy()
    k = "FISRnet/level_3/enc/level_0/conv/0/w"
    # mean |difference| over all weights relative to the mean third update (the weight-gradient kernel's atomics make a run not
    # bit-reproducible, and Adam's g / sqrt(v) turns rounding noise on near-zero gradients into a visible fraction of lr on a
    # few elements: means, not maxima)
    num = lambda A, B: sum(float(np.abs(A[n] - B[n]).sum()) for n in A)
    upd = num(Wa, Wb)
    warm, cold_ = num(Wa, Wc) / upd, num(Wa, Wcold) / upd
    print(f"resume: |3 steps - (2 + restore + 1)| / |third update| = {warm:.4f}; without the Adam state {cold_:.4f}")
    assert warm < 0.03 and cold_ > 0.3
    assert float(np.abs(Wa[k] - Wc[k]).mean()) < 0.05 * float(np.abs(Wa[k] - Wb[k]).mean())


def test_phase_train_cli_writes_a_checkpoint_the_inference_engine_loads(env, tmp_path):
    """main.py --phase train (FISRnet.py:583-745) on synthetic samples: two epochs, validation, checkpoint; then the
    inference engine restores that checkpoint the way FISRnet.load does."""
    torch, L, lib = env
    from fisr_amd import main, weights
    from fisr_amd.fisrnet import FISRnet
    d = str(tmp_path)
    argv = ["--phase", "train", "--synthetic_train", "6", "--epoch", "2", "--batch_size", "2", "--val_data_size", "2",
            "--val_batch_size", "2", "--freq_display", "1", "--lr_type", "no_decay", "--save_tf_bundle",
            "--checkpoint_dir", d + "/ck", "--text_dir", d + "/tx", "--log_dir", d + "/lg", "--test_img_dir", d + "/ti"]
    assert main.main(argv) == 0
    path, kind, step = weights.find_checkpoint(d + "/ck", "FISRnet_exp1")
    assert kind in ("npz", "tf_bundle") and step == 4
    W = weights.load_weights(path, kind)
    from fisr_amd import tf_bundle                                  # the TF checkpoint-V2 twin holds the same tensors
    Wb = tf_bundle.read_bundle(os.path.join(d, "ck", "FISRnet_exp1", "FISRnet-4"), name_filter="FISRnet")
    assert all(np.array_equal(Wb[k], W[k]) for k in W)
    gs = tf_bundle.read_bundle(os.path.join(d, "ck", "FISRnet_exp1", "FISRnet-4"))["Variable"]      # the reference's global step (FISRnet.py:232)
    assert gs.dtype == np.int32 and gs.shape == () and int(gs) == 4
    W0 = weights.xavier_weights(1)                                  # the from-scratch initialiser (ops.py:8-9), seed = exp_num
    assert any(not np.array_equal(W[k], W0[k]) for k in W)
    # Saver(max_to_keep=1): only the last step's files are left; they hold the Adam slots and beta powers
    left = sorted(f for f in os.listdir(os.path.join(d, "ck", "FISRnet_exp1")) if f != "checkpoint")
    assert left == ["FISRnet-4.data-00000-of-00001", "FISRnet-4.index", "FISRnet-4.npz"], left
    st = weights.load_optimizer_state(path, kind)
    assert st is not None and len(st) == 2 * 276 + 2
    assert abs(float(st["beta2_power"]) - 0.999 ** 5) < 1e-6 and any(np.abs(v).max() > 0 for k, v in st.items() if k.endswith("/Adam"))
    stb = weights.load_optimizer_state(os.path.join(d, "ck", "FISRnet_exp1", "FISRnet-4"), "tf_bundle")
    assert stb is not None and all(np.array_equal(stb[k], st[k]) for k in st if "beta" not in k)
    # resuming for one more epoch continues from step 4 with that state
    assert main.main(argv[:4] + ["--epoch", "3"] + argv[6:]) == 0
    assert weights.find_checkpoint(d + "/ck", "FISRnet_exp1")[2] == 6
    # ... and the checkpoint the run was RESTORED from stays (tf.train.Saver only prunes what it saved itself: restore() does not
    # register the loaded files -- a user's pretrained bundle must survive a fine-tuning epoch)
    left = sorted(f for f in os.listdir(os.path.join(d, "ck", "FISRnet_exp1")) if f != "checkpoint")
    assert left == sorted(f"FISRnet-{s_}{e}" for s_ in (4, 6) for e in (".data-00000-of-00001", ".index", ".npz")), left
    net = FISRnet(device="cuda:0", precision="fp32")
    net.set_weights(W)
    x = torch.rand(1, 32, 32, 29, device="cuda:0")
    p1, p2, p3 = net.model(x)
    assert torch.isfinite(p3).all()
    net.close()


def test_winograd_training_conv_splits_a_batch_beyond_32_bit_offsets(env):
    """train.py allocates no direct-kernel pack for layers that have Winograd slabs, and the Winograd kernel addresses
    one launch's batch with 32-bit byte offsets -- a batch beyond 4 GiB used to fail the step with FISR_EINVAL.  Now the entry
    launches it in chunks of images: 1900 images of 96 x 96 x 64 (4.5 GB in, 4.5 GB out) through the Winograd-only call, equal bit
    for bit to the same images run as two calls that fit."""
    torch, L, lib = env
    n, h, w, c = 1900, 96, 96, 64
    assert n * h * w * c * 4 > 2 ** 32
    g = torch.Generator(device="cuda:0").manual_seed(3)
    x = torch.randn((n, h, w, c), device="cuda:0", generator=g)
    wt = torch.randn((3, 3, c, c), device="cuda:0", generator=g) * 0.05
    bias = torch.randn(1024, device="cuda:0", generator=g) * 0.1
    pkw = torch.empty(L.fisr_train_wino_bytes(c, c, 0) // 4, device="cuda:0")
    assert L.fisr_train_pack_wino(_ptr(wt), c, c, 0, _ptr(pkw), None) == 0
    y = torch.empty((n, h, w, c), device="cuda:0")
    assert L.fisr_train_conv3x3(_ptr(x), c, None, 0, None, _ptr(bias), c, None, _ptr(y), n, h, w, 1, 0, 0, 0, 0, _ptr(pkw), None) == 0
    torch.cuda.synchronize()
    half = n // 2
    y2 = torch.empty((half, h, w, c), device="cuda:0")
    for k in (0, half):
        assert L.fisr_train_conv3x3(_ptr(x[k:k + half]), c, None, 0, None, _ptr(bias), c, None, _ptr(y2), half, h, w, 1, 0, 0, 0, 0, _ptr(pkw), None) == 0
        torch.cuda.synchronize()
        assert torch.equal(y[k:k + half], y2), k
    assert float(y.abs().max()) > 0.5


def _dp_worker(rank, world, port, q):
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        sys.path.insert(0, os.path.join(ROOT, "oracle"))
        import fisr_train_oracle as fo
        from fisr_amd import train, weights
        torch.cuda.set_device(0)
        W = weights.synthetic_weights(2020)
        full = fo.synthetic_batch(21, world, 32, 32)
        net = train.TrainNet(W)
        net.grad_scale = 1.0 / world
        shard = train.to_device_batch({k: v[rank::world] for k, v in full.items()})
        net.train_step(shard, lr=1e-4)                       # loss_and_grads on the shard, all-reduce, Adam
        got = net.weights_numpy()
        ok, worst = True, 0.0
        if rank == 0:
            ref = train.TrainNet(W)                          # the whole batch in this process alone: no collective
            ref.zero_grad()
            ref.loss_and_grads(train.to_device_batch(full))
            ref.adam_step(1e-4)
            exp = ref.weights_numpy()
            for k in exp:
                du, dr = got[k].astype(np.float64) - W[k], exp[k].astype(np.float64) - W[k]
                frac = float(np.isclose(du, dr, rtol=2e-3, atol=2e-8).mean())
                worst = max(worst, 1.0 - frac)
                ok = ok and frac > 0.995
        q.put((rank, ok, worst))
    except Exception:  # noqa: BLE001  (a silent worker death would leave the parent waiting for the queue)
        import traceback
        q.put((rank, False, traceback.format_exc()[-3000:]))
    finally:
        try:
            dist.barrier()                                   # nobody leaves while a peer may still be receiving
        except Exception:  # noqa: BLE001
            pass
        dist.destroy_process_group()


def test_data_parallel_step_equals_the_full_batch_step(env):
    """Two ranks (gloo processes sharing cuda:0; RCCL on a real node) each take half of a batch, all-reduce the flat
    gradient buffer and apply Adam: the weights must move as in the single-process step on the whole batch."""
    torch, L, lib = env
    import socket
    import torch.multiprocessing as mp
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_dp_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    print("data-parallel vs full batch:", sorted(res))
    for r_ in res:
        if not r_[1]:
            print(r_[2])
    assert all(ok for _, ok, _ in res), res


@pytest.mark.parametrize("transpose,flags,with_res", [(0, 3, True), (1, 0, False), (0, 7, False)])
def test_train_conv_winograd_path_equals_direct_path(env, transpose, flags, with_res):
    """Large dense layers of the training forward / data-gradient pass run on the persistent Winograd kernel with
    device-packed slabs (fisr_train_pack_wino); the same call without the slabs runs the direct kernel."""
    torch, L, lib = env
    r = np.random.default_rng(17 + transpose)
    n, h, w, ci, co = (8 if transpose else 4), 64, 256, 64, 128        # >= 512 work items either way
    wt = _dev(torch, (r.standard_normal((3, 3, ci, co)) * 0.05).astype(np.float32))
    cin, cout = (co, ci) if transpose else (ci, co)
    x = _dev(torch, r.standard_normal((n, h, w, cin)).astype(np.float32))
    bias = _dev(torch, r.standard_normal(1024).astype(np.float32) * 0.1)
    d2s = bool(flags & 4)
    res = _dev(torch, r.standard_normal((n, h, w, cout)).astype(np.float32)) if with_res else None
    pk = torch.empty(L.fisr_train_packed_bytes(ci, co, transpose) // 4, device="cuda:0")
    nb = L.fisr_train_wino_bytes(ci, co, transpose)
    assert nb > 0
    pkw = torch.empty(nb // 4, device="cuda:0")
    assert L.fisr_train_pack(_ptr(wt), ci, co, transpose, _ptr(pk), None) == 0
    assert L.fisr_train_pack_wino(_ptr(wt), ci, co, transpose, _ptr(pkw), None) == 0
    shape = (n, 2 * h, 2 * w, cout // 4) if d2s else (n, h, w, cout)
    y0, y1 = torch.empty(shape, device="cuda:0"), torch.empty(shape, device="cuda:0")
    for y, slabs in ((y0, None), (y1, pkw)):
        assert L.fisr_train_conv3x3(_ptr(x), cin, None, 0, _ptr(pk), _ptr(bias), cout, _ptr(res), _ptr(y), n, h, w, flags,
                                    0, 0, 0, 0, _ptr(slabs), None) == 0
    a, b = y0.cpu().numpy(), y1.cpu().numpy()
    assert np.abs(a).max() > 0.5
    assert np.abs(a - b).max() <= 2e-5 * max(1.0, np.abs(a).max())
    assert not np.array_equal(a, b)          # (two different algorithms: identical bits would mean the slabs were ignored)


def test_pack_all_equals_the_single_packs(env):
    """fisr_train_pack_all (every layout of every conv in one launch, what the training step uses) writes the same bytes as
    fisr_train_pack / fisr_train_pack_wino conv by conv; layouts a conv is not eligible for are skipped (null destination)."""
    torch, L, lib = env
    r = np.random.default_rng(5)
    shapes = [(29, 64), (64, 64), (64, 48), (128, 256), (64, 3), (64, 6), (16, 64), (512, 256)]
    keep, rows = [], []
    for ci, co in shapes:
        w = _dev(torch, r.standard_normal((3, 3, ci, co)).astype(np.float32))
        bufs = []
        for tr in (0, 1):
            bufs.append(torch.full((L.fisr_train_packed_bytes(ci, co, tr) // 4,), 7.0, device="cuda:0"))
        for tr in (0, 1):
            nb = L.fisr_train_wino_bytes(ci, co, tr)
            bufs.append(torch.full((nb // 4,), 7.0, device="cuda:0") if nb else None)
        keep.append((w, bufs))
        rows.append((w.data_ptr(),) + tuple(b.data_ptr() if b is not None else 0 for b in bufs) + (ci, co))
    descs = np.array(rows, dtype=[("w", "u8"), ("pk", "u8"), ("pk_t", "u8"), ("pkw", "u8"), ("pkw_t", "u8"), ("ci", "i4"), ("co", "i4")])
    assert descs.itemsize == 48
    table = torch.from_numpy(descs.view(np.uint8).copy()).to("cuda:0")
    assert L.fisr_train_pack_all(_ptr(table), len(shapes), None) == 0
    torch.cuda.synchronize()
    n_wino = 0
    for (ci, co), (w, bufs) in zip(shapes, keep):
        for k, b in enumerate(bufs):
            if b is None:
                continue
            ref = torch.full_like(b, 7.0)
            if k < 2:
                assert L.fisr_train_pack(_ptr(w), ci, co, k, _ptr(ref), None) == 0
            else:
                assert L.fisr_train_pack_wino(_ptr(w), ci, co, k - 2, _ptr(ref), None) == 0
                n_wino += 1
            assert torch.equal(b, ref), (ci, co, k)
    assert n_wino >= 8
    assert L.fisr_train_pack_all(None, 1, None) < 0


@pytest.mark.parametrize("n,h,w,ci,co", [(32, 3, 3, 256, 128), (8, 6, 6, 128, 128), (4, 12, 12, 64, 64), (2, 24, 24, 32, 64)])
def test_train_conv_winograd_on_small_maps(env, n, h, w, ci, co):
    """The training step sends every eligible layer to the Winograd kernel, the 3 x 3 ... 24 x 24 maps of the lower U-Net
    levels included (one mostly empty 8 x 32 tile per image): same result as the direct kernel."""
    torch, L, lib = env
    r = np.random.default_rng(h)
    wt = _dev(torch, (r.standard_normal((3, 3, ci, co)) * 0.05).astype(np.float32))
    x = _dev(torch, r.standard_normal((n, h, w, ci)).astype(np.float32))
    bias = _dev(torch, r.standard_normal(1024).astype(np.float32) * 0.1)
    res = _dev(torch, r.standard_normal((n, h, w, co)).astype(np.float32))
    pk = torch.empty(L.fisr_train_packed_bytes(ci, co, 0) // 4, device="cuda:0")
    pkw = torch.empty(L.fisr_train_wino_bytes(ci, co, 0) // 4, device="cuda:0")
    assert pkw.numel() > 0
    assert L.fisr_train_pack(_ptr(wt), ci, co, 0, _ptr(pk), None) == 0
    assert L.fisr_train_pack_wino(_ptr(wt), ci, co, 0, _ptr(pkw), None) == 0
    y0, y1 = torch.empty((n, h, w, co), device="cuda:0"), torch.empty((n, h, w, co), device="cuda:0")
    for y, slabs in ((y0, None), (y1, pkw)):
        assert L.fisr_train_conv3x3(_ptr(x), ci, None, 0, _ptr(pk), _ptr(bias), co, _ptr(res), _ptr(y), n, h, w, 3, 0, 0, 0, 0, _ptr(slabs), None) == 0
    a, b = y0.cpu().numpy(), y1.cpu().numpy()
    assert np.abs(a).max() > 0.5
    assert np.abs(a - b).max() <= 2e-5 * max(1.0, np.abs(a).max())
    assert not np.array_equal(a, b)
