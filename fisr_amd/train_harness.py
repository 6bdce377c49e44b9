"""`main.py --phase train`: the reference's FISRnet.build_model data handling and FISRnet.train loop
(FISRnet.py:175-224, 583-745) around fisr_amd/train.py.

Same flags, same data files (MATLAB-v7.3 `.mat` through fisr_amd/hdf5_min.py or h5py, 5-D `.flo`), the same train /
validation split, learning-rate schedules, status prints and checkpoint naming; TensorBoard summaries are out of scope
(SURVEY.md 2).  The reference's pre-made training set (10 086 samples of 96x96x5 frames) is not in its tree, so
`--synthetic_train N` substitutes N seeded samples in the same value ranges to exercise the phase end to end.
"""
from __future__ import annotations

import math
import os
import time

import numpy as np

from . import io as fio
from . import weights as _weights


def read_mat_5d(path, key):
    """utils.py:29-43: dataset [N, N_seq, C, W, H] on disk -> float32 [N, N_seq, H, W, C] / 255."""
    try:
        import h5py
    except ImportError:
        from . import hdf5_min
        a = hdf5_min.read_dataset(path, key)
    else:
        with h5py.File(path, "r") as f:
            a = np.array(f[key])
    return np.swapaxes(np.asarray(a, np.float32) / 255.0, 2, 4)


def synthetic_train_set(n, patch=96, seed=0):
    """Stand-in for the reference's pre-made training files: smooth random frames so that the network has something to
    fit; flows small, warped frames near the frames."""
    r = np.random.default_rng(seed)
    f32 = np.float32
    hr = r.random((n, 7, 2 * patch // 8, 2 * patch // 8, 3), dtype=f32)
    hr = np.repeat(np.repeat(hr, 8, axis=2), 8, axis=3)                      # blocky "content"
    hr = np.clip(hr + r.normal(0, 0.02, hr.shape).astype(f32), 0, 1)
    lr = hr[:, ::2, ::2, ::2]                                                # 4 of the 7 HR frames ... 5 needed:
    lr5 = np.concatenate([hr[:, 0:1], hr[:, 2:3], hr[:, 3:4], hr[:, 4:5], hr[:, 6:7]], axis=1)[:, :, ::2, ::2]
    del lr
    return dict(data=fio.merge_seq_dim(lr5), label=fio.merge_seq_dim(hr),
                flow=(r.normal(0, 0.01, (n, patch, patch, 16))).astype(f32), flow_ss2=(r.normal(0, 0.02, (n, patch, patch, 8))).astype(f32),
                warp=np.clip(np.repeat(fio.merge_seq_dim(lr5)[..., 3:15], 2, axis=3), 0, 1).astype(f32),
                warp_ss2=np.clip(np.repeat(fio.merge_seq_dim(lr5)[..., 3:9], 2, axis=3), 0, 1).astype(f32))


def load_train_set(args):
    """FISRnet.py:177-209."""
    if getattr(args, "synthetic_train", 0):
        return synthetic_train_set(args.synthetic_train, seed=args.exp_num)
    print(" Start to read 4K data.")
    data = fio.merge_seq_dim(read_mat_5d(args.train_data_path, "LR_data"))
    label = fio.merge_seq_dim(read_mat_5d(args.train_label_path, "HR_data"))
    print(" Successfully load.")
    print(" Start to read flow data.")
    flow = fio.merge_seq_dim(fio.read_flo_file_5dim(args.train_flow_data_path)) / data.shape[1] / 2          # :198
    flow_ss2 = fio.merge_seq_dim(fio.read_flo_file_5dim(args.train_flow_ss2_data_path)) / data.shape[1] / 2  # :203
    print(" Successfully load.")
    print(" Start to read warped data.")
    warp = fio.merge_seq_dim(fio.read_warp_file(args.train_warped_data_path) / 255.0)
    warp_ss2 = fio.merge_seq_dim(fio.read_warp_file(args.train_wapred_ss2_data_path) / 255.0)
    print(" Successfully load.")
    return dict(data=data, label=label, flow=flow.astype(np.float32), flow_ss2=flow_ss2.astype(np.float32), warp=warp, warp_ss2=warp_ss2)


def learning_rate(args, epoch, step, train_iter):
    """FISRnet.py:227-245, 637-640."""
    if args.lr_type == "stair_decay":
        k = sum(1 for p in args.lr_stair_decay_points if step > p * train_iter)       # tf.train.piecewise_constant: values[i] while x <= boundaries[i]
        return args.init_lr * args.lr_decreasing_factor ** k
    if args.lr_type == "linear_decay":
        return args.init_lr if epoch < args.lr_linear_decay_point else args.init_lr * (args.epoch - epoch) / (args.epoch - args.lr_linear_decay_point)
    return args.init_lr


def _ovlp(p3):
    """Groups2Ovlp (ops.py:119-144) on three [B,H,W,9] predictions -> [B,7,H,W,3]."""
    import torch
    f = [p[..., 3 * i:3 * i + 3] for p in p3 for i in range(3)]
    return torch.stack([f[0], f[1], (f[2] + f[3]) / 2, f[4], (f[5] + f[6]) / 2, f[7], f[8]], dim=1)


def _psnr(pred7, label21):
    """tf.reduce_mean(tf.image.psnr(pred, gt, max_val=1.0)) over [B,7] images (FISRnet.py:486-487)."""
    import torch
    b, _, h, w, _ = pred7.shape
    gt = label21.reshape(b, h, w, 7, 3).permute(0, 3, 1, 2, 4)
    mse = ((pred7 - gt) ** 2).mean(dim=(2, 3, 4))
    return float((10.0 * torch.log10(1.0 / mse)).mean())


def run_train(args):
    """FISRnet.train (FISRnet.py:583-745).  Returns the last epoch's mean total loss."""
    import torch
    import torch.distributed as dist
    from . import train as ft
    dev = args.device or "cuda:0"
    # data parallel under `python -m torch.distributed.run --nproc-per-node N -m fisr_amd.main --phase train ...`: every
    # rank takes batch_size / N samples of each batch, the gradients are all-reduced (RCCL), every rank applies Adam
    world = dist.get_world_size() if dist.is_initialized() else 1
    rank = dist.get_rank() if dist.is_initialized() else 0
    if args.batch_size % world:
        raise ValueError(f"--batch_size {args.batch_size} must be a multiple of the {world} ranks")
    if rank:
        import builtins
        print = lambda *a, **k: None  # noqa: A001,E731  (rank 0 reports)
    else:
        import builtins
        print = builtins.print  # noqa: A001
    d = load_train_set(args)
    n = d["data"].shape[0]
    nv = min(args.val_data_size, max(0, n - args.batch_size))
    tr = {k: v[:n - nv] for k, v in d.items()}
    va = {k: v[n - nv:] for k, v in d.items()}
    train_iter = (n - nv) // args.batch_size
    val_iter = nv // args.val_batch_size
    model_dir = f"FISRnet_exp{args.exp_num}"
    ckpt_dir = os.path.join(args.checkpoint_dir, model_dir)
    os.makedirs(ckpt_dir, exist_ok=True)
    path, kind, counter = _weights.find_checkpoint(args.checkpoint_dir, model_dir)
    opt_state = None
    if path:
        W = _weights.load_weights(path, kind)
        opt_state = _weights.load_optimizer_state(path, kind)
        print(" [*] Load SUCCESS")
    else:
        # tf.global_variables_initializer (FISRnet.py:588): Xavier-normal kernels, zero biases (ops.py:8-9);
        # --synthetic_weights SEED keeps the damped parity-test set instead
        W = _weights.xavier_weights(args.exp_num) if args.synthetic_weights is None else _weights.synthetic_weights(args.synthetic_weights)
        counter = 0
        print(" [!] Load failed...")
    lam = dict(recn=args.recn_lambda, tm1=args.tm1_lambda, tm2=args.tm2_lambda, tmm=args.tmm_lambda, td=args.td_lambda, ss2=args.ss2_lambda)
    net = ft.TrainNet(W, device=dev, lambdas=lam)
    if opt_state is not None:
        net.load_optimizer_state(opt_state, fallback_step=counter)   # Adam moments + the step of the bias correction (saver.restore, FISRnet.py:1108)
    else:
        # weights only (an inference checkpoint): the moments start at zero, so the bias correction must start at t = 0
        # too -- with t = counter the first updates would be ~3x the nominal step
        net.step_count = 0
        if path:
            print(" [!] the checkpoint holds no Adam slots: the optimizer state starts from zero")
    net.grad_scale = 1.0 / world
    if world > 1:
        np.random.seed(1234 + args.exp_num)          # every rank must draw the same permutations
    start_epoch = counter // max(train_iter, 1)
    start_time = time.time()
    keymap = (("data15", "data"), ("label21", "label"), ("flow16", "flow"), ("warp24", "warp"), ("flow_ss2", "flow_ss2"), ("warp_ss2", "warp_ss2"))
    last = float("nan")
    written_prev = []        # files of the checkpoint THIS run wrote last (tf.train.Saver._last_checkpoints: restore() never registers the loaded one)
    for epoch in range(start_epoch, args.epoch):
        rows = []
        rand_idx = np.random.permutation(n - nv)
        lr = args.init_lr
        for idx in range(train_iter):
            sel = rand_idx[args.batch_size * idx:args.batch_size * (idx + 1)][rank::world]
            batch = ft.to_device_batch({a: tr[b][sel] for a, b in keymap}, dev)
            lr = learning_rate(args, epoch, counter, train_iter)
            net.zero_grad()
            net.keep_preds = True
            total, t = net.loss_and_grads(batch)
            net.allreduce_grads()
            net.adam_step(lr)
            if world > 1:                             # the logged numbers are the mean over the ranks' shards
                v = torch.tensor([total] + [t[k] for k in sorted(t)], dtype=torch.float64)
                dist.all_reduce(v)
                v /= world
                total, t = float(v[0]), dict(zip(sorted(t), v[1:].tolist()))
            psnr = _psnr(_ovlp([p[2] for p in net.last_preds[:3]]), batch["label21"])
            if world > 1:                             # ... and so is the PSNR (same samples as the losses)
                pv = torch.tensor([psnr], dtype=torch.float64)
                dist.all_reduce(pv)
                psnr = float(pv[0]) / world
            s1 = lam["recn"] * t["recn"] + lam["tm1"] * t["tm"] + lam["tmm"] * t["tmm"] + lam["td"] * t["td"]
            s2 = lam["recn"] * t["recn_ss2"] + lam["td"] * t["td_ss2"] + lam["tm2"] * t["tm_ss2"]
            row = (psnr, t["recn"], t["tm"], t["tmm"], t["td"], s1, t["recn_ss2"], t["td_ss2"], t["tm_ss2"], s2, total)
            if idx % args.freq_display == 0:
                print("Epoch: [%3d], [%4d/%4d]-th batch, time: %4.2f(min.), "
                      "train_PSNR: %.3f, recnLoss: %.6f, tmLoss: %.6f, tmmLoss: %.6f, tdLoss: %.6f, "
                      "totalLoss_s1: %.6f,recnLoss_ss2: %.6f,"
                      "tdLoss_ss2: %.6f, tmLoss_ss2: %.6f, totalLoss_ss2: %.6f, total_loss: %.6f"
                      % ((epoch, idx, train_iter, (time.time() - start_time) / 60) + row))
            counter += 1
            rows.append(row)
        if rows:
            m = np.mean(np.array(rows), axis=0)
            last = float(m[-1])
            print("# (average) Epoch: [%4d], LR: %1.10f, time: %4.2f(minutes), "
                  "train_PSNR: %.3f, recnLoss: %.6f, tmLoss: %.6f, tmmLoss: %.6f, tdLoss: %.6f, "
                  "totalLoss_s1: %.6f,recnLoss_ss2: %.6f,"
                  "tdLoss_ss2: %.6f, tmLoss_ss2: %.6f, totalLoss_ss2: %.6f, total_loss: %.6f"
                  % ((epoch, lr, (time.time() - start_time) / 60) + tuple(m)))
        # validation (FISRnet.py:713-742): stride-1 windows only, reconstruction loss and PSNR of the overlapped sequence
        vl, vp = [], []
        for vi in range(val_iter):
            sl = slice(args.val_batch_size * vi, args.val_batch_size * (vi + 1))
            vb = ft.to_device_batch({a: va[b][sl] for a, b in keymap}, dev)
            net.tape = []
            p3 = [net.model(net.window_input(vb, k))[2] for k in range(3)]
            net.tape = []
            ov = _ovlp(p3)
            b_, _, h_, w_, _ = ov.shape
            gt = vb["label21"].reshape(b_, h_, w_, 7, 3).permute(0, 3, 1, 2, 4)
            vl.append(float(((ov - gt) ** 2).mean()))
            vp.append(_psnr(ov, vb["label21"]))
        if vl:
            print("######### Validation (average),Epoch: [%4d/%4d]-th epoch, time: %4.2f(min.), val_PSNR: %.3f[dB], "
                  "recnLoss: %.6f #########" % (epoch, args.epoch, (time.time() - start_time) / 60, np.mean(vp), np.mean(vl)))
        # save_checkpoint (FISRnet.py:1091-1099): <checkpoint_dir>/<model_dir>/FISRnet-<global_step>
        name = f"FISRnet-{counter}"
        if rank == 0:
            # weights + Adam slots + beta powers, as the reference's Saver (created after build_model, FISRnet.py:585) stores them
            Wn = net.weights_numpy()
            Wn.update(net.optimizer_state_numpy())
            np.savez(os.path.join(ckpt_dir, name + ".npz"), **Wn)
            if getattr(args, "save_tf_bundle", False):
                # tf.train.Saver's checkpoint-V2 files (<name>.index / .data-00000-of-00001) under the reference's variable
                # names, so that the reference's FISRnet.load (FISRnet.py:1101-1115) restores what was trained here
                from . import tf_bundle
                # ... incl. the global step the reference's training-mode Saver also stores and restores: the unnamed
                # tf.Variable(0, trainable=False) of FISRnet.py:232 is called `Variable` (int32 scalar) and drives
                # piecewise_constant (without it saver.restore of this bundle fails with NotFound in train mode)
                Wb = dict(Wn)
                Wb["Variable"] = np.array(counter, np.int32)
                tf_bundle.write_bundle(os.path.join(ckpt_dir, name), Wb)
            with open(os.path.join(ckpt_dir, "checkpoint"), "w") as f:
                f.write(f'model_checkpoint_path: "{name}"\nall_model_checkpoint_paths: "{name}"\n')
            # Saver(max_to_keep=1) (FISRnet.py:585): the previous checkpoint SAVED BY THIS RUN goes once the state file names the new
            # one.  A checkpoint the run was restored from (a pretrained / inference-only bundle the user put here) was never
            # registered with the Saver and stays, exactly as under the reference.
            written_now = [fn for fn in os.listdir(ckpt_dir) if fn.split(".")[0] == name]
            for fn in written_prev:
                if fn not in written_now:
                    try:
                        os.remove(os.path.join(ckpt_dir, fn))
                    except OSError:
                        pass
            written_prev = written_now
    return last
