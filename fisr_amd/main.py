"""Command line of the MI355X-native FISR inference path -- same flags as the reference's
`main.py` (main.py:23-106) for the phases this build implements.

    python -m fisr_amd.main --phase test [--test_data_path ... --checkpoint_dir ...]
    python -m fisr_amd.main --phase FISR_for_video --frame_folder_path DIR --flow_file X.flo

Differences from the reference, all deliberate:
  * `--test_patch`, `--test_input_size`, `--FISR_input_size`, `--FISR_test_patch` parse "2,2" /
    "(2,2)" into integer tuples (the reference's `type=tuple` turns a CLI string into a tuple of
    characters, main.py:89-103); `--scale_factor` is an int (main.py:29 makes 2.0).
  * `--phase train` runs the reference's training graph on the GPU (fisr_amd/train.py, train_harness.py: the four
    weight-sharing passes, seven loss terms, Adam, the lr schedules, validation, checkpoints; no TensorBoard).  The
    reference's pre-made training files are not in its tree: `--synthetic_train N` substitutes N seeded samples.
  * `--phase FISR_for_video` computes the optical flow on the GPU like the reference (main.py:210): PWC-Net-large
    (fisr_amd/pwcnet.py), weights from `--pwc_ckpt` (TF bundle prefix or .npz; the reference hard-codes
    './models/pwcnet-lg-6-2-multisteps-chairsthingsmix/pwcnet.ckpt-595000') or seeded stand-ins with
    `--synthetic_weights`; the flow is written as the reference's 5-D `.flo` next to the frames.  `--flow_file` skips
    the estimator and uses a pre-computed file.  The frame warp (main.py:213) runs on the GPU too.
  * `--phase test` makes its own `.flo` / `_warp.mat` when neither exists (`--prepare auto|always|never|only`, `--prepare_ss`):
    the reference's two pre-processing scripts (FISR_pwcnet_predict_from_img_test.py:84-147, FISR_warp_mat_with_flo.py:95-129)
    on the GPU, for the scene folder `--test_data_path`.
  * extra flags: `--precision {fp32,bf16x3,f16f8,fp16}` (default fp32, the reference's arithmetic),
    `--device`, `--synthetic_weights SEED`, `--no_batch_tiles` (one tile per forward, the reference's
    schedule: smallest workspace).
"""
from __future__ import annotations

import argparse
import os
import sys


from .fisrnet import DEFAULT_PRECISION, PRECISIONS


def _tuple2(s):
    if isinstance(s, (tuple, list)):
        return tuple(int(v) for v in s)
    vals = [v for v in str(s).replace("(", " ").replace(")", " ").replace(",", " ").split() if v]
    if len(vals) != 2:
        raise argparse.ArgumentTypeError(f"expected two integers like 2,2 -- got {s!r}")
    return int(vals[0]), int(vals[1])


def check_folder(d):
    os.makedirs(d, exist_ok=True)
    return d


def parse_args(argv=None):
    desc = "FISR (MI355X-native inference path): joint frame interpolation and super-resolution"
    p = argparse.ArgumentParser(description=desc)
    p.add_argument("--net_type", type=str, default="FISRnet", choices=["FISRnet"])
    p.add_argument("--fraction_gpu", type=float, default=1.0, help="accepted for compatibility; unused")
    p.add_argument("--phase", type=str, default="FISR_for_video", choices=["train", "test", "FISR_for_video"])
    p.add_argument("--scale_factor", type=int, default=2)
    p.add_argument("--train_data_path", type=str, default="./data/train/LR_LFR/LR_Surfing_SlamDunk_5seq.mat")
    p.add_argument("--train_flow_data_path", type=str, default="./data/train/flow/LR_Surfing_SlamDunk_5seq_ss1.flo")
    p.add_argument("--train_flow_ss2_data_path", type=str, default="./data/train/flow/LR_Surfing_SlamDunk_5seq_ss2.flo")
    p.add_argument("--train_warped_data_path", type=str, default="./data/train/warped/LR_Surfing_SlamDunk_5seq_ss1_warp.mat")
    p.add_argument("--train_wapred_ss2_data_path", type=str, default="./data/train/warped/LR_Surfing_SlamDunk_5seq_ss2_warp.mat")
    p.add_argument("--train_label_path", type=str, default="./data/train/HR_HFR/HR_Surfing_SlamDunk_5seq.mat")
    p.add_argument("--test_data_path", type=str, default="./data/test/LR_LFR")
    p.add_argument("--test_flow_data_path", type=str, default="./data/test/flow/LR_Surfing_SlamDunk_test_ss1.flo")
    p.add_argument("--test_warped_data_path", type=str, default="./data/test/warped/LR_Surfing_SlamDunk_test_ss1_warp.mat")
    p.add_argument("--test_label_path", type=str, default="./data/test/HR_HFR")
    p.add_argument("--test_img_dir", type=str, default="./test_img_dir")
    p.add_argument("--text_dir", type=str, default="./text_dir")
    p.add_argument("--checkpoint_dir", type=str, default="./checkpoint_dir")
    p.add_argument("--log_dir", type=str, default="./logdir")
    p.add_argument("--exp_num", type=int, default=1)
    # training hyper-parameters (main.py:64-85)
    p.add_argument("--epoch", type=int, default=100)
    p.add_argument("--freq_display", type=int, default=100)
    p.add_argument("--init_lr", type=float, default=0.0001)
    p.add_argument("--lr_type", type=str, default="stair_decay", choices=["linear_decay", "stair_decay", "no_decay"])
    p.add_argument("--lr_stair_decay_points", type=int, nargs="+", default=[80, 90])
    p.add_argument("--lr_decreasing_factor", type=float, default=0.1)
    p.add_argument("--lr_linear_decay_point", type=int, default=50)
    p.add_argument("--batch_size", type=int, default=8)
    p.add_argument("--n_train_img_showed", type=int, default=3, help="accepted for compatibility (TensorBoard images); unused")
    p.add_argument("--val_batch_size", type=int, default=2)
    p.add_argument("--val_data_size", type=int, default=320)
    p.add_argument("--recn_lambda", type=float, default=1.0)
    p.add_argument("--tm1_lambda", type=float, default=1.0)
    p.add_argument("--tm2_lambda", type=float, default=0.1)
    p.add_argument("--tmm_lambda", type=float, default=1.0)
    p.add_argument("--td_lambda", type=float, default=0.1)
    p.add_argument("--ss2_lambda", type=float, default=1.0)
    p.add_argument("--save_tf_bundle", action="store_true",
                   help="--phase train: also write every checkpoint as a TensorFlow checkpoint-V2 bundle (FISRnet-<step>.index/.data-*)")
    p.add_argument("--synthetic_train", type=int, default=0, metavar="N",
                   help="train on N seeded synthetic samples (the reference's pre-made training .mat/.flo files are not in its tree)")
    p.add_argument("--test_patch", type=_tuple2, default=(2, 2))
    p.add_argument("--test_input_size", type=_tuple2, default=(1080, 1920))
    p.add_argument("--frame_folder_path", type=str, default="./FISR_test_folder/scene1")
    p.add_argument("--FISR_input_size", type=_tuple2, default=(1080, 1920))
    p.add_argument("--frame_num", type=int, default=5)
    p.add_argument("--FISR_test_patch", type=_tuple2, default=(2, 2))
    # build-specific
    p.add_argument("--precision", type=str, default=DEFAULT_PRECISION, choices=sorted(PRECISIONS))
    p.add_argument("--check_published", action="store_true",
                   help="--phase test: compare the four averages with the figures the reference publishes for its pre-trained "
                        "weights on its 4K test set (README.md:97: PSNR 37.86 / 48.07 dB, SSIM 0.9743 / 0.9921) within the "
                        "tolerance of BASELINE.json (+-0.02 dB, 1e-3 SSIM); exit status 4 if any of them is outside")
    p.add_argument("--prepare", type=str, default="auto", choices=["auto", "always", "never", "only"],
                   help="--phase test: make --test_flow_data_path / --test_warped_data_path from the scene folder --test_data_path on the "
                        "GPU (the reference's FISR_pwcnet_predict_from_img_test.py + FISR_warp_mat_with_flo.py): 'auto' = when neither "
                        "file exists, 'always', 'never', 'only' = make the files (also the ss2 pair with --prepare_ss 2) and stop; "
                        "--phase train --prepare only: the TRAINING set's flow / warp files from --train_data_path's LR_data (r06)")
    p.add_argument("--prepare_ss", type=int, default=1, choices=[1, 2], help="temporal stride of --prepare only (ss1 / ss2 files)")
    p.add_argument("--pad_mode", action="store_true",
                   help="not in the reference (FISRnet.py:820-825 crops the frame to a multiple of 32 x patches: 1080 -> 1024 rows, a "
                        "2048 x 3840 output): replicate-pad the inputs instead (1080 -> 1088), predict, and crop the output to 2 H x 2 W "
                        "(2160 x 3840).  Default off: the reference's crop, bit for bit")
    p.add_argument("--no_batch_tiles", dest="batch_tiles", action="store_false",
                   help="run the tiles of a window one forward at a time (reference schedule, smallest workspace)")
    p.add_argument("--device", type=str, default=None, help="default cuda:<LOCAL_RANK>")
    p.add_argument("--parallelism", type=str, default="none", choices=["none", "frame", "tile"],
                   help="FISR_for_video under `python -m torch.distributed.run --nproc-per-node N -m fisr_amd.main ...`: "
                        "'frame' = windows round-robin over the GPUs; 'tile' = the test_patch tiles of a window on "
                        "different GPUs with an RCCL all-gather of the 32-px halos and of the uint8 output tiles "
                        "(N must be a multiple of the tile count: 8 GPUs = 2 windows x 2x2 tiles)")
    p.add_argument("--flow_file", type=str, default=None, help="pre-computed 5-D .flo for FISR_for_video (default: PWC-Net on the GPU)")
    p.add_argument("--pwc_ckpt", type=str, default="./models/pwcnet-lg-6-2-multisteps-chairsthingsmix/pwcnet.ckpt-595000",
                   help="PWC-Net weights: TF checkpoint-V2 bundle prefix or .npz keyed by the TF variable names")
    p.add_argument("--flow_precision", type=str, default="fp32", choices=["fp32", "fp16"],
                   help="arithmetic of the on-GPU PWC-Net: fp32, or fp16 feature tensors with fp32 accumulation and fp32 flows (cfg5)")
    p.add_argument("--warp_file", type=str, default=None, help="pre-computed warp (.mat/.npy); default: warp on the GPU")
    p.add_argument("--synthetic_weights", type=int, default=None, metavar="SEED",
                   help="use seeded stand-in weights instead of a checkpoint (no checkpoint ships with the reference)")
    args = p.parse_args(argv)
    return check_args(args)


def check_args(args):
    """main.py:108-121."""
    for d in (args.checkpoint_dir, args.text_dir, args.log_dir, args.test_img_dir):
        check_folder(d)
    return args


# what the reference publishes for checkpoint_dir/FISRnet_exp1 on data/test (README.md:97), and BASELINE.json's tolerance
PUBLISHED = (("FISR_PSNR", 37.86, 0.02, "[dB]"), ("SR_PSNR", 48.07, 0.02, "[dB]"), ("FISR_SSIM", 0.9743, 1e-3, ""), ("SR_SSIM", 0.9921, 1e-3, ""))


def check_published(res) -> bool:
    """The day `checkpoint_dir/FISRnet_exp1` and `data/test` are on the box, `python -m fisr_amd.main --phase test --check_published`
    is the one command that pins the whole path to the reference: prints each average of `FISRnet.test` (FISRnet.py:922-933) next to
    README.md:97's figure and says whether it is inside the tolerance.  (The published figures are rounded to 2 / 4 digits: half
    a unit of the last digit is added to the tolerance.)"""
    ok = True
    for key, want, tol, unit in PUBLISHED:
        got = float(res[key])
        slack = tol + (0.005 if unit else 0.00005)
        inside = abs(got - want) <= slack
        ok &= inside
        print("######### published check: %-9s %.4f%s vs README.md:97 %.4f%s  (difference %+.4f, tolerance %.4f): %s #########"
              % (key, got, unit, want, unit, got - want, slack, "ok" if inside else "OUTSIDE"))
    print("######### published check: %s #########" % ("all four averages inside the tolerance" if ok else "FAILED"))
    return ok


def main(argv=None):
    args = parse_args(argv)
    if args.phase == "train":
        from . import train_harness
        if args.device is None:
            args.device = f"cuda:{int(os.environ.get('LOCAL_RANK', '0'))}"
        if args.prepare == "only":
            # the training set's pre-processing (FISR_pwcnet_predict_from_mat.py + FISR_warp_mat_with_flo.py): --train_data_path's
            # LR_data -> the ss1 (or, --prepare_ss 2, the ss2) .flo / _warp.mat pair --phase train reads, on the GPU, and stop
            from . import harness
            from .fisrnet import FISRnet
            net = FISRnet(args)
            try:
                _, _, fp, wp = harness.prepare_patch_set(net, args, ss=args.prepare_ss)
            finally:
                net.close()
            print(" [*] Flow file saved: %s" % fp)
            print(" [*] Warp file saved: %s" % wp)
            return 0
        if int(os.environ.get("WORLD_SIZE", "1")) > 1:
            import torch
            import torch.distributed as dist
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            torch.cuda.set_device(torch.device(args.device))
            backend = os.environ.get("FISR_DIST_BACKEND", "nccl")
            if backend == "nccl":
                dist.init_process_group("nccl", device_id=torch.device(args.device))
            else:
                dist.init_process_group(backend)
        train_harness.run_train(args)
        print(" [*] Training finished!")
        return 0
    from . import io as fio
    from . import harness, weights
    from .fisrnet import FISRnet

    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.device is None:
        args.device = f"cuda:{local_rank}"
    if world > 1:
        # one process per GPU; backend "nccl" is RCCL over xGMI on ROCm (FISR_DIST_BACKEND=gloo for CPU-staged tests)
        import torch
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(torch.device(args.device))
        backend = os.environ.get("FISR_DIST_BACKEND", "nccl")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device(args.device))
        else:
            dist.init_process_group(backend)
        if args.parallelism == "none":
            args.parallelism = "frame"
    net = FISRnet(args)
    if args.synthetic_weights is not None:
        net.set_weights(weights.synthetic_weights(args.synthetic_weights))
    if args.phase == "test":
        if args.prepare == "only":
            _, _, fp, wp = harness.prepare_scene_set(net, args, ss=args.prepare_ss)
            print(" [*] Flow file saved: %s" % fp)
            print(" [*] Warp file saved: %s" % wp)
            return 0
        res = net.test()
        print(" [*] Test finished!")
        if getattr(args, "check_published", False):
            return 0 if check_published(res) else 4
        return 0
    # FISR_for_video (main.py:207-235)
    flow_file = args.flow_file
    if not flow_file:
        # FISR_for_video_Compute_Flow (main.py:210): PWC-Net-large on the GPU, both directions of every frame pair.  With several
        # ranks the pairs are sharded, gathered in memory on every rank, and rank 0 writes the reference's .flo file.
        if world > 1:
            import torch.distributed as dist
            _, flow_file = harness.compute_flow(net, args, dist.get_rank(), world, return_array=True)      # (the array, not the name)
        else:
            flow_file = harness.compute_flow(net, args)
        print("[*] Flow file saved!")
    warp_file = args.warp_file
    if warp_file is None:
        flow = fio.read_flo_file_5dim(flow_file) if isinstance(flow_file, str) else flow_file
        frames = harness.sorted_pngs(args.frame_folder_path)
        warp_file = harness.warp_img(net, frames, flow)          # ndarray, stays in memory
        print("[*] Warp done on the GPU")
    net.FISR_for_video(flow_file, warp_file)
    print(" [*] FISR finished!")
    return 0


if __name__ == "__main__":
    sys.exit(main())
