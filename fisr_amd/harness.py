"""The reference's evaluation / video drivers on top of the HIP hot path.

  run_test            <-> FISRnet.test            (FISRnet.py:746-935)
  run_fisr_for_video  <-> FISRnet.FISR_for_video  (FISRnet.py:937-1084)
  warp_img            <-> FISR_for_video_Warp_Img (FISR_tfoptflow/FISR_for_video_warp_img_with_flo.py:97-151)
  prepare_scene_set   <-> the two pre-processing scripts that make `--phase test`'s inputs for a folder of 5-frame scenes:
                          FISR_tfoptflow/FISR_pwcnet_predict_from_img_test.py:84-147 (flow, [N_scenes, 8/ss, H, W, 2] .flo) and
                          FISR_tfoptflow/FISR_warp_mat_with_flo.py:95-129 (warped frames, [N_scenes, 8/ss, H, W, 3] HDF5 .mat)
  prepare_patch_set   <-> the same two steps for the TRAINING set, from the `.mat` of LR patch sequences (r06):
                          FISR_tfoptflow/FISR_pwcnet_predict_from_mat.py:80-133 and FISR_warp_mat_with_flo.py:95-129

Everything between "frames are on the GPU" and "uint8 predictions come back" runs in
libfisr_hip.so kernels (pack -> tiled forward -> stitch -> clip/quantise/colour -> SSE).  PNG
decoding/encoding and the SSIM statistic stay on the host like in the reference.

Deliberate deviations from the reference: file lists are sorted
(`glob.glob` is unsorted at FISRnet.py:953), and the final timing print of FISR_for_video uses
FISR_test_patch (the reference uses test_patch, FISRnet.py:1084).
"""
from __future__ import annotations

import glob
import math
import os
import re
import time

import numpy as np

from . import io as fio
from . import tiling


def _natural_key(p):
    return [int(t) if t.isdigit() else t for t in re.split(r"(\d+)", os.path.basename(p))]


def sorted_pngs(folder):
    return sorted(glob.glob(os.path.join(folder, "*.png")), key=_natural_key)


def ssim_pil(a_u8: np.ndarray, b_u8: np.ndarray, tile: int = 7) -> float:
    """Host restatement of SSIM_PIL.compare_ssim (requirements.txt:13; call site FISRnet.py:890-891):
    non-overlapping 7x7 tiles per channel, C1=(0.01*255)^2, C2=(0.03*255)^2, unbiased variance,
    mean over tiles and channels.  SSIM_PIL is not in the reference tree: parity unpinned."""
    a = np.asarray(a_u8, np.float64)
    b = np.asarray(b_u8, np.float64)
    h, w, c = a.shape
    th, tw = h // tile, w // tile
    a = a[:th * tile, :tw * tile].reshape(th, tile, tw, tile, c)
    b = b[:th * tile, :tw * tile].reshape(th, tile, tw, tile, c)
    n = tile * tile
    ma = a.mean(axis=(1, 3))
    mb = b.mean(axis=(1, 3))
    da = a - ma[:, None, :, None]
    db = b - mb[:, None, :, None]
    va = (da * da).sum(axis=(1, 3)) / (n - 1)
    vb = (db * db).sum(axis=(1, 3)) / (n - 1)
    cov = (da * db).sum(axis=(1, 3)) / (n - 1)
    c1, c2 = (255 * 0.01) ** 2, (255 * 0.03) ** 2
    s = ((2 * ma * mb + c1) * (2 * cov + c2)) / ((ma * ma + mb * mb + c1) * (va + vb + c2))
    return float(s.mean())


def _psnr_from_sse(sse: float, count: int, peak: float = 1.0) -> float:
    """utils.py:23-26."""
    return 10 * math.log10(peak * peak / (sse / count))


def warp_img(net, frame_paths, flow, out_path=None):
    """frames (sorted PNG paths, YUV) + flow [num_fr-1, 2, h, w, 2] -> pred [num_fr-1, 2, h, w, 3]
    float32 0..255, pair fr: [0] = warp(frame fr+1, 0.5*flow[fr,0]) ("1 -> 2"), [1] = warp(frame fr,
    0.5*flow[fr,1]) ("2 -> 1") -- warp script :112-129."""
    import torch
    num_pairs = flow.shape[0]
    h, w = flow.shape[2:4]
    pred = np.zeros((num_pairs, 2, h, w, 3), np.float32)
    frames = [torch.from_numpy(fio.read_png(p)[:h, :w]).to(net.device) for p in frame_paths[:num_pairs + 1]]
    for fr in range(num_pairs):
        f01 = torch.from_numpy(np.ascontiguousarray(flow[fr, 0])).to(net.device)
        f10 = torch.from_numpy(np.ascontiguousarray(flow[fr, 1])).to(net.device)
        pred[fr, 0] = net.warp(frames[fr + 1], f01).cpu().numpy()
        pred[fr, 1] = net.warp(frames[fr], f10).cpu().numpy()
    if out_path:
        fio.write_warp_file(out_path, pred)
    return pred


def flow_file_name(args):
    """`<folder>/<folder name>_test_ss1_fr<frame_num>.flo` (flow script :143-144)."""
    folder = args.frame_folder_path.rstrip("/")
    return os.path.join(folder, os.path.basename(folder) + "_test_ss{}_fr{}.flo".format(1, args.frame_num))


def compute_flow(net, args, rank: int = 0, world: int = 1, return_array: bool = False):
    """`FISR_for_video_Compute_Flow(args)` (FISR_tfoptflow/FISR_for_video_pwcnet_predict_from_img_test.py:84-147) on the
    GPU: the first frame_num YUV frames of the folder, PWC-Net-large in both directions per consecutive pair, written
    as the reference's 5-D .flo `<folder>/<folder name>_test_ss1_fr<frame_num>.flo` [frame_num-1, 2, h, w, 2].
    Returns the file name (and the array with return_array).

    world > 1 (one process per GPU, torch.distributed initialised): the frame PAIRS are sharded over the ranks (pair p on rank
    p % world), every rank gathers all flows in memory and rank 0 writes the file -- no rank idles in a barrier for the length of
    the clip (the process group's watchdog would abort a long one), and nobody re-reads the file, so the ranks need no shared
    file system."""
    import torch
    paths = sorted_pngs(args.frame_folder_path)
    h, w = args.FISR_input_size
    num_fr = args.frame_num
    if len(paths) < num_fr:
        raise FileNotFoundError(f"{args.frame_folder_path}: {len(paths)} frames, need {num_fr}")
    pwc = open_pwc(net, args)
    try:
        if world <= 1:
            frames = [torch.from_numpy(np.ascontiguousarray(fio.read_png(p)[:h, :w])) for p in paths[:num_fr]]
            pred = pwc.compute_flow(frames).cpu().numpy()
        else:
            pred = _gather_pair_flows(pwc, paths, h, w, num_fr, rank, world, net.device)
    finally:
        pwc.close()
    print(pred.shape)
    name = flow_file_name(args)
    if rank == 0:
        tmp = name + ".tmp%d" % os.getpid()
        fio.write_flow(pred, tmp)
        os.replace(tmp, name)                  # atomic: a concurrent reader sees the old file or the whole new one
    return (name, pred) if return_array else name


def open_pwc(net, args):
    """The flow estimator of the pre-processing scripts (ModelPWCNet(mode='test'), script :84-102) on net's device: weights from
    `--pwc_ckpt` (TF bundle prefix or .npz) or seeded stand-ins with `--synthetic_weights`; the caller closes it."""
    from . import pwcnet
    pwc = pwcnet.PWCNet(str(net.device), precision=getattr(args, "flow_precision", "fp32") or "fp32")
    try:
        if getattr(args, "synthetic_weights", None) is not None:
            pwc.set_weights(pwcnet.synthetic_weights(595000 + int(args.synthetic_weights)))
        else:
            ck = getattr(args, "pwc_ckpt", None)
            if not ck or not (os.path.isfile(ck) or os.path.isfile(ck + ".index")):
                raise FileNotFoundError(f"PWC-Net weights not found at {ck!r} (the reference downloads them separately, "
                                        "script :31); pass --pwc_ckpt, --flow_file or --synthetic_weights")
            pwc.load(ck)
    except BaseException:
        pwc.close()
        raise
    return pwc


def _gather_pair_flows(pwc, paths, h, w, num_fr, rank, world, device):
    """Multi-rank flow of a clip: pair p is computed on rank p % world; every rank ends up with [num_fr-1, 2, h, w, 2].
    The exchange is one float32 tensor per pair, broadcast from its owner into a preallocated array (no pickling, no
    world x max_size staging buffer -- ADVICE r04), behind an all-reduced ok flag: a rank whose share failed (unreadable PNG,
    out of memory) makes EVERY rank raise instead of leaving the others blocked in the collective."""
    import torch
    import torch.distributed as dist
    on_gpu = dist.get_backend() == "nccl"
    comm_dev = device if on_gpu else torch.device("cpu")
    mine, err = {}, None
    try:
        for pair in range(rank, num_fr - 1, world):
            fr = [torch.from_numpy(np.ascontiguousarray(fio.read_png(p)[:h, :w])) for p in paths[pair:pair + 2]]
            mine[pair] = pwc.compute_flow(fr)[0]                              # [2, h, w, 2] on the device
    except Exception as e:                                                    # noqa: BLE001 -- reported to every rank below
        err = e
    ok = torch.tensor([0 if err is not None else 1], dtype=torch.int32, device=comm_dev)
    dist.all_reduce(ok, op=dist.ReduceOp.MIN)
    if int(ok.item()) == 0:
        raise RuntimeError(f"compute_flow: rank {rank} " + (f"failed: {err!r}" if err is not None else "stops because another rank failed"))
    pred = np.empty((num_fr - 1, 2, h, w, 2), np.float32)
    buf = torch.empty((2, h, w, 2), dtype=torch.float32, device=comm_dev)
    for pair in range(num_fr - 1):
        owner = pair % world
        if owner == rank:
            buf.copy_(mine.pop(pair))
        dist.broadcast(buf, src=owner)
        pred[pair] = buf.cpu().numpy()
    return pred


def scene_set_pairs(n_seq: int, ss: int):
    """The frame pairs one scene contributes at temporal stride ss (flow script :121-128, warp script :112-114):
    seq = 0 .. n_seq - 2 ss, frames (ss seq, ss (seq + 1)); entry 2 seq is "1 -> 2", entry 2 seq + 1 is "2 -> 1"."""
    if ss not in (1, 2):
        raise ValueError("temporal stride ss must be 1 or 2 (main.py:38-44: the ss1 / ss2 files)")
    return [(ss * seq, ss * (seq + 1)) for seq in range(n_seq - (ss * 2 - 1))]


def prepare_scene_set(net, args, ss: int = 1, data_path=None, flow_path=None, warp_path=None, n_seq: int = 5, pwc=None,
                      write: bool = True):
    """The reference's two pre-processing scripts for a folder of `n_seq`-frame scenes (natural-sorted YUV PNGs, README.md:38-55),
    both on the GPU:

      flow  FISR_tfoptflow/FISR_pwcnet_predict_from_img_test.py:84-147: PWC-Net-large on the x2 up-resized RGB pair in both
            directions, resized back with anti-aliasing and halved -> pred[num, 2 seq] (frame ss seq -> ss (seq + 1)) and
            pred[num, 2 seq + 1] (the reverse), float32 [N_scenes, 8 / ss, h, w, 2], written by `write_flow` (script :57-81);
      warp  FISR_tfoptflow/FISR_warp_mat_with_flo.py:95-129: pred[num, 2 seq] = RGB2YUV(remap(YUV2RGB(frame ss (seq + 1)),
            0.5 flow[num, 2 seq])), pred[num, 2 seq + 1] = the same of frame ss seq with flow[num, 2 seq + 1], float32 0..255
            [N_scenes, 8 / ss, h, w, 3], written as the MATLAB-compatible HDF5 `.mat` hdf5storage makes (script :131-136; `.npy` too).

    The scripts' paths are edited by hand ("# check" marks); here they default to `--test_data_path`, `--test_flow_data_path` and
    `--test_warped_data_path` (with `ss1` -> `ss2` in the two file names for ss = 2: main.py:36-44's naming).  Unlike the scripts'
    unsorted `glob.glob` the file list is natural-sorted.  Returns (flow, warp, flow_path, warp_path); the arrays
    are what was written, bit for bit."""
    import torch
    data_path = data_path or net.test_data_path
    flow_path = flow_path or (net.test_flow_data_path if ss == 1 else net.test_flow_data_path.replace("ss1", "ss2"))
    warp_path = warp_path or (net.test_warped_data_path if ss == 1 else net.test_warped_data_path.replace("ss1", "ss2"))
    if ss == 2 and write and (os.path.abspath(flow_path) == os.path.abspath(net.test_flow_data_path) or
                              os.path.abspath(warp_path) == os.path.abspath(net.test_warped_data_path)):
        # the derived name only differs from the ss1 file when that one carries "ss1": never overwrite the stride-1 files with
        # [N, 4, ...] stride-2 data that `--phase test` would then index as stride 1
        raise ValueError(f"--prepare_ss 2: the stride-2 file names are derived by 'ss1' -> 'ss2', but {net.test_flow_data_path!r} / "
                         f"{net.test_warped_data_path!r} do not contain 'ss1'; pass flow_path= / warp_path= (or rename the ss1 files)")
    paths = sorted_pngs(data_path)
    n_scenes = len(paths) // n_seq
    if n_scenes == 0:
        raise FileNotFoundError(f"{data_path}: {len(paths)} PNG frames, need at least one scene of {n_seq}")
    h, w = net.test_input_size
    pairs = scene_set_pairs(n_seq, ss)
    flow = np.zeros((n_scenes, 2 * len(pairs), h, w, 2), np.float32)           # (script :117: 8 // ss entries)
    warp = np.zeros((n_scenes, 2 * len(pairs), h, w, 3), np.float32)
    own = pwc is None
    if own:
        pwc = open_pwc(net, args)
    try:
        for num in range(n_scenes):
            scene = paths[num * n_seq:(num + 1) * n_seq]
            frames = [torch.from_numpy(np.ascontiguousarray(fio.read_png(p)[:h, :w])).to(net.device) for p in scene[::ss]]
            if frames[0].shape[0] < h or frames[0].shape[1] < w:
                raise ValueError(f"{scene[0]}: {tuple(frames[0].shape[:2])} is smaller than --test_input_size {h}x{w}")
            flow[num], warp[num] = _flows_and_warps(net, pwc, frames[:len(pairs) + 1])
            print(num)                                                           # (script :139, warp script :125)
    finally:
        if own:
            pwc.close()
    print(flow.shape)
    if write:
        _write_flow_and_warp(flow, warp, flow_path, warp_path)
    return flow, warp, flow_path, warp_path


def _flows_and_warps(net, pwc, frames):
    """One sample of the two pre-processing scripts: `frames` = k + 1 consecutive (already temporally strided) YUV uint8 frames on the
    device -> flow [2 k, h, w, 2] (entry 2 seq = frame seq -> seq + 1, entry 2 seq + 1 the reverse) and the warped middle frames
    [2 k, h, w, 3] float32 0..255 (entry 2 seq = frame seq + 1 pulled back by half of flow 2 seq, entry 2 seq + 1 = frame seq by half of
    flow 2 seq + 1) -- FISR_pwcnet_predict_from_{img_test,mat}.py and FISR_warp_mat_with_flo.py:115-124, both as numpy arrays."""
    k = len(frames) - 1
    h, w = frames[0].shape[:2]
    f = pwc.compute_flow(frames)                                                 # [k, 2, h, w, 2]: [seq, 0] = 1 -> 2, [seq, 1] = 2 -> 1
    warp = np.zeros((2 * k, h, w, 3), np.float32)
    for seq in range(k):
        warp[2 * seq] = net.warp(frames[seq + 1], f[seq, 0]).cpu().numpy()
        warp[2 * seq + 1] = net.warp(frames[seq], f[seq, 1]).cpu().numpy()
    return f.reshape(2 * k, h, w, 2).cpu().numpy(), warp


def _write_flow_and_warp(flow, warp, flow_path, warp_path):
    """`write_flow` (script :57-81) and the MATLAB-compatible `.mat` of hdf5storage (warp script :131-136; `.npy` by extension), each
    through a per-process temporary name and an atomic rename."""
    for name, arr, writer in ((flow_path, flow, lambda a, fn: fio.write_flow(a, fn)), (warp_path, warp, lambda a, fn: fio.write_warp_file(fn, a))):
        os.makedirs(os.path.dirname(os.path.abspath(name)), exist_ok=True)
        root, ext = os.path.splitext(name)
        tmp = root + ".tmp%d" % os.getpid() + ext                            # (the extension picks the container)
        writer(arr, tmp)
        os.replace(tmp, name)


def read_mat_frames(path, key="LR_data"):
    """The 5-D frame arrays of the reference's pre-made `.mat` files (utils.py:29-43 / the scripts' read_mat_file): dataset
    [N, N_seq, C, W, H] on disk -> uint8 [N, N_seq, H, W, C] (YUV 0..255; `.npy` / `.npz` hold [N, N_seq, H, W, C] directly)."""
    ext = os.path.splitext(path)[1].lower()
    if ext == ".npy":
        a = np.load(path)
    elif ext == ".npz":
        a = np.load(path)[key]
    else:
        try:
            import h5py
        except ImportError:
            from . import hdf5_min
            a = hdf5_min.read_dataset(path, key)
        else:
            with h5py.File(path, "r") as f:
                a = np.array(f[key])
        a = np.swapaxes(a, 2, 4)
    if a.ndim != 5 or a.shape[4] != 3:
        raise ValueError(f"{path}: expected [N, N_seq, H, W, 3] frames under {key!r}, got {a.shape}")
    if a.dtype != np.uint8:
        r = np.rint(a)
        if float(np.abs(a - r).max()) > 1e-3 or r.min() < 0 or r.max() > 255:
            raise ValueError(f"{path}: {key!r} is not 8-bit YUV data (the scripts cast the frames to uint8 for PWC-Net)")
        a = r.astype(np.uint8)
    return np.ascontiguousarray(a)


def prepare_patch_set(net, args, ss: int = 1, data_path=None, flow_path=None, warp_path=None, key: str = "LR_data", pwc=None, write: bool = True):
    """The reference's TRAINING-set pre-processing (r06), both scripts on the GPU, from the pre-made `.mat` of LR patch sequences:

      flow  FISR_tfoptflow/FISR_pwcnet_predict_from_mat.py:80-133: `LR_data` [N, 5, h, w, 3] (YUV) -> per sample and frame pair
            (ss seq, ss (seq + 1)) PWC-Net-large on the x2 up-resized RGB frames (`adapt_info` (1, 2 h, 2 w, 2)) in both directions,
            resized back with anti-aliasing and halved -> float32 [N, 8 / ss, h, w, 2], written by `write_flow`;
      warp  FISR_tfoptflow/FISR_warp_mat_with_flo.py:95-129: [N, 8 / ss, h, w, 3] float32 0..255, the MATLAB-compatible `.mat`
            `FISRnet.train` reads (`--train_warped_data_path` / `--train_wapred_ss2_data_path`).

    Paths default to `--train_data_path` and, per stride, `--train_flow_data_path` / `--train_flow_ss2_data_path` and
    `--train_warped_data_path` / `--train_wapred_ss2_data_path` (main.py:30-44).  Returns (flow, warp, flow_path, warp_path)."""
    import torch
    if ss not in (1, 2):
        raise ValueError("temporal stride ss must be 1 or 2")
    data_path = data_path or args.train_data_path
    flow_path = flow_path or (args.train_flow_data_path if ss == 1 else args.train_flow_ss2_data_path)
    warp_path = warp_path or (args.train_warped_data_path if ss == 1 else args.train_wapred_ss2_data_path)
    data = read_mat_frames(data_path, key)
    n, n_seq, h, w, _ = data.shape
    pairs = scene_set_pairs(n_seq, ss)
    flow = np.zeros((n, 2 * len(pairs), h, w, 2), np.float32)                    # (script :117: 8 // ss entries for 5 frames)
    warp = np.zeros((n, 2 * len(pairs), h, w, 3), np.float32)
    own = pwc is None
    if own:
        pwc = open_pwc(net, args)
    try:
        for num in range(n):
            frames = [torch.from_numpy(data[num, k]).to(net.device) for k in range(0, n_seq, ss)][:len(pairs) + 1]
            flow[num], warp[num] = _flows_and_warps(net, pwc, frames)
            print(num)
    finally:
        if own:
            pwc.close()
    print(flow.shape)
    if write:
        _write_flow_and_warp(flow, warp, flow_path, warp_path)
    return flow, warp, flow_path, warp_path


def _pwc_available(args) -> bool:
    """open_pwc's precondition without opening anything: seeded stand-ins asked for, or a checkpoint (TF bundle prefix / .npz) present"""
    if getattr(args, "synthetic_weights", None) is not None:
        return True
    ck = getattr(args, "pwc_ckpt", None)
    return bool(ck) and (os.path.isfile(ck) or os.path.isfile(ck + ".index"))


def _pad_mode(net) -> bool:
    return bool(getattr(getattr(net, "args", None), "pad_mode", False))


def _frame_hw(net, H, W, num_patch):
    """the size the network runs on: the reference's crop (FISRnet.py:820-824) or, with --pad_mode, the padded frame"""
    return tiling.pad_hw(H, W, num_patch) if _pad_mode(net) else tiling.crop_hw(H, W, num_patch)


def _pad_window(src, H, W, hp, wp):
    """--pad_mode: frames / flows / warps [H, W, c] -> [hp, wp, c], the last row / column repeated (edge replication)"""
    import torch
    if (hp, wp) == (H, W):
        return src
    dev = src[0][0].device
    iy = torch.arange(hp, device=dev).clamp_(max=H - 1)
    ix = torch.arange(wp, device=dev).clamp_(max=W - 1)
    pad = lambda t: t[:H, :W].index_select(0, iy).index_select(1, ix).contiguous()
    return tuple([pad(t) for t in group] for group in src)


def _window_inputs(net, frame_paths, flow_seq, warp_seq, scene_i, sample_i, n_test_in_seq, H, W, num_patch):
    """Device tensors for one 3-frame window (FISRnet.py:803-843): (frames, flows, warps) as pack_input / forward_tiled_frames take
    them, and the crop size."""
    import torch
    h, w = _frame_hw(net, H, W, num_patch)
    frames = [torch.from_numpy(fio.read_png(frame_paths[scene_i * n_test_in_seq + sample_i + k])).to(net.device)
              for k in range(3)]
    # flow[scene, :, :, 4*s : 4*s+8] = flow sequence entries 2s..2s+3; warp likewise (6s..6s+12)
    flows = [torch.from_numpy(np.ascontiguousarray(flow_seq[scene_i, 2 * sample_i + k])).to(net.device) for k in range(4)]
    warps = [torch.from_numpy(np.ascontiguousarray(warp_seq[scene_i, 2 * sample_i + k])).to(net.device) for k in range(4)]
    src = (frames, flows, warps)
    if _pad_mode(net):
        src = _pad_window(src, H, W, h, w)
    return src, h, w


def run_test(net):
    """`--phase test`.  Returns a dict with the averages the reference prints."""
    import torch
    ok, _ = (True, 0) if net._finalized else net.load(net.checkpoint_dir)
    if not ok:
        raise FileNotFoundError(f"no checkpoint under {os.path.join(net.checkpoint_dir, net.model_dir)}")
    data_paths = sorted_pngs(net.test_data_path)
    label_paths = sorted_pngs(net.test_label_path)
    prepare = getattr(net.args, "prepare", "auto")
    have_flo, have_mat = os.path.isfile(net.test_flow_data_path), os.path.isfile(net.test_warped_data_path)
    if prepare == "auto" and have_flo != have_mat:
        # one of the two pre-made files without the other: they are made together (the warps FROM the flows), so say which is missing
        raise FileNotFoundError(f"only one of the pre-made files exists: {net.test_flow_data_path!r} ({'found' if have_flo else 'MISSING'}), "
                                f"{net.test_warped_data_path!r} ({'found' if have_mat else 'MISSING'}); remove the other or pass --prepare always "
                                "to make both again on the GPU")
    if prepare == "auto" and not have_flo and not _pwc_available(net.args):
        # auto mode would open PWC-Net: without its weights the missing FILES are the error to name (as the reference's reader would)
        raise FileNotFoundError(f"{net.test_flow_data_path!r} and {net.test_warped_data_path!r} not found, and no PWC-Net weights to make them "
                                "(--pwc_ckpt / --synthetic_weights); run the pre-processing first or pass --prepare always with weights")
    if prepare == "always" or (prepare == "auto" and not have_flo):
        # neither pre-made file exists (or --prepare always): make both on the GPU from the scene folder, as the reference's two
        # pre-processing scripts do, and continue with the arrays that were written.  Under torchrun rank 0 makes the files, the
        # others wait and read them (one PWC-Net pass over the scene set instead of world-size identical ones).
        rank, world = _dist_state()
        if world > 1:
            import torch.distributed as dist
            if rank == 0:
                print(" Start to make flow and warped data (test) on the GPU.")
                prepare_scene_set(net, net.args, ss=1)
            dist.barrier()
            flow = fio.read_flo_file_5dim(net.test_flow_data_path)
            warp = fio.read_warp_file(net.test_warped_data_path, "pred")
        else:
            print(" Start to make flow and warped data (test) on the GPU.")
            flow, warp, _, _ = prepare_scene_set(net, net.args, ss=1)
    else:
        print(" Start to read flow data (test).")
        flow = fio.read_flo_file_5dim(net.test_flow_data_path)        # [N_scenes, 8, H, W, 2]
        print(" Start to read warped data (test).")
        warp = fio.read_warp_file(net.test_warped_data_path, "pred")  # [N_scenes, 8, H, W, 3] 0..255
    if flow.ndim != 5 or flow.shape[1] != 8 or warp.shape[1] != 8:
        raise ValueError(f"{net.test_flow_data_path!r} / {net.test_warped_data_path!r}: expected [N_scenes, 8, H, W, 2 | 3] (temporal stride 1, "
                         f"main.py:36-44), got {tuple(flow.shape)} / {tuple(warp.shape)} -- stride-2 (ss2) files have 4 entries per scene")
    if flow.ndim != 5 or flow.shape[1] != 8 or warp.shape[1] != 8:
        raise ValueError(f"{net.test_flow_data_path!r} / {net.test_warped_data_path!r}: expected [N_scenes, 8, H, W, 2 | 3] (temporal stride 1, "
                         f"main.py:36-44), got {tuple(flow.shape)} / {tuple(warp.shape)} -- stride-2 (ss2) files have 4 entries per scene")
    num_patch = tuple(net.test_patch)
    H, W = net.test_input_size
    sf = net.scale_factor
    n_in_seq, n_test_in_seq = 3, 5
    n_gt_seq, n_test_label_seq = 3, 7
    test_img_dir = os.path.join(net.test_img_dir, net.model_dir)
    os.makedirs(test_img_dir, exist_ok=True)
    fisr_psnr, sr_psnr, fisr_ssim, sr_ssim = [], [], [], []
    fisr_psnr_y, sr_psnr_y = [], []
    net.inf_time = []
    start = time.time()
    n_scenes = len(data_paths) // n_test_in_seq
    for scene_i in range(n_scenes):
        for sample_i in range(n_test_in_seq - n_in_seq + 1):
            src, h, w = _window_inputs(net, data_paths, flow, warp, scene_i, sample_i, n_test_in_seq, H, W, num_patch)
            # input assembly (FISRnet.py:828-843), the tile loop (:847-880) and the forward in one call per group of equal tiles:
            # bit-identical to pack_input -> forward_tiled (tests/test_gpu_frames.py)
            full = net.forward_tiled_frames([src], h, w, num_patch, timed=True)
            if _pad_mode(net):          # the padded rows / columns were input only: the frame is H x W again
                h, w = min(h, H), min(w, W)
                full = full[:h * sf, :w * sf].contiguous()
            yuv_u8, rgb_u8 = net.unpack_output(full)
            have_gt = len(label_paths) >= (scene_i + 1) * n_test_label_seq
            psnr, ssim, psnr_y = [float("nan")] * 3, [float("nan")] * 3, [float("nan")] * 3
            for seq_i in range(n_gt_seq):
                name = (os.path.basename(label_paths[scene_i * n_test_label_seq + sample_i * 2 + seq_i])[3:]
                        if have_gt else f"s{scene_i}_w{sample_i}_f{seq_i}.png")
                if have_gt:
                    gt = fio.read_png(label_paths[scene_i * n_test_label_seq + sample_i * 2 + seq_i])[:h * sf, :w * sf]
                    gt_d = torch.from_numpy(np.ascontiguousarray(gt)).to(net.device)
                    sse = net.sse_vs_u8(full[:, :, 3 * seq_i:3 * seq_i + 3].contiguous(), gt_d)
                    psnr[seq_i] = _psnr_from_sse(sse, gt.size)
                    # Y-only PSNR, for information (BASELINE.json's metric says "PSNR-Y"; the reference itself
                    # scores the 3 YUV channels jointly, FISRnet.py:883-889)
                    sse_y = net.sse_vs_u8(full[:, :, 3 * seq_i:3 * seq_i + 1].contiguous(), gt_d[:, :, 0:1].contiguous())
                    psnr_y[seq_i] = _psnr_from_sse(sse_y, gt.size // 3)
                    # FISRnet.py:890-891: both sides as uint8(x*255) (GT: uint8 -> /255 -> *255 truncation)
                    # (x/255*255 truncates to x-1 for some x: done on the host exactly like the reference)
                    gt_q = (np.clip(gt.astype(np.float64) / 255., 0, 1) * 255).astype("uint8")
                    ssim[seq_i] = net.ssim_u8(yuv_u8[:, :, 3 * seq_i:3 * seq_i + 3],
                                              torch.from_numpy(np.ascontiguousarray(gt_q)))
                fio.write_png(os.path.join(test_img_dir, "pred_{}".format(name)), rgb_u8[seq_i].cpu().numpy())
            print(" <Test> [%4d/%4d]-th image, scene: %2d-%d, time: %4.4f(minutes), test_PSNR: fr1 (FI-SR) %.8f[dB], "
                  "fr2 (SR) %.8f[dB], fr3 (FI-SR) %.8f[dB]  " % (scene_i * 3 + sample_i, n_scenes * 3, scene_i, sample_i,
                                                                 (time.time() - start) / 60, psnr[0], psnr[1], psnr[2]))
            print(" ---- test_SSIM: fr1 (FI-SR) %.8f, fr2 (SR) %.8f, fr3 (FI-SR) %.8f  "
                  % (ssim[0], ssim[1], ssim[2]))                  # FISRnet.py:897-899
            fisr_psnr.append(psnr[0]); sr_psnr.append(psnr[1])
            fisr_ssim.append(ssim[0]); sr_ssim.append(ssim[1])
            fisr_psnr_y.append(psnr_y[0]); sr_psnr_y.append(psnr_y[1])
            if sample_i == 2:                                     # FISRnet.py:918-920
                fisr_psnr.append(psnr[2]); fisr_ssim.append(ssim[2]); fisr_psnr_y.append(psnr_y[2])
    res = dict(FISR_PSNR=float(np.mean(fisr_psnr)), SR_PSNR=float(np.mean(sr_psnr)),
               FISR_SSIM=float(np.mean(fisr_ssim)), SR_SSIM=float(np.mean(sr_ssim)),
               FISR_PSNR_Y=float(np.mean(fisr_psnr_y)), SR_PSNR_Y=float(np.mean(sr_psnr_y)),
               inference_time_per_frame=float(np.mean(net.inf_time)) * num_patch[0] * num_patch[1])
    print("######### Test (average) test_PSNR: FISR %.8f[dB], SR %.8f[dB]  #########" % (res["FISR_PSNR"], res["SR_PSNR"]))
    print("######### Test (average) test_SSIM: FISR %.8f, SR %.8f #########" % (res["FISR_SSIM"], res["SR_SSIM"]))
    print("######### (extra) Y-channel-only test_PSNR: FISR %.8f[dB], SR %.8f[dB]  #########" % (res["FISR_PSNR_Y"], res["SR_PSNR_Y"]))
    print("######### Estimated Inference Time (per one output 4K frame): %.8f[s]  #########" % res["inference_time_per_frame"])
    # (what is timed differs from the reference's sess.run of ONE tile, FISRnet.py:868-874: here a tile's share of one batched
    #  fisr_forward_frames call -- input assembly, tile cut and level inputs included -- times the tiles of a frame)
    res["inference_time_is"] = "tile's share of a batched fisr_forward_frames call (input assembly included) x tiles per frame"
    print("#########   (timed: %s)  #########" % res["inference_time_is"])
    return res


def _dist_state():
    """(rank, world) of the torch.distributed default group, (0, 1) when not initialised."""
    try:
        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized():
            return dist.get_rank(), dist.get_world_size()
    except ImportError:
        pass
    return 0, 1


def run_fisr_for_video(net, flow_file_name, warp_file_name, parallel=None):
    """`--phase FISR_for_video`: writes <frame_folder>/FISR_frames/pred_{k}.png (RGB) and
    pred_YUV_{k}.png, k = 2*fr + seq_i, later windows overwrite earlier (FISRnet.py:1064-1077).

    parallel (default: net.args.parallelism, else None) shards the work over the ranks of an initialised
    torch.distributed job (one process per GPU):
      "frame": window fr goes to rank fr % world (independent forwards, no data-path collective); a frame
               that two windows produce is written by the later window only, which is the file the
               reference's sequential overwrite leaves behind.
      "tile":  the ranks form tile groups of num_patch[0]*num_patch[1] ranks (fisr_amd/dist.py); window fr
               goes to group fr % n_groups; every rank packs only its core of the input, the 32-px halos
               are all-gathered, each rank runs the forward of ITS tile, the trimmed uint8 tiles are
               all-gathered and the group's first rank writes the PNGs."""
    import torch
    from . import dist as fdist
    ok, _ = (True, 0) if net._finalized else net.load(net.checkpoint_dir)
    if not ok:
        raise FileNotFoundError(f"no checkpoint under {os.path.join(net.checkpoint_dir, net.model_dir)}")
    if parallel is None:
        parallel = getattr(net.args, "parallelism", None)
    rank, world = _dist_state()
    if parallel in (None, "none") or world == 1 and parallel == "frame":
        parallel, rank, world = None, 0, 1
    if parallel not in (None, "frame", "tile"):
        raise ValueError(f"parallel must be None, 'frame' or 'tile', got {parallel!r}")
    paths = sorted_pngs(net.frame_folder_path)
    num_fr = net.frame_num
    out_dir = os.path.join(net.frame_folder_path, "FISR_frames")
    os.makedirs(out_dir, exist_ok=True)
    flow = fio.read_flo_file_5dim(flow_file_name) if isinstance(flow_file_name, str) else np.asarray(flow_file_name)
    warp = fio.read_warp_file(warp_file_name, "pred") if isinstance(warp_file_name, str) else np.asarray(warp_file_name)
    # [N, 2, ...] pairs -> windows of two consecutive pairs (FISRnet.py:965-966, 972-973)
    flow = np.concatenate((flow[0:num_fr - 2], flow[1:num_fr - 1]), axis=1)
    warp = np.concatenate((warp[0:num_fr - 2], warp[1:num_fr - 1]), axis=1)
    num_patch = tuple(net.FISR_test_patch)
    H, W = net.FISR_input_size
    sf = net.scale_factor
    digits = math.ceil(math.log10(2 * (num_fr - 1)))
    net.inf_time = []
    start = time.time()
    written = []
    n_win = num_fr - 2
    topo = grp = None
    if parallel == "tile":
        topo = fdist.TileTopology(num_patch, world, rank)
        grp = topo.make_group()
        my_windows = [fr for fr in range(n_win) if fr % topo.n_groups == topo.group_index]
        writer = topo.tile == 0
    elif parallel == "frame":
        my_windows = fdist.shard_units(n_win, world, rank)
        writer = True
    else:
        my_windows = list(range(n_win))
        writer = True
    if _pad_mode(net) and parallel == "tile":
        raise ValueError("--pad_mode is not available with tile-parallel ranks (the halo exchange works on the cropped frame)")
    h, w = _frame_hw(net, H, W, num_patch)

    def window_tensors(fr):
        src = ([torch.from_numpy(fio.read_png(paths[fr + k])).to(net.device) for k in range(3)],
               [torch.from_numpy(np.ascontiguousarray(flow[fr, k])).to(net.device) for k in range(4)],
               [torch.from_numpy(np.ascontiguousarray(warp[fr, k])).to(net.device) for k in range(4)])
        return _pad_window(src, H, W, h, w) if _pad_mode(net) else src

    # tile-parallel: the core packing + halo all-gather of the NEXT window run on a side stream under this window's forward
    prefetch = fdist.HaloPrefetcher(num_patch, net.device, group=grp) if parallel == "tile" else None
    pending = None
    next_t = None            # decoded + uploaded tensors of the window after the current one: PNG reads and host-to-device copies
                             # happen OUTSIDE the timed region (they used to sit between t0 and the closing synchronize,
                             # with the GPU idle behind the host's PNG decode)
    if prefetch is not None and my_windows:
        fr0 = my_windows[0]
        core0 = fdist.pack_core(net, *window_tensors(fr0), h, w, num_patch, topo.tile)
        pending = (core0, prefetch.start(core0))
        if len(my_windows) > 1:
            next_t = window_tensors(my_windows[1])
    for wi, fr in enumerate(my_windows):
        if parallel != "tile":
            frames, flows, warps = window_tensors(fr)
        if parallel == "tile":
            torch.cuda.synchronize(net.device)
            t0 = time.time()
            core, handle = pending
            tile_in = prefetch.finish(handle)
            if wi + 1 < len(my_windows):
                nxt = fdist.pack_core(net, *next_t, h, w, num_patch, topo.tile)
                pending = (nxt, prefetch.start(nxt))
            yuv_b, rgb_b = fdist.tile_parallel_engine_window(net, core, num_patch, group=grp, want_rgb=True, tile_in=tile_in)
            yuv_u8, rgb_u8 = yuv_b[0], rgb_b[0]
            torch.cuda.synchronize(net.device)
            # the reference's figure is (mean time of one tile forward) x tiles: here the tiles run
            # concurrently, so the whole window took `dt` = one tile's time
            net.inf_time.append(time.time() - t0)
            next_t = window_tensors(my_windows[wi + 2]) if wi + 2 < len(my_windows) else None
        else:
            full = net.forward_tiled_frames([(frames, flows, warps)], h, w, num_patch, timed=True)
            if _pad_mode(net):
                full = full[:min(h, H) * sf, :min(w, W) * sf].contiguous()
            yuv_u8, rgb_u8 = net.unpack_output(full)
        if writer:
            yuv_host = yuv_u8.cpu().numpy()
            for seq_i in range(3):
                if parallel is not None and seq_i == 2 and fr != n_win - 1:
                    continue        # frame 2fr+2 is also frame 0 of window fr+1, whose file the reference keeps
                k = str(fr * 2 + seq_i).zfill(digits)
                fio.write_png(os.path.join(out_dir, f"pred_{k}.png"), rgb_u8[seq_i].cpu().numpy())
                fio.write_png(os.path.join(out_dir, f"pred_YUV_{k}.png"), yuv_host[:, :, 3 * seq_i:3 * seq_i + 3])
                written.append(k)
        print(" <FISR processing> [%4d/%4d]-th input multiple data sample (stride1), time: %4.4f(minutes)  "
              % (fr + 1, num_fr - 2, (time.time() - start) / 60))
    tiles_serial = 1 if parallel == "tile" else num_patch[0] * num_patch[1]
    per_frame = float(np.mean(net.inf_time)) * tiles_serial if net.inf_time else float("nan")
    if world > 1:
        import torch.distributed as dist
        dist.barrier()
    print("######### Estimated Inference Time (per one output 4K frame): %.8f[s]  #########" % per_frame)
    print("          (timed: a tile's share of one fisr_forward_frames call = input assembly + forward of up to 16 tiles; the reference times sess.run of one tile, FISRnet.py:868-874)")
    return dict(frames=sorted(set(written)), out_dir=out_dir, inference_time_per_frame=per_frame)
