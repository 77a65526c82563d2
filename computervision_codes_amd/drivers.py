"""Stage drivers behind the reference's `Scripts/` entry points (SURVEY 8(b) "CLI"): same flag names, relative paths
(`./__checkpoint__/run_<version>/`, `../0-5fold/data_feats/run_<version>/`), checkpoint names and feature-file layout
as `Spatial_cnn/test.py`, `Spatial_transformer/test.py`, `Temporal_mstct/test.py` and `Temporal_tenco/run.py -e`.
Unknown flags are ignored like the reference's `parse_known_args`.  Two additions: `--dtype {fp32,bf16}` and, when
torch.distributed is initialised (torchrun), whole videos are sharded over ranks (extract.shard_videos)."""
from __future__ import annotations

import argparse
import sys
import os
import pickle
import time
from typing import Dict

import numpy as np
import torch

from . import cholect, extract, featfile
from .metrics import Recognition, final_report, gather_recognition, recognition_from


def extraction_batch(img_size, device_batch):
    from .spatial_transformer import extraction_batch as f      # (the transformer stage loads on demand)
    return f(img_size, device_batch)


def _common(p: argparse.ArgumentParser):
    p.add_argument("--model", type=str, default="rendezvous")
    p.add_argument("--version", type=str, default="")
    p.add_argument("--version1", type=str, default="")
    p.add_argument("-t", "--train", action="store_true")
    p.add_argument("-e", "--test", action="store_true")
    p.add_argument("--data_dir", type=str, default="/home/shuangchun/Data/Video/CholecT45/CholecT45")
    p.add_argument("--dataset_variant", type=str, default="cholect45-crossval")
    p.add_argument("-k", "--kfold", type=int, default=1)
    p.add_argument("--image_width", type=int, default=448)
    p.add_argument("--image_height", type=int, default=256)
    p.add_argument("-b", "--batch", type=int, default=32)
    p.add_argument("--loss_type", type=str, default="all")
    p.add_argument("--test_ckpt", type=str, default=None)
    p.add_argument("--gpu", type=str, default="0")
    p.add_argument("--seed", type=int, default=47)
    p.add_argument("--dtype", type=str, default="fp32", choices=["fp32", "bf16"])
    p.add_argument("--device_batch", type=int, default=512, help="frames per extraction pass on the GPU (results do not depend on it)")
    p.add_argument("--decode_workers", type=int, default=8, help="host threads decoding PNGs")
    p.add_argument("--png_decode", type=str, default="host", choices=["host", "device"],
                   help="device: inflate + PNG unfiltering on the GPU (mt4_png_inflate / mt4_png_unfilter_rgb8), the host only reads the files")


def _dist():
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


def _barrier():
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.barrier()


def _chlg(F) -> bool:
    """`set_chlg_eval` of the reference's drivers (`Spatial_cnn/run.py:122`): the challenge evaluation protocol (null triplets left out of the
    100-way AP) for the `*challenge*` dataset variants"""
    return "challenge" in str(getattr(F, "dataset_variant", ""))


def _log(path: str, msg: str):
    print(msg)
    os.makedirs(os.path.dirname(path), exist_ok=True)
    with open(path, "a+") as f:
        print(msg, file=f)


def _sigmoid(x: torch.Tensor) -> np.ndarray:
    return torch.sigmoid(x.float()).cpu().numpy()


def _write_report(logfile: str, m, loss_type: str, chlg: bool, style: str, pckl: str = None) -> Dict[str, float]:
    """the reference's closing report (`metrics.final_report`: per-category AP vectors, the mean-AP row with I / V / T disentangled from the
    triplet head, top-K rows) into the log file, and -- temporal drivers -- the pickled metric objects (`Temporal_tenco/run.py:529-533`:
    `{'ivt': mAP, 'i': mAPi, 'v': mAPv, 't': mAPt}`, here `metrics.Recognition` objects with ivtmetrics' attribute names)"""
    if pckl:
        os.makedirs(os.path.dirname(os.path.abspath(pckl)), exist_ok=True)
        with open(pckl, "wb") as f:
            pickle.dump({k: m[k] for k in ("ivt", "i", "v", "t")}, f)
    lines, res = final_report(m, loss_type, chlg, style)
    for ln in lines:
        _log(logfile, ln)
    return res


# ------------------------------------------------------------------------------------------------ Spatial_cnn/test.py
def _spatial_cnn_videos(F, model, vids, labels):
    """the per-video loop of `test_loop` (`Spatial_cnn/test.py:143-177`, `run.py:226-256`) over `vids`: -> ({video key -> feat [N,D]},
    {video -> {head -> (labels [N,K], sigmoid scores [N,K])}})"""
    feats_local: Dict[str, np.ndarray] = {}
    scores_local = {}
    dev_dec = F.png_decode == "device"

    def loader(v):
        ids_all = labels[v]["ivt"][:, 0]                       # file order, no shuffle, drop_last False (`test.py:227-242`)
        return lambda s, e: cholect.load_frames_device(F.data_dir, v, ids_all[s:e], F.image_height, F.image_width,   # decode on the host (or
                                                       workers=F.decode_workers, decode=F.png_decode)          # device), Resize on the GPU
    # ONE loader pipeline over all videos (`extract.extract_videos_device`): the first loads of the next video are read and decoded while this
    # one's last passes run.  The device PNG decoder runs one wave per frame and takes 75-105 ms per call for 512-2048 frames: it is handed loads
    # of 1024 frames, two in flight on streams of their own (sweep: profiles/r04_png_pipeline_sweep.txt -- 9.7-10.2 k frames/s from 480 x 854 files
    # through ResNet-50; the host reader alone delivers > 100 k files/s, what bounds the loop is inflate time + extractor time, which share the CUs).
    plan = [(v, len(labels[v]["ivt"]), loader(v)) for v in vids]
    for v, feat, lgs in extract.extract_videos_device(model, plan, F.device_batch, prefetch=2 if dev_dec else 1,
                                                          load_batch=(1023 // F.device_batch + 1) * F.device_batch if dev_dec else None):      # (>= 1024 frames per device decode)
        lab = labels[v]
        scores_local[v] = {key: (lab[key][:, 1:], torch.sigmoid(torch.from_numpy(lg)).numpy())       # `test.py:162-169`
                           for key, lg in zip(("i", "v", "t", "ivt"), lgs)}
        feats_local[featfile.video_key(v)] = np.array(feat)    # (own copy: the pinned staging buffer is released)
    return feats_local, scores_local


def spatial_cnn_eval(argv=None) -> Dict[str, float]:
    """`Spatial_cnn/run.py -e` (:503-560): the TEST-split videos through the best checkpoint and the closing report -- per-category AP, the
    mean-AP row (I / V / T disentangled from the 100-way triplet head when --loss_type all, head-wise otherwise, `:518-525`), top-5 / 10 / 20
    per component.  Under torchrun whole videos are sharded over the ranks and their (labels, scores) meet in one host-side gather, so N
    ranks log exactly the single-rank report; rank 0 writes it."""
    from .spatial_cnn import VideoNas
    p = argparse.ArgumentParser()
    _common(p)
    p.add_argument("--network", type=str, default="resnet18")
    p.add_argument("--student_dim", type=int, default=512)
    p.add_argument("--teacher_dim", type=int, default=1536)
    F, _ = p.parse_known_args(argv)
    F.train = False
    rank, world = _dist()
    kfold = F.kfold if "crossval" in F.dataset_variant else 0
    modelname = f"{F.model}_l{F.dataset_variant}_cholect{kfold}"
    model_dir = f"./__checkpoint__/run_{F.version}"
    logfile = os.path.join(model_dir, modelname + ".log")
    ckpt = F.test_ckpt or os.path.join(model_dir, modelname + ".pth")
    model = VideoNas(args=F, dtype=torch.float32 if F.dtype == "fp32" else torch.bfloat16).eval()
    model.load_state_dict(torch.load(ckpt, map_location="cpu"))
    _, _, videos = cholect.split_videos(F.dataset_variant, kfold)
    labels = {v: cholect.load_labels(F.data_dir, v) for v in videos}
    mine = extract.shard_videos(videos, [len(labels[v]["ivt"]) for v in videos], rank, world)
    _, scores_local = _spatial_cnn_videos(F, model, [videos[vi] for vi in mine], labels)
    m = gather_recognition(scores_local, videos)
    res = {}
    try:
        if rank == 0:
            res = _write_report(logfile, m, F.loss_type, _chlg(F), "spatial_cnn")
    finally:
        _barrier()
    return res


def spatial_cnn_test(argv=None) -> Dict[str, np.ndarray]:
    from .spatial_cnn import VideoNas
    p = argparse.ArgumentParser()
    _common(p)
    p.add_argument("--network", type=str, default="resnet18")
    p.add_argument("--student_dim", type=int, default=512)
    p.add_argument("--teacher_dim", type=int, default=1536)
    F, _ = p.parse_known_args(argv)
    F.train = False
    rank, world = _dist()
    modelname = f"{F.model}_l{F.dataset_variant}_cholect{F.kfold}"   # `test.py:126-128`
    model_dir = f"./__checkpoint__/run_{F.version}"
    logfile = os.path.join(model_dir, modelname + ".log")
    ckpt = F.test_ckpt or f"./__checkpoint__/run_{F.version}/rendezvous_l{F.dataset_variant}_cholect{F.kfold}.pth"
    model = VideoNas(args=F, dtype=torch.float32 if F.dtype == "fp32" else torch.bfloat16).eval()
    model.load_state_dict(torch.load(ckpt, map_location="cpu"))
    videos = cholect.extraction_videos(F.dataset_variant, F.kfold)
    labels = {v: cholect.load_labels(F.data_dir, v) for v in videos}
    mine = extract.shard_videos(videos, [len(labels[v]["ivt"]) for v in videos], rank, world)
    t0 = time.time()
    feats_local, scores_local = _spatial_cnn_videos(F, model, [videos[vi] for vi in mine], labels)
    merged = extract.gather_feats(feats_local)
    m = gather_recognition(scores_local, videos)               # the videos of ALL ranks in file order: N ranks log the 1-rank numbers
    all_feats = {featfile.video_key(v): merged[featfile.video_key(v)] for v in videos}
    if rank == 0:
        featfile.write_feats(featfile.feats_path("..", F.version, F.kfold, F.loss_type), all_feats)
        _log(logfile, f"save time:::::: : {time.time() - t0:.4f} secs")
        _log(logfile, " ".join(f"AP_{k}={m[k].compute_video_AP(ignore_null=_chlg(F))['mAP']:.4f}" for k in m) + f" (all {len(videos)} videos, world={world})")
    return all_feats


# ------------------------------------------------------------------------------------------------ Spatial_cnn/run.py -t
def _augment(im, rng, names):
    """the reference's PIL-side train augmentations (`Spatial_cnn/dataloader.py:89-100`; its dict lists 'contrast' twice, so the later
    RandomAutocontrast is the one in effect), drawn from `rng` (python `random.Random`)"""
    from PIL import Image, ImageOps
    for n in names:
        if n == "vflip" and rng.random() < 0.4:
            im = ImageOps.flip(im)
        elif n == "hflip" and rng.random() < 0.4:
            im = ImageOps.mirror(im)
        elif n == "contrast" and rng.random() < 0.5:
            im = ImageOps.autocontrast(im)
        elif n == "rot90":
            im = im.rotate(rng.uniform(-90.0, 90.0), resample=Image.NEAREST, expand=True)
    return im


def load_train_frames_u8(data_dir, video, frame_ids, height, width, rng, aug_names) -> np.ndarray:
    """`Resize -> augmentations -> Resize` of the train transform (`dataloader.py:153-162`) -> uint8 [N,H,W,3]"""
    from PIL import Image
    out = np.empty((len(frame_ids), height, width, 3), np.uint8)
    for i, fid in enumerate(frame_ids):
        with Image.open(os.path.join(data_dir, "data", video, "{}.png".format(str(int(fid)).zfill(6)))) as im:
            im = im.convert("RGB").resize((width, height), Image.BILINEAR)
            im = _augment(im, rng, aug_names)
            if im.size != (width, height):
                im = im.resize((width, height), Image.BILINEAR)
            out[i] = np.asarray(im)
    return out


def spatial_cnn_train(argv=None) -> Dict[str, float]:
    """`Spatial_cnn/run.py -t` (:296-470): student distillation.  Shuffled frames of all training videos in batches of --batch; with
    torchrun every rank takes its own batch of a step (frame-DDP: global batch = world x --batch), BatchNorm statistics stay per
    GPU and the flat gradient buffer is all-reduced over RCCL once per step.  SGD without momentum, LinearLR warm-up ->
    ExponentialLR per epoch, validation mAP every --val_interval epochs with `_latest.pth` / best `.pth` like `weight_mgt` (:258-269)."""
    import random

    from .spatial_cnn import VideoNas
    from .spatial_cnn_train import SpatialCnnTrainer
    from .tenco_train import lr_at_epoch
    from . import shapes, synth
    p = argparse.ArgumentParser()
    _common(p)
    p.add_argument("--network", type=str, default="resnet18")
    p.add_argument("--student_dim", type=int, default=512)
    p.add_argument("--teacher_dim", type=int, default=1536)
    p.add_argument("--teacher_feat_version", type=str, default="Q2L")
    p.add_argument("--teacher_pred_version", type=str, default="Q2LMSTCT")
    p.add_argument("--augmentation_list", type=str, nargs="*", default=["original", "vflip", "hflip", "contrast", "rot90"])
    p.add_argument("--epochs", type=int, default=100)
    p.add_argument("-w", "--warmups", type=int, nargs="+", default=[9, 18, 58])
    p.add_argument("-l", "--initial_learning_rates", type=float, nargs="+", default=[0.01, 0.01, 0.01])
    p.add_argument("--rates", type=float, nargs="+", default=[1, 0, 0.1])
    p.add_argument("--weight_decay", type=float, default=1e-5)
    p.add_argument("--temp", type=int, default=4)
    p.add_argument("--decay_rate", type=float, default=0.99)
    p.add_argument("--power", type=float, default=0.1)
    p.add_argument("--val_interval", type=int, default=1)
    p.add_argument("--pretrain_dir", type=str, default="")
    p.add_argument("--operand_dtype", type=str, default="fp32", choices=["fp32", "bf16"],
                   help="bf16: the convolutions' GEMM operands in bf16 (activations, gradients, weight copies), master weights and sums fp32")
    F, _ = p.parse_known_args(argv)
    if F.loss_type not in ("all", "i", "v", "t"):
        raise ValueError("--loss_type all | i | v | t (`Spatial_cnn/run.py:165-192`)")
    single = F.loss_type != "all"
    rank, world = _dist()
    kfold = F.kfold if "crossval" in F.dataset_variant else 0
    modelname = f"{F.model}_l{F.dataset_variant}_cholect{kfold}"
    model_dir = f"./__checkpoint__/run_{F.version}"
    logfile = os.path.join(model_dir, modelname + ".log")
    ckpt, latest = os.path.join(model_dir, modelname + ".pth"), os.path.join(model_dir, modelname + "_latest.pth")
    val_interval = F.epochs - 1 if F.val_interval == -1 else F.val_interval
    tr = SpatialCnnTrainer(F.network, lr=F.initial_learning_rates[2], weight_decay=F.weight_decay, rates=F.rates, temp=float(F.temp),
                           teacher_dim=F.teacher_dim, loss_type=F.loss_type,
                           operand_dtype=torch.bfloat16 if F.operand_dtype == "bf16" else torch.float32)
    table = shapes.spatial_cnn_shapes(F.network, F.student_dim, F.teacher_dim, F.loss_type)
    sd = synth.fill_from_shapes(table, seed=F.seed)          # no torch.nn init here: deterministic synthetic start
    for src in (F.pretrain_dir, latest):                     # `load_model` (:272-278): keys present in the model, strict=False
        if src and os.path.exists(src):
            sd.update({k: v for k, v in torch.load(src, map_location="cpu").items() if k in sd})
    tr.load_state_dict(sd)
    train_videos, val_videos, _ = cholect.split_videos(F.dataset_variant, kfold)
    labels = {v: cholect.load_labels(F.data_dir, v) for v in train_videos + val_videos}
    tdir = lambda ver, task, kind: featfile.feats_path("..", ver, kfold, task, kind)
    # the teacher files feed the distillation losses only (`dataloader.py:216-238` loads them regardless; a single-task run never uses them)
    tpred = {} if single else {t: featfile.read_feats(tdir(F.teacher_pred_version, t, "pred")) for t in "ivt"}
    tfeat = {} if single else {t: featfile.read_feats(tdir(F.teacher_feat_version, t, "feats")) for t in "ivt"}
    samples = [(v, i) for v in train_videos for i in range(len(labels[v]["ivt"]))]
    order_rng, aug_rng = random.Random(F.seed), random.Random(F.seed * 1000003 + rank)
    eval_args = argparse.Namespace(**vars(F))
    eval_args.train = False
    best, last = 0.0, {}
    for epoch in range(F.epochs):
        tr.lr = lr_at_epoch(epoch, F.initial_learning_rates[2], F.power, F.warmups[2], F.decay_rate)
        order = list(samples)
        order_rng.shuffle(order)                             # the same permutation on every rank
        nb = (len(order) + F.batch - 1) // F.batch
        steps = (nb + world - 1) // world
        t0, tot = time.time(), 0.0
        for s in range(steps):
            bi = (s * world + rank) % nb
            batch = order[bi * F.batch:(bi + 1) * F.batch]
            frames = np.concatenate([load_train_frames_u8(F.data_dir, v, [labels[v]["ivt"][i, 0]], F.image_height, F.image_width, aug_rng,
                                                          F.augmentation_list) for v, i in batch])
            lab = [torch.from_numpy(np.stack([labels[v][k][i, 1:] for v, i in batch])) for k in ("i", "v", "t", "ivt")]
            key = lambda v: featfile.video_key(v)
            tp = [] if single else [torch.from_numpy(np.stack([tpred[t][key(v)][i] for v, i in batch]).astype(np.float32)) for t in "ivt"]
            tf = [] if single else [torch.from_numpy(np.stack([tfeat[t][key(v)][i] for v, i in batch]).astype(np.float32)) for t in "ivt"]
            terms = tr.train_step(torch.from_numpy(frames).cuda(), lab, tp, tf)
            tot += terms["loss"]
        last = {"loss": tot / steps, "lr": tr.lr}
        if rank == 0:
            _log(logfile, f"Traning | lr: {tr.lr:.6f} | epoch {epoch} | loss {tot / steps:.4f} | {time.time() - t0:.2f} secs")
        if epoch % val_interval == 0 and rank == 0:          # `weight_mgt`: latest every validation, best by triplet mAP
            state = tr.state_dict()
            torch.save(state, latest)
            model = VideoNas(args=eval_args, dtype=torch.float32).eval()
            model.load_state_dict(state)
            vt = F.loss_type if single else "ivt"               # the head the validation mAP is taken on (`run.py:416-451`)
            m = Recognition({"i": 6, "v": 10, "t": 15, "ivt": 100}[vt])
            for v in val_videos:
                lv = labels[v][vt]
                vb = max(F.batch, min(getattr(F, "device_batch", F.batch), 256))     # validation passes in device batches (results do not depend on it)
                for s0 in range(0, len(lv), vb):
                    fr = cholect.load_frames_device(F.data_dir, v, lv[s0:s0 + vb, 0], F.image_height, F.image_width,
                                                    workers=getattr(F, "decode_workers", 0), decode=getattr(F, "png_decode", "host"))
                    m.update(lv[s0:s0 + vb, 1:], _sigmoid(model.extract_u8(fr)["ivt".index(vt) if single else 3][1]))
                m.video_end()
            score = float(m.compute_video_AP(ignore_null=_chlg(F))["mAP"]) if val_videos else 0.0
            last["val_mAP_ivt"] = score
            if score > best or not os.path.exists(ckpt):
                best = max(best, score)
                torch.save(state, ckpt)
                _log(logfile, f">>> Saving checkpoint for epoch {epoch + 1} at {ckpt}, time {time.ctime()} ")
            _log(logfile, f"\t\t\t\t\t\t\t video-wise | eta {time.time() - t0:.2f} secs | mAP => ivt: [{score:.5f}] ")
    return last


# ------------------------------------------------------------------------------------------------ Temporal_tenco/run.py -e
def tenco_eval(argv=None) -> Dict[str, float]:
    from .temporal_tenco import VideoNas
    p = argparse.ArgumentParser()
    _common(p)
    p.add_argument("--num_layers_PG", default=11, type=int)
    p.add_argument("--num_layers_R", default=10, type=int)
    p.add_argument("--num_R", default=3, type=int)
    p.add_argument("--fpn", action="store_true")
    p.add_argument("--mask", action="store_true")
    p.add_argument("--output", default=False, type=bool)
    p.add_argument("--hier", default=False, type=bool)
    p.add_argument("--input_dim", type=int, default=512)
    p.add_argument("--epochs", type=int, default=100)
    p.add_argument("-w", "--warmups", type=int, nargs="+", default=[9, 18, 58])
    p.add_argument("-l", "--initial_learning_rates", type=float, nargs="+", default=[0.01, 0.01, 0.01])
    p.add_argument("--weight_decay", type=float, default=1e-5)
    p.add_argument("--decay_rate", type=float, default=0.99)
    p.add_argument("--power", type=float, default=0.1)
    p.add_argument("--val_interval", type=int, default=1)
    F, _ = p.parse_known_args(argv)
    if F.train:
        _tenco_train(F)
        if not F.test:
            return {}
    if _dist()[0] != 0:      # under torchrun the evaluation, its log lines and the mAP pickle belong to rank 0 alone
        _barrier()
        return {}
    try:
        return _tenco_eval_rank0(F)
    finally:
        _barrier()


def _tenco_eval_rank0(F) -> Dict[str, float]:
    from .temporal_tenco import VideoNas
    modelname = f"{F.model}_l8_cholect{F.dataset_variant}_k{F.kfold}_batchnorm_lowres"   # `run.py:137-142`
    model_dir = f"./__checkpoint__/run_{F.version}"
    logfile = os.path.join(model_dir, modelname + ".log")
    ckpt = F.test_ckpt or os.path.join(model_dir, modelname + ".pth")
    if not os.path.exists(ckpt):   # the shipped scripts always pass --test_ckpt ..._latest.pth (Scripts/test_fold1.sh)
        ckpt = os.path.join(model_dir, modelname + "_latest.pth")
    model = VideoNas(F, F.num_layers_PG, F.num_layers_R, F.num_R, 512, F.input_dim, 100).eval()
    sd = torch.load(ckpt, map_location="cpu")
    model.load_state_dict({k: v for k, v in sd.items() if k in dict(model._table)}, strict=False)   # `run.py:520`
    _, _, test_videos = cholect.split_videos(F.dataset_variant, F.kfold)
    feats = featfile.read_feats(featfile.feats_path("..", F.version1, F.kfold, "all"))
    t0 = time.time()
    m = recognition_from(_tenco_scores(model, feats, test_videos, F.data_dir), test_videos)     # (rank 0 alone runs this pass)
    _log(logfile, f"eta {time.time() - t0:.3f} secs")
    # `run.py:529-570`: the pickled metric objects, then head-wise ('singletest') and disentangled per-category AP and both mean-AP rows
    res = _write_report(logfile, m, F.loss_type, _chlg(F), "temporal_tenco", pckl=os.path.join(model_dir, f"mAPs_k{F.kfold}.pckl"))
    return res


def _tenco_scores(model, feats, vids, data_dir):
    """`test_loop` of `Temporal_tenco/run.py:238-270`: whole video, batch 1, the finest FPN level's logits [K,T] -> sigmoid [T,K]"""
    out_scores = {}
    for v in vids:
        lab = cholect.load_labels(data_dir, v)
        x = torch.from_numpy(feats[featfile.video_key(v)]).unsqueeze(0).cuda()
        out, out_i, out_v, out_t, _, _ = model(x, False)
        n = x.shape[1]
        out_scores[v] = {key: (lab[key][:n, 1:], _sigmoid(lg[0][0].transpose(0, 1))) for key, lg in (("ivt", out), ("i", out_i), ("v", out_v), ("t", out_t))}
    return out_scores


def _tenco_train(F):
    """`Temporal_tenco/run.py -t` (:181-235, :341-348, :260-271): whole-video batch 1, SGD without momentum, LinearLR warm-up ->
    ExponentialLR per epoch, `_latest.pth` after every epoch.  With torchrun the shuffled videos of an epoch are dealt round-robin
    to the ranks (one video per rank per step) and the flat gradient buffer is all-reduced over RCCL each step."""
    import random

    from .tenco_train import TencoTrainer, lr_at_epoch
    rank, world = _dist()
    modelname = f"{F.model}_l8_cholect{F.dataset_variant}_k{F.kfold}_batchnorm_lowres"
    model_dir = f"./__checkpoint__/run_{F.version}"
    logfile = os.path.join(model_dir, modelname + ".log")
    if not F.fpn:
        # the reference's own train loop cannot run a model without --fpn: `out_list_i / _v / _t` stay empty (`network.py:56-66`), so `loss_i`,
        # `loss_v`, `loss_t` stay the int 0 they start as (`run.py:190`) and `loss_i.item()` raises AttributeError at `run.py:214` in the first step
        raise NotImplementedError("Temporal_tenco training needs --fpn (Scripts/train_fold1.sh:28): without it the reference's train loop itself fails "
                                  "in its first step (run.py:190,214: .item() on the int 0 that loss_i stays when the model returns no per-component logits)")
    tr = TencoTrainer(F.num_layers_PG, F.num_layers_R, F.num_R, 512, F.input_dim, lr=F.initial_learning_rates[2], weight_decay=F.weight_decay,
                      hier=bool(getattr(F, "hier", False)))      # --hier True: pooled refinement levels (`network.py:147,154-155`, `run.py:159-179`)
    from . import shapes, synth
    init = os.path.join(model_dir, modelname + "_latest.pth")
    if os.path.exists(init):
        tr.load_state_dict(torch.load(init, map_location="cpu"))
    else:   # no torch.nn init here: deterministic synthetic start (the reference starts from torch's default init)
        tr.load_state_dict(synth.fill_from_shapes(shapes.tenco_shapes(F.num_layers_PG, F.num_layers_R, F.num_R, 512, F.input_dim, 100, fpn=True),
                                                  seed=F.seed))
    train_videos, val_videos, _ = cholect.split_videos(F.dataset_variant, F.kfold)
    val_interval = F.epochs - 1 if F.val_interval == -1 else max(1, F.val_interval)
    best, best_path, vmodel = 0.0, os.path.join(model_dir, modelname + ".pth"), None
    feats = featfile.read_feats(featfile.feats_path("..", F.version1, F.kfold, "all"))
    # features and labels of every training video are uploaded ONCE (a per-step pageable host->device copy stalls the step)
    xs, zs = {}, {}
    for v in train_videos:
        lab = cholect.load_labels(F.data_dir, v)
        xs[v] = torch.from_numpy(feats[featfile.video_key(v)]).unsqueeze(0).cuda()
        zs[v] = tr.prepare_labels({k: torch.from_numpy(lab[n][:, 1:]) for k, n in (("", "ivt"), ("_i", "i"), ("_v", "v"), ("_t", "t"))})
    rng = random.Random(F.seed)
    gen = torch.Generator().manual_seed(F.seed + rank)
    for epoch in range(F.epochs):
        tr.lr = lr_at_epoch(epoch, F.initial_learning_rates[2], F.power, F.warmups[2], F.decay_rate)
        order = list(train_videos)
        rng.shuffle(order)                                         # same permutation on every rank
        steps = (len(order) + world - 1) // world
        t0, tot = time.time(), 0.0
        for s in range(steps):
            v = order[(s * world + rank) % len(order)]
            x = xs[v]
            masks = tr.draw_masks(x.shape[1], gen)                 # Dropout2d + per-layer Dropout are always on in train mode
            if not F.mask:                                         # (`network.py:123-127,194-196`); --mask gates the 75 % input mask only
                masks["input_mask"] = None                         # (`network.py:43-48`)
            loss, _ = tr.train_step(x, zs[v], masks=masks)
            tot += loss
        if rank == 0:
            _log(logfile, f"Traning | lr: {tr.lr:.6f} | epoch {epoch} | loss {tot / steps:.4f} | {time.time() - t0:.2f} secs")
            state = tr.state_dict()
            torch.save(state, init + ".tmp")
            os.replace(init + ".tmp", init)                        # readers never see a half-written checkpoint
            if epoch % val_interval == 0:                          # validation + `weight_mgt` (`run.py:416-452,270-282`): best `.pth` by the triplet mAP
                t1 = time.time()
                if vmodel is None:
                    from .temporal_tenco import VideoNas
                    vmodel = VideoNas(F, F.num_layers_PG, F.num_layers_R, F.num_R, 512, F.input_dim, 100).eval()
                vmodel.load_state_dict(state)
                vm = recognition_from(_tenco_scores(vmodel, feats, val_videos, F.data_dir), val_videos) if val_videos else None
                head = F.loss_type if F.loss_type in ("i", "v", "t") else "ivt"
                score = float(vm[head].compute_video_AP()["mAP"]) if vm else 0.0
                if score > best or not os.path.exists(best_path):
                    best = max(best, score)
                    torch.save(state, best_path + ".tmp")
                    os.replace(best_path + ".tmp", best_path)
                    _log(logfile, f">>> Saving checkpoint for epoch {epoch + 1} at {best_path}, time {time.ctime()} ")
                ivt = float(vm["ivt"].compute_video_AP("ivt", ignore_null=_chlg(F))["mAP"]) if vm else 0.0
                _log(logfile, f"\t\t\t\t\t\t\t video-wise | eta {time.time() - t1:.2f} secs | mAP => ivt: [{ivt:.5f}] ")
    _barrier()                                                     # the last checkpoint is on disk before any rank goes on to -e


# ------------------------------------------------------------------------------------------------ Spatial_transformer/test.py
def spatial_transformer_test(argv=None) -> Dict[str, np.ndarray]:
    from .spatial_transformer import build_q2l
    p = argparse.ArgumentParser()
    _common(p)
    p.add_argument("--backbone", type=str, default="swin_L_384_22k")
    p.add_argument("--img_size", type=int, default=384)
    p.add_argument("--hidden_dim", type=int, default=1536)
    F, _ = p.parse_known_args(argv)
    # only the CHECKPOINT directory carries the task suffix (`test.py:93-95`); the feature file goes to run_<version as given>
    # (`test.py:364-372`: `version1`), which is where Temporal_mstct and the student's dataloader look for it
    ckpt_version = F.version + ("_" + F.loss_type if F.loss_type != "all" else "")
    ckpt = F.test_ckpt or f"./__checkpoint__/run_{ckpt_version}/rendezvous_l{F.dataset_variant}_cholect{F.kfold}.pth"
    model = build_q2l(F, dtype=torch.float32 if F.dtype == "fp32" else torch.bfloat16).eval()
    model.load_state_dict(torch.load(ckpt, map_location="cpu"), strict=True)
    rank, world = _dist()
    videos = cholect.extraction_videos(F.dataset_variant, F.kfold)
    labels = {v: cholect.load_labels(F.data_dir, v) for v in videos}
    mine = extract.shard_videos(videos, [len(labels[v]["ivt"]) for v in videos], rank, world)
    feats_local = {}
    for vi in mine:
        v, chunks = videos[vi], []
        ids_all = labels[v]["ivt"][:, 0]
        # device batches (a frame's feature does not depend on the batch it rides in), decode on --decode_workers threads (or on the device), the
        # video's features stay on the GPU until its end: one D2H per video instead of one synchronous copy per --batch frames (`test.py:357-376`)
        step = extraction_batch(F.img_size, F.device_batch)
        load = lambda s, e: cholect.load_frames_device(F.data_dir, v, ids_all[s:e], F.img_size, F.img_size, workers=F.decode_workers,
                                                       decode=F.png_decode)
        dev_dec = F.png_decode == "device"                     # (the device decoder wants >= 1024 frames per call; two loads run ahead)
        lb = (1023 // step + 1) * step if dev_dec else step
        spans = [(s, min(len(ids_all), s + lb)) for s in range(0, len(ids_all), lb)]
        for span in extract.iter_chunks(spans, load, 2 if dev_dec else 1):      # the next load is decoded while this one runs
            for s in range(0, span.shape[0], step):
                chunks.append(model(span[s:s + step])[3][0].float())
        feats_local[featfile.video_key(v, "transformer")] = torch.vstack(chunks).cpu().numpy()
    merged = extract.gather_feats(feats_local)
    if rank == 0:
        featfile.write_feats(featfile.feats_path("..", F.version, F.kfold, F.loss_type), merged)
    return merged


def _q2l_scores(F, model, vids, labels):
    """`test_loop` of `Spatial_transformer/run.py:231-262` over `vids`: device batches in file order, the teacher features the loader hands a
    `loss_type all` model off the train split are zeros (`dataloader.py:240-246`) -> {video -> {head -> (labels, sigmoid scores)}}; heads the
    model does not have score sigmoid(0) like the reference's zero logits (`network.py:84-89`)"""
    out_scores = {}
    single = F.loss_type != "all"
    for v in vids:
        ids_all = labels[v]["ivt"][:, 0]
        step = extraction_batch(F.img_size, F.device_batch)
        load = lambda s, e: cholect.load_frames_device(F.data_dir, v, ids_all[s:e], F.img_size, F.img_size, workers=F.decode_workers, decode=F.png_decode)
        spans = [(s, min(len(ids_all), s + step)) for s in range(0, len(ids_all), step)]
        acc = {k: [] for k in ("i", "v", "t", "ivt")}
        for span in extract.iter_chunks(spans, load, 1):
            zt = [] if single else [torch.zeros((span.shape[0], F.teacher_dim), device=span.device)] * 3
            o = model(span, *zt)
            for gi, key in enumerate(("i", "v", "t", "ivt")):
                acc[key].append(_sigmoid(o[gi][1]))
        out_scores[v] = {key: (labels[v][key][:, 1:], np.concatenate(acc[key])) for key in acc}
    return out_scores


def spatial_transformer_eval(argv=None) -> Dict[str, float]:
    """`Spatial_transformer/run.py -e` (:482-527): the TEST-split videos through the best checkpoint of run_<version>[_<task>]/ and the closing
    report (per-category AP, mean-AP row; I / V / T from the component heads for a single-task teacher, disentangled from the triplet head for
    --loss_type all).  Videos sharded over the ranks, (labels, scores) gathered on the host, rank 0 writes: N ranks log the 1-rank report."""
    from .spatial_transformer import build_q2l
    p = argparse.ArgumentParser()
    _common(p)
    p.add_argument("--backbone", type=str, default="swin_L_384_22k")
    p.add_argument("--img_size", type=int, default=384)
    p.add_argument("--hidden_dim", type=int, default=1536)
    p.add_argument("--teacher_dim", type=int, default=512)
    F, _ = p.parse_known_args(argv)
    rank, world = _dist()
    kfold = F.kfold if "crossval" in F.dataset_variant else 0
    single = F.loss_type != "all"
    modelname = f"{F.model}_l{F.dataset_variant}_cholect{kfold}"
    model_dir = f"./__checkpoint__/run_{F.version}" + (f"_{F.loss_type}" if single else "")   # `run.py:86-88`
    logfile = os.path.join(model_dir, modelname + ".log")
    ckpt = F.test_ckpt or os.path.join(model_dir, modelname + ".pth")
    model = build_q2l(F, dtype=torch.float32 if F.dtype == "fp32" else torch.bfloat16).eval()
    model.load_state_dict(torch.load(ckpt, map_location="cpu"), strict=True)
    _, _, videos = cholect.split_videos(F.dataset_variant, kfold)
    labels = {v: cholect.load_labels(F.data_dir, v) for v in videos}
    mine = extract.shard_videos(videos, [len(labels[v]["ivt"]) for v in videos], rank, world)
    m = gather_recognition(_q2l_scores(F, model, [videos[vi] for vi in mine], labels), videos)
    res = {}
    try:
        if rank == 0:
            res = _write_report(logfile, m, F.loss_type, _chlg(F), "spatial_transformer")
    finally:
        _barrier()
    return res


# ------------------------------------------------------------------------------------------------ Temporal_mstct/test.py
def mstct_test(argv=None):
    from .temporal_mstct import VideoNas
    p = argparse.ArgumentParser()
    _common(p)
    p.add_argument("--input_dim", type=int, default=1536)
    p.add_argument("--final_embedding_dim", type=int, default=512)
    F, _ = p.parse_known_args(argv)
    F.in_feat_dim = F.input_dim
    # checkpoint directory: run_<version>_<task> for a single-task teacher (`test.py:88-90,131,326`); the feature / prediction files it
    # writes go to run_<version as given> (`test.py:342-366`) and its input comes from run_<version1> (`dataloader_test.py:220`)
    model_dir = f"./__checkpoint__/run_{F.version}" + ("_" + F.loss_type if F.loss_type != "all" else "")
    modelname = f"{F.model}_l8_cholect{F.dataset_variant}_k{F.kfold}_batchnorm_lowres"
    ckpt = F.test_ckpt or os.path.join(model_dir, modelname + "latest.pth")              # no underscore (`run.py:268`)
    model = VideoNas(F, [256, 384, 576, 864], 2, 8, 8, F.input_dim, F.final_embedding_dim,
                     dtype=torch.float32 if F.dtype == "fp32" else torch.bfloat16).eval()
    model.load_state_dict(torch.load(ckpt, map_location="cpu"))
    feats = featfile.read_feats(featfile.feats_path("..", F.version1, F.kfold, F.loss_type))
    out_feats, out_preds = {}, {}
    gi = {"i": 0, "v": 1, "t": 2, "ivt": 3}[F.loss_type]
    graphed = None                                                                         # full chunks: one hipGraph replay each (~100 launches, launch-bound)
    # under torchrun (`Scripts/train_fold1.sh` with NGPU > 1 runs `run.py -t -e`) whole videos are sharded over the ranks like the spatial
    # extractors' (no data-path collective), the per-rank dicts meet in one host-side gather and rank 0 alone writes the two files
    rank, world = _dist()
    keys = list(feats.keys())
    mine = set(extract.shard_videos(keys, [feats[k].shape[0] for k in keys], rank, world))
    for ki, (key, f) in enumerate(feats.items()):
        if ki not in mine:
            continue
        fs, ps = [], []
        for s in range(0, f.shape[0], 256):                                                # non-overlapping 256-frame chunks
            x = torch.from_numpy(f[s:s + 256]).unsqueeze(0).cuda()
            if x.shape[1] == 256:
                if graphed is None:
                    from .graph import GraphedForward
                    graphed = GraphedForward(lambda xx: model.forward_btd(xx), [x])
                o = graphed(x)
            else:
                o = model.forward_btd(x)
            ps.append(o[gi][0][0].float().cpu())                                           # raw logits [T,K]
            fs.append(o[3][1][0].transpose(0, 1).float().cpu())                            # concat feature [T,2048]
        out_feats[key], out_preds[key] = torch.vstack(fs).numpy(), torch.vstack(ps).numpy()
    out_feats, out_preds = extract.gather_feats(out_feats), extract.gather_feats(out_preds)
    out_feats, out_preds = {k: out_feats[k] for k in keys}, {k: out_preds[k] for k in keys}      # file order = input order, whatever the sharding
    try:
        if rank == 0:
            featfile.write_feats(featfile.feats_path("..", F.version, F.kfold, F.loss_type, "feats"), out_feats)
            featfile.write_feats(featfile.feats_path("..", F.version, F.kfold, F.loss_type, "pred"), out_preds)
    finally:
        _barrier()                                                                         # the files exist before any rank goes on to the next stage
    return out_feats, out_preds


def _mstct_scores(model, feats, vids, data_dir, loss_type):
    """`test_loop` of `Temporal_mstct/run.py:237-262` behind its batch-256 loaders (`:371,378`): non-overlapping 256-frame chunks, each an
    independent window; the heads the single-task model lacks are zero logits (`network.py:85-99`) = sigmoid 0.5"""
    out_scores = {}
    gi = {"i": 0, "v": 1, "t": 2, "ivt": 3}[loss_type]
    for v in vids:
        lab = cholect.load_labels(data_dir, v)
        key = featfile.video_key(v)
        f = feats[key] if key in feats else feats[v[3:]]
        ps = []
        for s in range(0, f.shape[0], 256):
            o = model.forward_btd(torch.from_numpy(f[s:s + 256]).unsqueeze(0).cuda())
            ps.append(_sigmoid(o[gi][0][0]))
        p_own = np.concatenate(ps)
        n = p_own.shape[0]                                          # (a feature file may hold fewer frames than the label file lists: the first n)
        out_scores[v] = {h: (lab[h][:n, 1:], p_own if h == loss_type else np.full(lab[h][:n, 1:].shape, 0.5)) for h in ("i", "v", "t", "ivt")}
    return out_scores


def mstct_eval(argv=None) -> Dict[str, float]:
    """`Temporal_mstct/run.py -e` (:527-580): the TEST-split videos in 256-frame chunks through the checkpoint of run_<version>[_<task>]/ (best
    `.pth`, else `latest.pth`), the pickled metric objects (`mAPs.pckl` in the working directory, `:546-549`) and the closing report.  Rank 0
    alone (a window takes < 1 ms)."""
    from .temporal_mstct import VideoNas
    p = argparse.ArgumentParser()
    _common(p)
    p.add_argument("--input_dim", type=int, default=1536)
    p.add_argument("--final_embedding_dim", type=int, default=512)
    F, _ = p.parse_known_args(argv)
    if _dist()[0] != 0:
        _barrier()
        return {}
    try:
        model_dir = f"./__checkpoint__/run_{F.version}" + ("_" + F.loss_type if F.loss_type != "all" else "")
        modelname = f"{F.model}_l8_cholect{F.dataset_variant}_k{F.kfold}_batchnorm_lowres"
        logfile = os.path.join(model_dir, modelname + ".log")
        ckpt = F.test_ckpt or os.path.join(model_dir, modelname + ".pth")
        if not os.path.exists(ckpt):
            ckpt = os.path.join(model_dir, modelname + "latest.pth")                         # no underscore (`run.py:268`)
        model = VideoNas(F, [256, 384, 576, 864], 2, 8, 8, F.input_dim, F.final_embedding_dim,
                         dtype=torch.float32 if F.dtype == "fp32" else torch.bfloat16).eval()
        model.load_state_dict(torch.load(ckpt, map_location="cpu"))
        feats = featfile.read_feats(featfile.feats_path("..", F.version1, F.kfold, F.loss_type))
        _, _, test_videos = cholect.split_videos(F.dataset_variant, F.kfold)
        m = recognition_from(_mstct_scores(model, feats, test_videos, F.data_dir, F.loss_type), test_videos)
        return _write_report(logfile, m, F.loss_type, _chlg(F), "temporal_mstct", pckl="mAPs.pckl")
    finally:
        _barrier()


# ------------------------------------------------------------------------------------------------ teacher run.py entry points
def _wants_train(argv) -> bool:
    argv = list(sys.argv[1:] if argv is None else argv)
    return "-t" in argv or "--train" in argv


# `Spatial_transformer/models/backbone.py:31-41` (get_model_path)
SWIN_PRETRAIN_FILES = {"swin_L_384_22k": "swin_large_patch4_window12_384_22k.pth", "swin_B_384_22k": "swin_base_patch4_window12_384_22k.pth",
                       "swin_T_224_1k": "swin_tiny_patch4_window7_224.pth"}


def spatial_transformer_train(argv=None) -> Dict[str, float]:
    """`Spatial_transformer/run.py -t` (:150-229, 296-470): the single-task teachers of the recipe (`Scripts/train_fold1.sh:12`, --loss_type
    i | v | t) and the four-decoder distillation variant (--loss_type all, `run.py:183-197`: hard + DistillKL + feature-MSE terms with --rates,
    teacher predictions / features from the files `dataloader.py:216-238` reads, zeros at validation `:240-246`).  Shuffled
    frames of all training videos in batches of --batch, the train transform at img_size x img_size (`dataloader.py:154-161`), DropPath and
    the transformer's dropout drawn per step on the device, SGD without momentum (`run.py:360`), LinearLR warm-up -> ExponentialLR per
    epoch, validation mAP of the task's head every --val_interval epochs with `_latest.pth` / best `.pth` (`weight_mgt`, :265-277).
    With torchrun every rank takes its own batch of a step and the flat gradient buffer is all-reduced over RCCL once per step."""
    import random

    from .q2l_train import Q2LTrainer
    from .spatial_transformer import build_q2l
    from .tenco_train import lr_at_epoch
    from . import shapes, synth
    p = argparse.ArgumentParser()
    _common(p)
    p.add_argument("--backbone", type=str, default="swin_L_384_22k")
    p.add_argument("--img_size", type=int, default=384)
    p.add_argument("--hidden_dim", type=int, default=1536)
    p.add_argument("--augmentation_list", type=str, nargs="*", default=["original", "vflip", "hflip", "contrast", "rot90"])
    p.add_argument("--epochs", type=int, default=100)
    p.add_argument("-w", "--warmups", type=int, nargs="+", default=[9, 18, 58])
    p.add_argument("-l", "--initial_learning_rates", type=float, nargs="+", default=[0.01, 0.01, 0.01])
    p.add_argument("--weight_decay", type=float, default=1e-5)
    p.add_argument("--decay_rate", type=float, default=0.99)
    p.add_argument("--power", type=float, default=0.1)
    p.add_argument("--val_interval", type=int, default=1)
    p.add_argument("--pretrain_dir", type=str, default="")
    p.add_argument("--drop_path_rate", type=float, default=0.1)          # `swin_transformer.py:488`
    p.add_argument("--operand_dtype", type=str, default="fp32", choices=["fp32", "bf16"],
                   help="bf16: the nn.Linear GEMMs on bf16 operand copies (fp32 activations, accumulation and master weights)")
    p.add_argument("--rates", type=float, nargs="+", default=[1, 0, 0.1])          # `run.py:66`
    p.add_argument("--temp", type=int, default=4)                                 # `run.py:70`
    p.add_argument("--teacher_dim", type=int, default=512)                        # `run.py:82`
    # the reference's dataloader reads args.teacher_pred_version / teacher_feat_version (`dataloader.py:217-238`) though its run.py declares
    # neither flag; same names and defaults as the student's `Spatial_cnn/run.py`
    p.add_argument("--teacher_feat_version", type=str, default="Q2L")
    p.add_argument("--teacher_pred_version", type=str, default="Q2LMSTCT")
    F, _ = p.parse_known_args(argv)
    if F.loss_type not in ("i", "v", "t", "all"):
        raise ValueError("--loss_type i | v | t | all (`Spatial_transformer/run.py:168-197`)")
    single = F.loss_type != "all"
    F.student_dim = F.hidden_dim                                             # `run.py:93`
    rank, world = _dist()
    kfold = F.kfold if "crossval" in F.dataset_variant else 0
    modelname = f"{F.model}_l{F.dataset_variant}_cholect{kfold}"
    model_dir = f"./__checkpoint__/run_{F.version}" + (f"_{F.loss_type}" if single else "")   # `run.py:86-88`
    logfile = os.path.join(model_dir, modelname + ".log")
    ckpt, latest = os.path.join(model_dir, modelname + ".pth"), os.path.join(model_dir, modelname + "_latest.pth")
    val_interval = F.epochs - 1 if F.val_interval == -1 else F.val_interval
    tr = Q2LTrainer(F.backbone, F.img_size, F.hidden_dim, F.loss_type, lr=F.initial_learning_rates[2], weight_decay=F.weight_decay,
                    drop_path_rate=F.drop_path_rate, operand_dtype=torch.bfloat16 if F.operand_dtype == "bf16" else torch.float32,
                    teacher_dim=F.teacher_dim, rates=F.rates, temp=float(F.temp))
    table = shapes.q2l_param_shapes(F.backbone, F.img_size, F.hidden_dim, F.loss_type, teacher_dim=F.teacher_dim)
    sd = synth.fill_from_shapes(table, seed=F.seed)          # deterministic synthetic start when no pretrained file is on disk
    # `build_backbone` (`backbone.py:188-196`): the upstream Swin checkpoint ../Pretrain/<file> ('model' entry, `head.*` dropped) into the backbone
    swin_file = os.path.join("..", "Pretrain", SWIN_PRETRAIN_FILES.get(F.backbone, ""))
    if os.path.isfile(swin_file):
        up = torch.load(swin_file, map_location="cpu")
        up = up.get("model", up)
        hit = {"backbone.0." + k: v for k, v in up.items() if "head" not in k and ("backbone.0." + k) in sd and tuple(v.shape) == tuple(sd["backbone.0." + k].shape)}
        sd.update(hit)
        if rank == 0:
            _log(logfile, f"backbone: {len(hit)} tensors from {swin_file}")
    for src in (F.pretrain_dir, latest):                     # `load_model` (:280-287): keys present in the model, strict=False
        if src and os.path.exists(src):
            sd.update({k: v for k, v in torch.load(src, map_location="cpu").items() if k in sd})
    tr.load_state_dict(sd)
    train_videos, val_videos, _ = cholect.split_videos(F.dataset_variant, kfold)
    labels = {v: cholect.load_labels(F.data_dir, v) for v in train_videos + val_videos}
    samples = [(v, i) for v in train_videos for i in range(len(labels[v]["ivt"]))]
    tdir = lambda ver, task, kind: featfile.feats_path("..", ver, kfold, task, kind)
    tpred = {} if single else {t: featfile.read_feats(tdir(F.teacher_pred_version, t, "pred")) for t in "ivt"}
    tfeat = {} if single else {t: featfile.read_feats(tdir(F.teacher_feat_version, t, "feats")) for t in "ivt"}
    order_rng, aug_rng = random.Random(F.seed), random.Random(F.seed * 1000003 + rank)
    eval_args = argparse.Namespace(**vars(F))
    best, last, step_no = 0.0, {}, 0
    for epoch in range(F.epochs):
        tr.lr = lr_at_epoch(epoch, F.initial_learning_rates[2], F.power, F.warmups[2], F.decay_rate)
        order = list(samples)
        order_rng.shuffle(order)                             # the same permutation on every rank
        nb = (len(order) + F.batch - 1) // F.batch
        steps = (nb + world - 1) // world
        t0, tot = time.time(), 0.0
        for s in range(steps):
            bi = (s * world + rank) % nb
            batch = order[bi * F.batch:(bi + 1) * F.batch]
            frames = np.concatenate([load_train_frames_u8(F.data_dir, v, [labels[v]["ivt"][i, 0]], F.img_size, F.img_size, aug_rng,
                                                          F.augmentation_list) for v, i in batch])
            masks = tr.draw_masks_device(len(batch), F.seed * 1000003 + rank, step_no)
            if single:
                lab = torch.from_numpy(np.stack([labels[v][F.loss_type][i, 1:] for v, i in batch]))
                tot += tr.train_step(torch.from_numpy(frames).cuda(), lab, masks)
            else:
                lab = [torch.from_numpy(np.stack([labels[v][k][i, 1:] for v, i in batch])) for k in ("i", "v", "t", "ivt")]
                key = lambda v: featfile.video_key(v)
                tp = [torch.from_numpy(np.stack([tpred[t][key(v)][i] for v, i in batch]).astype(np.float32)) for t in "ivt"]
                tf = [torch.from_numpy(np.stack([tfeat[t][key(v)][i] for v, i in batch]).astype(np.float32)) for t in "ivt"]
                tot += tr.train_step(torch.from_numpy(frames).cuda(), lab, masks, teacher_pred=tp, teacher_feat=tf)["loss"]
            step_no += 1
        last = {"loss": tot / steps, "lr": tr.lr}
        if rank == 0:
            _log(logfile, f"Traning | lr: {tr.lr:.6f} | epoch {epoch} | loss {tot / steps:.4f} | {time.time() - t0:.2f} secs")
        if epoch % val_interval == 0 and rank == 0:          # `weight_mgt`: latest every validation, best by the task's mAP (:416-421)
            state = tr.state_dict()
            torch.save(state, latest)
            model = build_q2l(eval_args, dtype=torch.float32).eval()
            model.load_state_dict(state)
            vt = F.loss_type if single else "ivt"                  # the head the validation mAP is taken on (`run.py:443-450`)
            m = Recognition({"i": 6, "v": 10, "t": 15, "ivt": 100}[vt])
            gi = ("i", "v", "t", "ivt").index(vt)
            for v in val_videos:
                lv = labels[v][vt]
                vb = max(F.batch, min(getattr(F, "device_batch", F.batch), 128))     # validation passes in device batches (results do not depend on it)
                for s0 in range(0, len(lv), vb):
                    fr = cholect.load_frames_device(F.data_dir, v, lv[s0:s0 + vb, 0], F.img_size, F.img_size,
                                                    workers=getattr(F, "decode_workers", 0), decode=getattr(F, "png_decode", "host"))
                    zt = [] if single else [torch.zeros((fr.shape[0], F.teacher_dim), device=fr.device)] * 3   # (`dataloader.py:240-246`: zeros off the train split)
                    m.update(lv[s0:s0 + vb, 1:], _sigmoid(model(fr, *zt)[gi][1]))
                m.video_end()
            score = float(m.compute_video_AP(ignore_null=_chlg(F))["mAP"]) if val_videos else 0.0
            last["val_mAP"] = score
            if score > best or not os.path.exists(ckpt):
                best = max(best, score)
                torch.save(state, ckpt)
                _log(logfile, f">>> Saving checkpoint for epoch {epoch + 1} at {ckpt}, time {time.ctime()} ")
            _log(logfile, f"\t\t\t\t\t\t\t video-wise | eta {time.time() - t0:.2f} secs | mAP => {vt}: [{score:.5f}] ")
        _barrier()
    return last


def _wants_test(argv) -> bool:
    argv = list(sys.argv[1:] if argv is None else argv)
    return "-e" in argv or "--test" in argv


def spatial_cnn_run(argv=None):
    """`Spatial_cnn/run.py`: -t trains the student (`spatial_cnn_train`), -e evaluates the test split and writes the closing report
    (`spatial_cnn_eval`, `run.py:503-560`); the extraction pass over all videos is `test.py` (`spatial_cnn_test`)."""
    last = spatial_cnn_train(argv) if _wants_train(argv) else None
    return spatial_cnn_eval(argv) if _wants_test(argv) else last


def spatial_transformer_run(argv=None):
    """`Spatial_transformer/run.py`: -t trains the teacher (`spatial_transformer_train`), -e evaluates the test split and writes the closing
    report (`spatial_transformer_eval`, `run.py:482-527`); the extraction pass over all videos is `test.py` (`spatial_transformer_test`)."""
    last = spatial_transformer_train(argv) if _wants_train(argv) else None
    return spatial_transformer_eval(argv) if _wants_test(argv) else last


def mstct_run(argv=None):
    """`Temporal_mstct/run.py`: -t trains the MS-TCT teacher on random 256-frame windows (`run.py:147-235`), -e evaluates the test split and
    writes the closing report + `mAPs.pckl` (`mstct_eval`, `run.py:527-580`); features / raw predictions for the student come from `test.py`
    (`mstct_test`)."""
    last = None
    if _wants_train(argv):
        from . import mstct_train
        mstct_train.train_driver(sys.argv[1:] if argv is None else argv)
    return mstct_eval(argv) if _wants_test(argv) else last
