"""Video-wise average precision, restating what the drivers use from the un-vendored `ivtmetrics 0.0.6`
(`Spatial_cnn/run.py:331-338,426-451`): per video, per class AP (sklearn `average_precision_score`), NaN for classes
without positives; `compute_video_AP` = nan-mean over videos per class, then nan-mean over classes.
Component disentangling (`compute_video_AP('i'|'v'|'t'|'iv'|'it')` on the 100-way triplet scores, `run.py:438-444`): a component
class takes the MAX over the triplets that contain it, for scores and labels alike, through the CholecT45/T50 triplet dictionary
(the table of `Spatial_cnn/maps.txt`, which no reference code reads -- ivtmetrics bundles the same one).
PARITY UNPINNED: ivtmetrics is not installed and the reference holds no fixture for it; pinned here against sklearn
and hand-built cases only.  `ignore_null=True` (the CholecTriplet challenge protocol, off for every shipped recipe): restated from ivtmetrics'
published behaviour -- the six null triplets (instrument, null_verb, null_target), ids 94-99, are dropped from the 100-way scores before the
AP; other components ignore the flag -- equally unpinned."""
from __future__ import annotations

import warnings
from typing import List

import numpy as np


# CholecT45/T50 triplet dictionary: triplet id -> (instrument, verb, target); iv = 10*i + v, it = 15*i + t (dataset label data)
_TRIPLETS = (
    (0,2,1), (0,2,0), (0,2,10), (0,0,3), (0,0,2), (0,0,4), (0,0,1), (0,0,0), (0,0,12), (0,0,8),
    (0,0,10), (0,0,11), (0,0,13), (0,8,0), (0,1,2), (0,1,4), (0,1,1), (0,1,0), (0,1,12), (0,1,8),
    (0,1,10), (0,1,11), (1,3,7), (1,3,5), (1,3,3), (1,3,2), (1,3,4), (1,3,1), (1,3,0), (1,3,8),
    (1,3,10), (1,3,11), (1,2,9), (1,2,3), (1,2,2), (1,2,1), (1,2,0), (1,2,10), (1,0,1), (1,0,8),
    (1,0,13), (1,1,2), (1,1,4), (1,1,0), (1,1,8), (1,1,10), (2,3,5), (2,3,3), (2,3,2), (2,3,4),
    (2,3,1), (2,3,0), (2,3,8), (2,3,10), (2,5,5), (2,5,11), (2,2,5), (2,2,3), (2,2,2), (2,2,1),
    (2,2,0), (2,2,10), (2,2,11), (2,1,0), (2,1,8), (3,3,10), (3,5,9), (3,5,5), (3,5,3), (3,5,2),
    (3,5,1), (3,5,8), (3,5,10), (3,5,11), (3,2,1), (3,2,0), (3,2,10), (4,4,5), (4,4,3), (4,4,2),
    (4,4,4), (4,4,1), (5,6,6), (5,2,2), (5,2,4), (5,2,1), (5,2,0), (5,2,10), (5,7,7), (5,7,4),
    (5,7,8), (5,1,0), (5,1,8), (5,1,10), (0,9,14), (1,9,14), (2,9,14), (3,9,14), (4,9,14), (5,9,14),
)


N_NULL_TRIPLETS = sum(1 for (_, v, t) in _TRIPLETS if v == 9 and t == 14)      # the last 6 ids: (instrument, null_verb, null_target)
assert all(v == 9 and t == 14 for (_, v, t) in _TRIPLETS[-N_NULL_TRIPLETS:]) and N_NULL_TRIPLETS == 6


def _component_index(component: str) -> np.ndarray:
    t = np.array(_TRIPLETS, dtype=np.int64)
    return {"i": t[:, 0], "v": t[:, 1], "t": t[:, 2], "iv": 10 * t[:, 0] + t[:, 1], "it": 15 * t[:, 0] + t[:, 2]}[component]


def disentangle(x: np.ndarray, component: str) -> np.ndarray:
    """[N,100] triplet scores or labels -> [N,K] of a component (K = 6, 10, 15, 26 iv pairs, 59 it pairs; classes in ascending id
    order): max over the triplets that share the component value"""
    if component == "ivt":
        return x
    idx = _component_index(component)
    return np.stack([x[:, idx == c].max(axis=1) for c in np.unique(idx)], axis=1)


class Recognition:
    def __init__(self, num_class: int = 100):
        self.num_class = num_class
        self.reset_global()

    def reset(self):
        self.targets: List[np.ndarray] = []
        self.predictions: List[np.ndarray] = []

    def reset_global(self):
        self.global_targets: List[np.ndarray] = []
        self.global_predictions: List[np.ndarray] = []
        self.reset()

    def update(self, targets, predictions):
        self.targets.append(np.asarray(targets, dtype=np.float64).reshape(-1, self.num_class))
        self.predictions.append(np.asarray(predictions, dtype=np.float64).reshape(-1, self.num_class))

    def video_end(self):
        if self.targets:
            self.global_targets.append(np.concatenate(self.targets, 0))
            self.global_predictions.append(np.concatenate(self.predictions, 0))
        self.reset()

    @staticmethod
    def _ap_per_class(t: np.ndarray, p: np.ndarray) -> np.ndarray:
        from sklearn.metrics import average_precision_score
        out = np.full(t.shape[1], np.nan)
        for c in range(t.shape[1]):
            if t[:, c].sum() > 0:
                out[c] = average_precision_score(t[:, c], p[:, c])
        return out

    def compute_video_AP(self, component: str = "ivt", ignore_null: bool = False):
        if component != "ivt" and self.num_class != 100:
            raise ValueError("component disentangling needs the 100-way triplet scores")
        drop = N_NULL_TRIPLETS if (ignore_null and component == "ivt" and self.num_class == 100) else 0
        cut = (lambda a: a[:, :a.shape[1] - drop]) if drop else (lambda a: a)
        per_video = [self._ap_per_class(cut(disentangle(t, component)), cut(disentangle(p, component)))
                     for t, p in zip(self.global_targets, self.global_predictions)]
        with warnings.catch_warnings():
            warnings.simplefilter("ignore", category=RuntimeWarning)
            ap = np.nanmean(np.stack(per_video, 0), axis=0) if per_video else np.full(self.num_class, np.nan)
            return {"AP": ap, "mAP": float(np.nanmean(ap)) if np.isfinite(ap).any() else float("nan")}

    def topK(self, k: int = 5, component: str = "ivt") -> float:
        """`mAP.topK(k, component)` (`Spatial_cnn/run.py:543-548`) as the reference spells it out itself in `Temporal_mstct/run.py:507-523`
        (`topk`): over ALL frames seen, (number of positive classes that are among the frame's k highest scores) / (number of positive
        classes); 0 / 1 when nothing is positive.  `component` disentangles 100-way triplet scores and labels first (`disentangle`)."""
        if component != "ivt" and self.num_class != 100:
            raise ValueError("component disentangling needs the 100-way triplet scores")
        correct, total = 0, 0
        for t, p in zip(self.global_targets, self.global_predictions):
            t, p = disentangle(t, component), disentangle(p, component)
            top = np.argsort(-p, axis=1, kind="stable")[:, :k]
            hit = np.take_along_axis(t != 0, top, axis=1)
            correct += int(hit.sum())
            total += int((t != 0).sum())
        return correct / (total if total else 1)

    # ---- the per-video record, for merging the videos of several ranks (host side; no reference counterpart: its drivers are single-process)
    def videos(self):
        return list(zip(self.global_targets, self.global_predictions))

    def set_videos(self, vids):
        self.reset_global()
        for t, p in vids:
            self.global_targets.append(np.asarray(t, dtype=np.float64).reshape(-1, self.num_class))
            self.global_predictions.append(np.asarray(p, dtype=np.float64).reshape(-1, self.num_class))
        return self


HEADS = (("i", 6), ("v", 10), ("t", 15), ("ivt", 100))


def gather_recognition(local, order, group=None):
    """local: {video key -> {head -> (targets [N,K], predictions [N,K])}} of THIS rank's videos.  Returns {head -> Recognition} holding the
    videos of ALL ranks in `order` (the single-process order), identical on every rank: an N-rank evaluation reports the 1-rank numbers.
    One host-side object gather, like `extract.gather_feats` (videos are independent units: no data-path collective)."""
    import torch.distributed as dist
    parts = [dict(local)]
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        parts = [None] * dist.get_world_size(group)
        dist.all_gather_object(parts, dict(local), group=group)
    merged = {}
    for part in parts:
        for k, v in part.items():
            if k in merged:
                raise ValueError(f"video {k} evaluated by two ranks")
            merged[k] = v
    return recognition_from(merged, order)


def recognition_from(scores, order):
    """{video -> {head -> (targets, predictions)}} -> {head -> Recognition} with the videos in `order` (no communication)"""
    missing = [k for k in order if k not in scores]
    if missing:
        raise KeyError(f"videos without predictions: {missing[:4]}")
    heads = [h for h, _ in HEADS if all(h in scores[k] for k in order)]
    return {h: Recognition(dict(HEADS)[h]).set_videos([scores[k][h] for k in order]) for h in heads}


def final_report(m, loss_type: str = "all", ignore_null: bool = False, style: str = "spatial_cnn"):
    """The lines of the reference's closing evaluation report, in order (the drivers write them to the log file):
    `style` 'spatial_cnn'          `Spatial_cnn/run.py:517-560` -- I / V / T from the component heads when `loss_type` is i | v | t, else DISENTANGLED from
                                   the 100-way triplet head; IV / IT / IVT from the triplet head; per-category vectors, mean-AP row, top-5 / 10 / 20 rows
          'spatial_transformer'    `Spatial_transformer/run.py:500-527` -- the same without top-K numbers (it prints the 'top 5' header only)
          'temporal_tenco'         `Temporal_tenco/run.py:534-570` -- head-wise ('singletest') AND disentangled vectors, both mean-AP rows
          'temporal_mstct'         `Temporal_mstct/run.py:550-580` -- head-wise vectors, both mean-AP rows
    m: {head -> Recognition} ('ivt' + the component heads).  Returns (lines, {name -> mAP}) -- the numbers of the README table (`readme.md:111-113`)."""
    ap = lambda head, comp=None: (m[head].compute_video_AP(ignore_null=ignore_null) if comp is None
                                  else m[head].compute_video_AP(comp, ignore_null=ignore_null))
    single = {c: ap(c) for c in ("i", "v", "t")}                       # the component heads' own AP
    dis = {c: ap("ivt", c) for c in ("i", "v", "t")}                   # disentangled from the triplet head
    iv, it, ivt = ap("ivt", "iv"), ap("ivt", "it"), ap("ivt", "ivt")
    row = lambda a: f':::::: : {a[0]:.4f} | {a[1]:.4f} | {a[2]:.4f} | {a[3]:.4f} | {a[4]:.4f} | {a[5]:.4f} '
    head_row = lambda name: f'{name}:  I  |  V  |  T  |  IV  |  IT  |  IVT '
    L = ['-' * 50, 'Test Results\nPer-category AP: ']
    if style in ("spatial_cnn", "spatial_transformer"):
        comp = single if loss_type in ("i", "v", "t") else dis
        L += [f'I   : {comp["i"]["AP"]}', f'V   : {comp["v"]["AP"]}', f'T   : {comp["t"]["AP"]}', f'IV  : {iv["AP"]}', f'IT  : {it["AP"]}',
              f'IVT : {ivt["AP"]}', '-' * 50, head_row('Mean AP'),
              row([comp["i"]["mAP"], comp["v"]["mAP"], comp["t"]["mAP"], iv["mAP"], it["mAP"], ivt["mAP"]])]
        res = {"AP_i": comp["i"]["mAP"], "AP_v": comp["v"]["mAP"], "AP_t": comp["t"]["mAP"]}
        if style == "spatial_cnn":
            for k in (5, 10, 20):
                L += [head_row(f'top {k}'), row([m["ivt"].topK(k, c) for c in ("i", "v", "t", "iv", "it", "ivt")])]
                res.update({f"top{k}_{c}": m["ivt"].topK(k, c) for c in ("i", "v", "t", "iv", "it", "ivt")})
        else:
            L += [head_row('top 5')]
    else:
        if style == "temporal_tenco":
            L += ['------------singletest-------------', f'I   : {single["i"]["AP"]}', f'V   : {single["v"]["AP"]}', f'T   : {single["t"]["AP"]}',
                  '-' * 50, 'Test Results\nPer-category AP: ', f'I   : {dis["i"]["AP"]}', f'V   : {dis["v"]["AP"]}', f'T   : {dis["t"]["AP"]}']
        else:
            L += [f'I   : {single["i"]["AP"]}', f'V   : {single["v"]["AP"]}', f'T   : {single["t"]["AP"]}']
        L += [f'IV  : {iv["AP"]}', f'IT  : {it["AP"]}', f'IVT : {ivt["AP"]}', '-' * 50, head_row('Mean AP'),
              row([dis["i"]["mAP"], dis["v"]["mAP"], dis["t"]["mAP"], iv["mAP"], it["mAP"], ivt["mAP"]]),
              '------------singletest-------------', head_row('Mean AP'),
              row([single["i"]["mAP"], single["v"]["mAP"], single["t"]["mAP"], iv["mAP"], it["mAP"], ivt["mAP"]])]
        res = {"AP_i": dis["i"]["mAP"], "AP_v": dis["v"]["mAP"], "AP_t": dis["t"]["mAP"],
               "AP_i_single": single["i"]["mAP"], "AP_v_single": single["v"]["mAP"], "AP_t_single": single["t"]["mAP"]}
    L.append('=' * 50)
    res.update({"AP_iv": iv["mAP"], "AP_it": it["mAP"], "AP_ivt": ivt["mAP"]})
    return L, res
