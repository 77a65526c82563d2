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

    def topK(self, k: int = 5) -> float:
        """fraction of frames whose positive classes intersect the top-k scores (`run.py:543-548`)"""
        hits, n = 0, 0
        for t, p in zip(self.global_targets, self.global_predictions):
            top = np.argsort(-p, axis=1)[:, :k]
            for i in range(t.shape[0]):
                pos = np.nonzero(t[i])[0]
                if len(pos):
                    n += 1
                    hits += int(len(np.intersect1d(pos, top[i])) > 0)
        return hits / n if n else float("nan")
