"""Video-wise average precision, restating what the drivers use from the un-vendored `ivtmetrics 0.0.6`
(`Spatial_cnn/run.py:331-338,426-451`): per video, per class AP (sklearn `average_precision_score`), NaN for classes
without positives; `compute_video_AP` = nan-mean over videos per class, then nan-mean over classes.
PARITY UNPINNED: ivtmetrics is not installed and the reference holds no fixture for it; the component
disentangling (i/v/t from ivt via maps.txt) is not restated -- the drivers feed each head's own logits."""
from __future__ import annotations

import warnings
from typing import List

import numpy as np


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

    def compute_video_AP(self, component: str = "ivt"):
        per_video = [self._ap_per_class(t, p) for t, p in zip(self.global_targets, self.global_predictions)]
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
