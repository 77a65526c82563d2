"""Frame-feature / teacher-prediction files: the on-disk seam between the stages.

Reference writers: `Spatial_cnn/test.py:266-284`, `Spatial_transformer/test.py:357-376`,
`Temporal_mstct/test.py:338-366`; readers: `Temporal_tenco/dataloader.py:212-214`,
`Temporal_mstct/dataloader.py:220-222`, `Spatial_cnn/dataloader.py:227-238`.

Format: ONE pickle per (run, fold[, task]) holding ``dict{video-key -> float32 ndarray [N_frames, D]}``;
video-key = last two characters of the video directory ('79' for VID79; `test.py:268`).  Path:
``../0-5fold/data_feats/run_<version>/k<fold>_feats.pkl`` for loss_type 'all', else
``k<fold>_<task>_feats.pkl``; predictions ``k<fold>_<task>_pred.pkl`` hold raw logits [N, K].
BASELINE.json also names a per-video ``.npy`` layout: `write_npy_dir` emits it next to the pickle (the
payload is the same C-contiguous float32 array).
"""
from __future__ import annotations

import os
import pickle
from typing import Dict, Mapping

import numpy as np


def video_key(video_dir: str, style: str = "cnn") -> str:
    """'cnn'/'mstct': last two chars (`Spatial_cnn/test.py:268`); 'transformer': name[3:] (`Spatial_transformer/test.py:357`)."""
    name = os.path.basename(os.path.normpath(video_dir))
    return name[3:] if style == "transformer" else name[-2:]


def feats_path(root: str, version: str, kfold, loss_type: str = "all", kind: str = "feats") -> str:
    d = os.path.join(root, "0-5fold", "data_feats", f"run_{version}")
    if loss_type == "all" and kind == "feats":
        return os.path.join(d, f"k{kfold}_feats.pkl")
    return os.path.join(d, f"k{kfold}_{loss_type}_{kind}.pkl")


def _canon(a) -> np.ndarray:
    a = np.ascontiguousarray(np.asarray(a), dtype=np.float32)
    if a.ndim != 2:
        raise ValueError(f"feature array must be [N_frames, D], got {a.shape}")
    return a


def write_feats(path: str, feats: Mapping[str, np.ndarray]) -> None:
    os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
    payload = {str(k): _canon(v) for k, v in feats.items()}
    tmp = f"{path}.{os.getpid()}.tmp"          # per-process name: two writers of one path never share a half-written file
    try:
        with open(tmp, "wb") as f:
            pickle.dump(payload, f)
        os.replace(tmp, path)
    finally:
        if os.path.exists(tmp):
            os.remove(tmp)


def read_feats(path: str) -> Dict[str, np.ndarray]:
    with open(path, "rb") as f:
        d = pickle.load(f)
    if not isinstance(d, dict):
        raise ValueError("feature file must hold a dict")
    return {str(k): _canon(v) for k, v in d.items()}


def write_npy_dir(dirpath: str, feats: Mapping[str, np.ndarray]) -> None:
    os.makedirs(dirpath, exist_ok=True)
    for k, v in feats.items():
        np.save(os.path.join(dirpath, f"VID{k}.npy"), _canon(v))
