"""Checkpoint interop: the loading conventions of the reference, so that its released `.pth` files and the upstream
Swin pre-training files go into the MI355X modules unchanged.

* `clean_state_dict` -- strip DataParallel's `module.` (`Spatial_transformer/utils/misc.py:392-398`).
* `load_partial` -- `load_model` of every driver (`Spatial_cnn/run.py:272-278`, `Spatial_transformer/run.py:281-287`, ...): keep the keys
  the model has, `strict=False`.
* `swin_pretrain_to_q2l` -- `build_backbone`'s Swin branch (`models/backbone.py:188-201`): an upstream Swin checkpoint
  (`{'model': state_dict}`, e.g. `swin_base_patch4_window12_384_22k.pth`) without its classification `head`, under the `backbone.0.`
  prefix the `Joiner` gives it inside `Qeruy2Label`.
"""
from __future__ import annotations

from collections import OrderedDict
from typing import Dict, Mapping

import torch


def clean_state_dict(state_dict: Mapping[str, torch.Tensor]) -> "OrderedDict[str, torch.Tensor]":
    return OrderedDict((k[7:] if k[:7] == "module." else k, v) for k, v in state_dict.items())


def unwrap(obj) -> Mapping[str, torch.Tensor]:
    """a `.pth` as saved by the reference (`torch.save(model.state_dict())`) or by upstream trainers (`{'model': ...}` /
    `{'state_dict': ...}`)"""
    if isinstance(obj, Mapping):
        for key in ("model", "state_dict"):
            if key in obj and isinstance(obj[key], Mapping):
                return obj[key]
    return obj


def load_partial(model, source, strict_shapes: bool = True) -> Dict[str, list]:
    """reference `load_model`: copy the entries of `source` (path or mapping) whose key the model has; returns what was used,
    ignored and left at its current value"""
    sd = clean_state_dict(unwrap(torch.load(source, map_location="cpu") if isinstance(source, (str, bytes)) else source))
    cur = dict(model.state_dict())
    used, ignored = [], []
    for k, v in sd.items():
        if k in cur and (not strict_shapes or tuple(v.shape) == tuple(cur[k].shape)):
            cur[k] = v
            used.append(k)
        else:
            ignored.append(k)
    model.load_state_dict(cur, strict=False)
    return {"used": used, "ignored": ignored, "kept": [k for k in cur if k not in sd]}


def swin_pretrain_to_q2l(checkpoint, prefix: str = "backbone.0.") -> "OrderedDict[str, torch.Tensor]":
    """upstream Swin checkpoint -> the `backbone.0.*` entries of a `Qeruy2Label` state dict (classification head dropped)"""
    sd = clean_state_dict(unwrap(checkpoint))
    return OrderedDict((prefix + k, v) for k, v in sd.items() if "head" not in k)
