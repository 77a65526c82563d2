"""Oracle for one Temporal_mstct training step (`Temporal_mstct/run.py:147-235`: loop over a batch of 256-frame windows, BCE-with-logits
per window averaged over the batch, `loss.backward()`, `torch.optim.SGD.step()`; optimizer `run.py:345`), by torch autograd on the
functional oracle (`oracle/mstct.py`).  Test infrastructure only.

Randomness is explicit so that both sides of a parity test consume the same draw:
  masks["input"]  [B,D,T]  nn.Dropout(0.5) on the raw features (`network.py:58,76`): 0 or 2
  masks["feat"]   [B,E,T]  nn.Dropout(0.5) between linear_fuse and linear_pred (`network.py:108,113`): 0 or 2
None = that piece is the identity (what the fixtures captured from the reference use: its Dropout modules set to p = 0)."""
from __future__ import annotations

from typing import Dict, Optional

import torch
import torch.nn.functional as F

from . import mstct as o_m
from .spatial_cnn_train import TARGET_W, TOOL_W, VERB_W

SD = Dict[str, torch.Tensor]
NCLS = {"i": 6, "v": 10, "t": 15, "ivt": 100}
POS_W = {"i": TOOL_W, "v": VERB_W, "t": TARGET_W, "ivt": None}     # `run.py:331-334`: pos_weight for i / v / t, none for ivt


def forward_train(sd: SD, x: torch.Tensor, loss_type: str, masks: Optional[dict] = None):
    """x [B,D,T] -> logits [B,T,K] of the one classifier that exists (`network.py:66-73`)"""
    masks = masks or {}
    if masks.get("input") is not None:
        x = x * masks["input"]
    feats = o_m.temporal_encoder(sd, "TemporalEncoder.", x)
    concat = o_m.temporal_mixer(sd, "Temporal_Mixer.", feats)
    q = f"classifier_{loss_type}"
    feat = F.conv1d(concat, sd[q + ".linear_fuse.weight"], sd[q + ".linear_fuse.bias"])
    if masks.get("feat") is not None:
        feat = feat * masks["feat"]
    return F.conv1d(feat, sd[q + ".linear_pred.weight"], sd[q + ".linear_pred.bias"]).permute(0, 2, 1)


def loss_fn(logits: torch.Tensor, labels: torch.Tensor, loss_type: str):
    """mean over the batch of the per-window BCEWithLogitsLoss (`run.py:159-187`); the heads that do not exist output zeros and add a
    constant for loss_type 'ivt' (`run.py:192`) which has no gradient and is left out of the reported term"""
    pw = torch.tensor(POS_W[loss_type], dtype=logits.dtype) if POS_W[loss_type] is not None else None
    per = [F.binary_cross_entropy_with_logits(logits[i], labels[i].to(logits.dtype), pos_weight=pw) for i in range(logits.shape[0])]
    return sum(per) / len(per)


def train_step(sd: SD, x: torch.Tensor, labels: torch.Tensor, loss_type: str, lr: float, weight_decay: float = 1e-5, masks: Optional[dict] = None):
    """one SGD step without momentum.  x [B,D,T], labels [B,T,K] multi-hot.  Returns (new_sd, loss, grads)."""
    params = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    loss = loss_fn(forward_train(params, x, loss_type, masks), labels, loss_type)
    names = list(params)
    grads = torch.autograd.grad(loss, [params[k] for k in names], allow_unused=True)
    g = {k: (gr if gr is not None else torch.zeros_like(params[k])) for k, gr in zip(names, grads)}
    used = {k for k, gr in zip(names, grads) if gr is not None}
    new = {k: (params[k].detach() - lr * (g[k] + weight_decay * params[k].detach())) if k in used else params[k].detach().clone() for k in names}
    return new, float(loss), g


def train_step_f64(sd: SD, x, labels, loss_type, lr, weight_decay=1e-5, masks=None):
    """the same step in float64 (how far fp32 arithmetic -- torch's or the HIP path's -- is from exact)"""
    sd64 = {k: v.double() for k, v in sd.items()}
    m64 = {k: (v.double() if v is not None else None) for k, v in (masks or {}).items()} or None
    return train_step(sd64, x.double(), labels, loss_type, lr, weight_decay, m64)
