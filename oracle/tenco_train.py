"""Oracle for one Temporal_tenco training step (`Temporal_tenco/run.py:181-235`, optimizer `:341-348`), by torch autograd
on the functional oracle.  Test infrastructure only.

Randomness is explicit: `masks` carries the train-time random pieces of `network.py` as tensors so that both sides of a
parity test consume the same draw --
  input_mask   [1,D,T]   the 75 % random element mask (`:43-48`; 0/1, multiplies the input)
  channel_mask [1,D,1]   Dropout2d on the input (`:123-127`; 0 or 2 per channel)
  layer_masks  {prefix: [1,C,T]}  Dropout(0.5) after each layer's 1x1 conv (`:194-196`; 0 or 2)
None = that piece is the identity (what `model.eval()` does)."""
from __future__ import annotations

from typing import Dict, Optional

import torch
import torch.nn.functional as F

SD = Dict[str, torch.Tensor]
HEADS = (("", 100, 1.0), ("_i", 6, 0.1), ("_v", 10, 0.1), ("_t", 15, 0.1))   # loss = 0.1 (i + v + t) + ivt  (`run.py:212`)


def forward_train(sd: SD, x: torch.Tensor, num_layers_PG=11, num_layers_R=10, num_R=3, masks: Optional[dict] = None, hier: bool = False):
    """hier = `args.hier` (`network.py:147,154-155`): every refinement stage ends in AvgPool1d(7, 3), level l has its own length and the FPN's
    `F.interpolate(x, size=W, mode='linear')` (:96) resamples; per-layer dropout masks then have their stage's length."""
    masks = masks or {}
    h = x.permute(0, 2, 1)
    if masks.get("input_mask") is not None:
        h = h * masks["input_mask"]
    if masks.get("channel_mask") is not None:
        h = h * masks["channel_mask"]
    lm = masks.get("layer_masks") or {}

    def stage(prefix, z, n, project):
        if project:
            z = F.conv1d(z, sd[prefix + ".conv_1x1.weight"], sd[prefix + ".conv_1x1.bias"])
        for i in range(n):
            p = f"{prefix}.layers.{i}"
            d = 2 ** i
            u = F.relu(F.conv1d(z, sd[p + ".conv_dilated.weight"], sd[p + ".conv_dilated.bias"], padding=d, dilation=d))
            o = F.conv1d(u, sd[p + ".conv_1x1.weight"], sd[p + ".conv_1x1.bias"])
            if p in lm:
                o = o * lm[p]
            z = z + o
        return z

    f = stage("PG", h, num_layers_PG, True)
    fl = [f]
    for r in range(num_R):
        f = stage(f"Rs.{r}", f, num_layers_R, False)
        if hier:
            f = F.avg_pool1d(f, kernel_size=7, stride=3)
        fl.append(f)
    w, b = sd["fpn.latlayer1.weight"], sd["fpn.latlayer1.bias"]
    c1, c2, c3, p4 = fl
    up = (lambda xx, yy: F.interpolate(xx, size=yy.shape[-1], mode="linear")) if hier else (lambda xx, yy: xx)
    l3 = F.conv1d(c3, w, b)
    p3 = up(p4, l3) + l3
    l2 = F.conv1d(c2, w, b)
    p2 = up(p3, l2) + l2
    l1 = F.conv1d(c1, w, b)
    p1 = up(p2, l1) + l1
    levels = [p1, p2, p3, p4]
    return {s: [F.conv1d(l, sd[f"conv_out{s}.weight"], sd[f"conv_out{s}.bias"]) for l in levels] for s, _, _ in HEADS}


def resize_labels(y: torch.Tensor, t: int) -> torch.Tensor:
    """`fusion` (`run.py:159-179`): the [T,K] labels resized to a level's length with `F.interpolate(mode='nearest')` (identity at equal length)"""
    if y.shape[0] == t:
        return y
    return F.interpolate(y.float().transpose(0, 1).unsqueeze(0), size=t, mode="nearest").squeeze(0).transpose(0, 1).long()


def loss_terms(logits: dict, labels: dict):
    """labels: {'': [T,100], '_i': [T,6], ...} multi-hot.  BCEWithLogits (mean) summed over the 4 FPN levels, un-weighted
    per head (`run.py:196-210`: every head uses loss_fn_ivt); a level of another length (--hier) is scored against the labels resized to it
    (`fusion`, `run.py:159-179`)."""
    out = {}
    for s, _, _ in HEADS:
        out[s] = sum(F.binary_cross_entropy_with_logits(l[0].transpose(0, 1), resize_labels(labels[s], l.shape[-1]).float()) for l in logits[s])
    total = sum(wgt * out[s] for s, _, wgt in HEADS)
    return total, out


def train_step(sd: SD, x: torch.Tensor, labels: dict, lr: float, weight_decay: float = 1e-5, masks: Optional[dict] = None, **cfg):
    """one SGD step (no momentum; `run.py:343`).  Returns (new_sd, loss, loss_terms, grads)."""
    params = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    total, terms = loss_terms(forward_train(params, x, masks=masks, **cfg), labels)
    names = list(params)
    grads = torch.autograd.grad(total, [params[k] for k in names], allow_unused=True)
    g = {k: (gr if gr is not None else torch.zeros_like(params[k])) for k, gr in zip(names, grads)}
    used = {k for k, gr in zip(names, grads) if gr is not None}
    # torch.optim.SGD only touches parameters that received a gradient (param.grad None -> skipped)
    new = {k: (params[k].detach() - lr * (g[k] + weight_decay * params[k].detach())) if k in used else params[k].detach().clone() for k in names}
    return new, float(total), {s: float(v) for s, v in terms.items()}, g
