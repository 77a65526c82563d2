"""Oracle for one training step of the Swin + Query2Label teacher (`Spatial_transformer/run.py:150-229`, single-task recipe of
`Scripts/train_fold1.sh:12`: loss = BCEWithLogitsLoss(pos_weight) of the one decoder that exists; optimizer `run.py:360`), by torch autograd
on the functional oracle (`oracle/swin_q2l.py`).  Test infrastructure only.

Randomness is explicit so that both sides of a parity test consume the same draw (None = identity, what the fixtures captured from the reference
use: its DropPath / Dropout modules set to p = 0):
  masks["droppath"]  list over all Swin blocks of (attn [B], mlp [B]): timm DropPath keep mask / keep_prob per sample (`swin_transformer.py:266,269`)
  masks["tx"]        dict of nn.Dropout(0.1) masks of the Q2L transformer (`transformer.py:160-196,260-302`), in ROW layout (rows b-major):
                     "enc.attn" [B,4,L,L]  "enc.d1" [B*L,d]  "enc.ffn" [B*L,F]  "enc.d2" [B*L,d]
                     "dec{l}.attn" [B,4,K,L]  "dec{l}.d2" [B*K,d]  "dec{l}.ffn" [B*K,F]  "dec{l}.d3" [B*K,d]"""
from __future__ import annotations

from typing import Dict, Optional

import torch
import torch.nn.functional as F

from . import swin_q2l as o
from .spatial_cnn_train import TARGET_W, TOOL_W, VERB_W

SD = Dict[str, torch.Tensor]
POS_W = {"i": TOOL_W, "v": VERB_W, "t": TARGET_W}


def _swin_block(sd, p, x, res, nh, ws, shift, dp):
    h, w = res
    if min(res) <= ws:
        shift, ws = 0, min(res)
    b, l, c = x.shape
    shortcut = x
    x = o._ln(sd, p + ".norm1", x).view(b, h, w, c)
    if shift > 0:
        x = torch.roll(x, shifts=(-shift, -shift), dims=(1, 2))
    xw = o.window_partition(x, ws).view(-1, ws * ws, c)
    mask = o.shift_attn_mask(h, w, ws, shift) if shift > 0 else None
    aw = o.window_attention(sd, p + ".attn", xw, nh, ws, mask).view(-1, ws, ws, c)
    x = o.window_reverse(aw, ws, h, w)
    if shift > 0:
        x = torch.roll(x, shifts=(shift, shift), dims=(1, 2))
    x = x.view(b, h * w, c)
    if dp is not None:
        x = x * dp[0].view(b, 1, 1)
    x = shortcut + x
    y = o._ln(sd, p + ".norm2", x)
    y = F.linear(F.gelu(F.linear(y, sd[p + ".mlp.fc1.weight"], sd[p + ".mlp.fc1.bias"])), sd[p + ".mlp.fc2.weight"], sd[p + ".mlp.fc2.bias"])
    if dp is not None:
        y = y * dp[1].view(b, 1, 1)
    return x + y


def swin_features(sd: SD, x, name, img_size, prefix, droppath=None):
    cfg = o.SWIN_CFG[name]
    x = F.conv2d(x, sd[prefix + "patch_embed.proj.weight"], sd[prefix + "patch_embed.proj.bias"], stride=4)
    x = x.flatten(2).transpose(1, 2)
    x = o._ln(sd, prefix + "patch_embed.norm", x)
    r = img_size // 4
    bi_all = 0
    for s, (depth, nh) in enumerate(zip(cfg["depths"], cfg["num_heads"])):
        res = (r // (2 ** s), r // (2 ** s))
        for bi in range(depth):
            shift = 0 if bi % 2 == 0 else cfg["window_size"] // 2
            x = _swin_block(sd, f"{prefix}layers.{s}.blocks.{bi}", x, res, nh, cfg["window_size"], shift, droppath[bi_all] if droppath else None)
            bi_all += 1
        if s < 3:
            x = o.patch_merging(sd, f"{prefix}layers.{s}.downsample", x, res)
    x = o._ln(sd, prefix + "norm", x)
    b, l, c = x.shape
    hh = img_size // 32
    return x.transpose(1, 2).reshape(b, c, hh, hh)


def _rows(m, b, l):        # row-layout mask [B*L, E] -> seq-first [L, B, E]
    return m.view(b, l, -1).transpose(0, 1)


def _mha(sd, p, q, k, v, nhead, attn_mask):
    e = q.shape[-1]
    w, bia = sd[p + ".in_proj_weight"], sd[p + ".in_proj_bias"]
    qq = F.linear(q, w[:e], bia[:e])
    kk = F.linear(k, w[e:2 * e], bia[e:2 * e])
    vv = F.linear(v, w[2 * e:], bia[2 * e:])
    lq, bs, _ = qq.shape
    lk = kk.shape[0]
    hd = e // nhead
    qq = qq.reshape(lq, bs * nhead, hd).transpose(0, 1) * (hd ** -0.5)
    kk = kk.reshape(lk, bs * nhead, hd).transpose(0, 1)
    vv = vv.reshape(lk, bs * nhead, hd).transpose(0, 1)
    pr = torch.softmax(qq @ kk.transpose(1, 2), dim=-1)
    if attn_mask is not None:
        pr = pr * attn_mask.reshape(bs * nhead, lq, lk)
    a = (pr @ vv).transpose(0, 1).reshape(lq, bs, e)
    return F.linear(a, sd[p + ".out_proj.weight"], sd[p + ".out_proj.bias"])


def decoder_logits(sd: SD, p: str, src, pos, tx=None, nhead=4, tp=None, want_feat=False, mkey=""):
    """`Decoder.forward` (`network.py:163-171`) in train mode: logits [B, K] (want_feat: and the pooled encoder memory [B, d], `:170`).
    tp: prefix of the transformer's parameters when it is shared (`loss_type all`: decoder_i.transformer., `network.py:66-73`);
    mkey: prefix of this decoder's dropout-mask keys ('ivt/' ...)"""
    tx = tx or {}
    g = lambda k: tx.get(mkey + k)
    x = F.conv2d(src, sd[p + "input_proj.weight"], sd[p + "input_proj.bias"])
    t = tp or (p + "transformer.")
    bs, c, h, w = x.shape
    L = h * w
    s = x.flatten(2).permute(2, 0, 1)
    ps = pos.flatten(2).permute(2, 0, 1)
    qe = sd[p + "query_embed.weight"].unsqueeze(1).repeat(1, bs, 1)
    K = qe.shape[0]
    e = t + "encoder.layers.0"
    qk = s + ps
    s2 = _mha(sd, e + ".self_attn", qk, qk, s, nhead, g("enc.attn"))
    if g("enc.d1") is not None:
        s2 = s2 * _rows(g("enc.d1"), bs, L)
    s = o._ln(sd, e + ".norm1", s + s2)
    f = F.relu(F.linear(s, sd[e + ".linear1.weight"], sd[e + ".linear1.bias"]))
    if g("enc.ffn") is not None:
        f = f * _rows(g("enc.ffn"), bs, L)
    s2 = F.linear(f, sd[e + ".linear2.weight"], sd[e + ".linear2.bias"])
    if g("enc.d2") is not None:
        s2 = s2 * _rows(g("enc.d2"), bs, L)
    memory = o._ln(sd, e + ".norm2", s + s2)
    tgt = torch.zeros_like(qe)
    for li in range(2):
        d = f"{t}decoder.layers.{li}"
        t2 = _mha(sd, d + ".multihead_attn", tgt + qe, memory + ps, memory, nhead, g(f"dec{li}.attn"))
        if g(f"dec{li}.d2") is not None:
            t2 = t2 * _rows(g(f"dec{li}.d2"), bs, K)
        tgt = o._ln(sd, d + ".norm2", tgt + t2)
        f = F.relu(F.linear(tgt, sd[d + ".linear1.weight"], sd[d + ".linear1.bias"]))
        if g(f"dec{li}.ffn") is not None:
            f = f * _rows(g(f"dec{li}.ffn"), bs, K)
        t2 = F.linear(f, sd[d + ".linear2.weight"], sd[d + ".linear2.bias"])
        if g(f"dec{li}.d3") is not None:
            t2 = t2 * _rows(g(f"dec{li}.d3"), bs, K)
        tgt = o._ln(sd, d + ".norm3", tgt + t2)
    hs = o._ln(sd, t + "decoder.norm", tgt).transpose(0, 1)          # [B,K,d]
    logits = (sd[p + "fc.W"] * hs).sum(-1) + sd[p + "fc.b"]
    if want_feat:
        return logits, memory.mean(dim=0)                            # AdaptiveAvgPool2d(1) over the h*w memory rows -> [B, d]
    return logits


def forward_train(sd: SD, img, backbone: str, img_size: int, hidden: int, task: str, masks: Optional[dict] = None):
    masks = masks or {}
    src = swin_features(sd, img, backbone, img_size, "backbone.0.", masks.get("droppath"))
    pos = o.sine_position_encoding(hidden, img_size // 32, img_size // 32).to(src.dtype).repeat(src.shape[0], 1, 1, 1)
    return decoder_logits(sd, f"decoder_{task}.", src, pos, masks.get("tx"))


def train_step(sd: SD, img, labels, backbone, img_size, hidden, task, lr, weight_decay=1e-5, masks=None):
    """one SGD step without momentum.  labels [B,K] multi-hot of the task.  Returns (new_sd, loss, grads)."""
    params = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    logits = forward_train(params, img, backbone, img_size, hidden, task, masks)
    loss = F.binary_cross_entropy_with_logits(logits, labels.to(logits.dtype), pos_weight=torch.tensor(POS_W[task], dtype=logits.dtype))
    names = list(params)
    grads = torch.autograd.grad(loss, [params[k] for k in names], allow_unused=True)
    g = {k: (gr if gr is not None else torch.zeros_like(params[k])) for k, gr in zip(names, grads)}
    used = {k for k, gr in zip(names, grads) if gr is not None}
    new = {k: (params[k].detach() - lr * (g[k] + weight_decay * params[k].detach())) if k in used else params[k].detach().clone() for k in names}
    return new, float(loss.detach()), g


def train_step_f64(sd, img, labels, backbone, img_size, hidden, task, lr, weight_decay=1e-5, masks=None):
    return train_step({k: v.double() for k, v in sd.items()}, img.double(), labels, backbone, img_size, hidden, task, lr, weight_decay, masks)


# ------------------------------------------------------------------------------------------------ loss_type all
TASKS_ALL = ("i", "v", "t", "ivt")


def forward_train_all(sd: SD, img, backbone: str, img_size: int, hidden: int, teacher_feat, masks: Optional[dict] = None):
    """`Qeruy2Label.forward` with `loss_type all` in train mode (`network.py:82-126`): four decoders over ONE shared transformer (parameters
    under decoder_i.transformer.*), feat = decoder_ivt's pooled memory, the KD mixing on it.  masks["tx"] keys carry the decoder's prefix
    ('i/enc.attn', ..., 'ivt/dec1.d3').  Returns (logits dict, cams (i, v, t), feat)."""
    from .spatial_cnn import kd_branch
    masks = masks or {}
    src = swin_features(sd, img, backbone, img_size, "backbone.0.", masks.get("droppath"))
    pos = o.sine_position_encoding(hidden, img_size // 32, img_size // 32).to(src.dtype).repeat(src.shape[0], 1, 1, 1)
    logits, feat = {}, None
    for task in TASKS_ALL:
        logits[task], feat = decoder_logits(sd, f"decoder_{task}.", src, pos, masks.get("tx"), tp="decoder_i.transformer.", want_feat=True,
                                            mkey=task + "/")
    cams = kd_branch(sd, feat, *teacher_feat)
    return logits, cams, feat


def loss_all(logits, cams, labels, teacher_pred, teacher_feat, rates, temp):
    """`Spatial_transformer/run.py:164-167,183-197`: hard = sum of the four BCEs (pos_weight on i / v / t, `:339-342`), soft = mean DistillKL
    against sigmoid(teacher logits), kd = mean MSE(cam, teacher feature)"""
    from .spatial_cnn_train import distill_kl
    dt = logits["i"].dtype
    hard = {t: F.binary_cross_entropy_with_logits(logits[t], y.to(dt), pos_weight=(torch.tensor(POS_W[t], dtype=dt) if t in POS_W else None))
            for t, y in zip(TASKS_ALL, labels)}
    soft = [distill_kl(logits[t], torch.sigmoid(tp.to(dt)), temp) for t, tp in zip(("i", "v", "t"), teacher_pred)]
    kdl = [F.mse_loss(c, f.to(dt)) for c, f in zip(cams, teacher_feat)]
    hard_loss = hard["i"] + hard["v"] + hard["t"] + hard["ivt"]
    soft_loss = (soft[0] + soft[1] + soft[2]) / 3
    kd_loss = (kdl[0] + kdl[1] + kdl[2]) / 3
    loss = rates[0] * hard_loss + rates[1] * soft_loss + rates[2] * kd_loss
    terms = dict(loss=float(loss.detach()), hard=float(hard_loss.detach()), soft=float(soft_loss.detach()), kd=float(kd_loss.detach()),
                 **{"hard_" + t: float(v.detach()) for t, v in hard.items()})
    return loss, terms


def train_step_all(sd: SD, img, labels, teacher_pred, teacher_feat, backbone, img_size, hidden, lr, weight_decay=1e-5, rates=(1.0, 0.0, 0.1), temp=4.0,
                   masks=None):
    """one SGD step of `run.py -t --loss_type all` (`:150-229`; the shared transformer is ONE parameter set: its gradient is the sum over the
    four decoders and it steps once, as `model.parameters()` lists it once).  labels (y_i, y_v, y_t, y_ivt) multi-hot; teacher_pred 3 x raw
    logits; teacher_feat 3 x [B, teacher_dim].  Returns (new_sd, loss terms, grads)."""
    params = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    logits, cams, _ = forward_train_all(params, img, backbone, img_size, hidden, [t.to(img.dtype) for t in teacher_feat], masks)
    loss, terms = loss_all(logits, cams, labels, teacher_pred, [t.to(img.dtype) for t in teacher_feat], rates, temp)
    names = list(params)
    grads = torch.autograd.grad(loss, [params[k] for k in names], allow_unused=True)
    g = {k: (gr if gr is not None else torch.zeros_like(params[k])) for k, gr in zip(names, grads)}
    used = {k for k, gr in zip(names, grads) if gr is not None}
    new = {k: (params[k].detach() - lr * (g[k] + weight_decay * params[k].detach())) if k in used else params[k].detach().clone() for k in names}
    return new, terms, g


def train_step_all_f64(sd, img, labels, teacher_pred, teacher_feat, backbone, img_size, hidden, lr, weight_decay=1e-5, rates=(1.0, 0.0, 0.1), temp=4.0,
                       masks=None):
    return train_step_all({k: v.double() for k, v in sd.items()}, img.double(), labels, [t.double() for t in teacher_pred],
                          [t.double() for t in teacher_feat], backbone, img_size, hidden, lr, weight_decay, rates, temp, masks)
