"""Oracle for Temporal_mstct (`MSTCT/Temporal_Encoder.py`, `MSTCT/TS_Mixer.py`, `network.py`), eval mode.
Test infrastructure only."""
from __future__ import annotations

from typing import Dict

import torch
import torch.nn.functional as F

SD = Dict[str, torch.Tensor]


def _ln(sd, p, x):
    return F.layer_norm(x, (x.shape[-1],), sd[p + ".weight"], sd[p + ".bias"], 1e-5)


def _lin(sd, p, x):
    return F.linear(x, sd[p + ".weight"], sd[p + ".bias"])


def global_block(sd, p, x, heads):
    """`Global_Relational_Block.forward` (`Temporal_Encoder.py:76-88`)"""
    b, n, c = x.shape
    q = _lin(sd, p + ".q", x).reshape(b, n, heads, c // heads).permute(0, 2, 1, 3)
    kv = _lin(sd, p + ".kv", x).reshape(b, -1, 2, heads, c // heads).permute(2, 0, 3, 1, 4)
    k, v = kv[0], kv[1]
    attn = ((q @ k.transpose(-2, -1)) * ((c // heads) ** -0.5)).softmax(dim=-1)
    return _lin(sd, p + ".proj", (attn @ v).transpose(1, 2).reshape(b, n, c))


def local_block(sd, p, x):
    """`Local_Relational_Block.forward` (`Temporal_Encoder.py:34-43`): linear -> depthwise conv k3 -> GELU -> linear"""
    x = _lin(sd, p + ".linear1", x).transpose(1, 2)
    x = F.conv1d(x, sd[p + ".TC.weight"], sd[p + ".TC.bias"], padding=1, groups=x.shape[1]).transpose(1, 2)
    return _lin(sd, p + ".linear2", F.gelu(x))


def temporal_encoder(sd: SD, p: str, x: torch.Tensor, num_block=2, heads=8):
    """`TemporalEncoder.forward` (`Temporal_Encoder.py:222-256`): x [B,D,T] -> 4 x [B,C_s,T]"""
    outs = []
    for s in range(1, 5):
        m = f"{p}Temporal_Merging_Block{s}"
        x = F.conv1d(x, sd[m + ".proj.weight"], sd[m + ".proj.bias"], padding=1).transpose(1, 2)
        x = _ln(sd, m + ".norm", x)
        for b in range(num_block):
            q = f"{p}block{s}.{b}"
            x = x + global_block(sd, q + ".Global_Relational_Block", _ln(sd, q + ".norm1", x), heads)
            x = x + local_block(sd, q + ".Local_Relational_Block", _ln(sd, q + ".norm2", x))
        x = _ln(sd, f"{p}norm{s}", x).permute(0, 2, 1).contiguous()
        outs.append(x)
    return outs


def temporal_mixer(sd: SD, p: str, feats):
    """`Temporal_Mixer.forward` (`TS_Mixer.py:50-84`); interpolate to equal length == identity, kept as the call"""
    f1, f2, f3, f4 = feats
    t = f1.shape[2:]

    def lf(name, f, resize=True):
        y = F.linear(f.transpose(1, 2), sd[f"{p}{name}.proj.weight"], sd[f"{p}{name}.proj.bias"]).permute(0, 2, 1)
        return F.interpolate(y, size=t, mode="linear", align_corners=False) if resize else y

    _f4, _f3, _f2, _f1 = lf("linear_f4", f4), lf("linear_f3", f3), lf("linear_f2", f2), lf("linear_f1", f1, False)
    c = lambda i, z: F.conv1d(z, sd[f"{p}linear{i}.weight"], sd[f"{p}linear{i}.bias"])
    f3v, f2v, f1v = c(1, _f4) + _f3, c(2, _f4) + _f2, c(3, _f4) + _f1
    f3t, f2t, f1t = c(4, _f4) + _f3, c(5, _f4) + _f2, c(6, _f4) + _f1
    f3i = c(7, _f4) + _f3 + f3v + f3t
    f2i = c(8, _f4) + _f2 + f2v + f2t
    f1i = c(9, _f4) + _f1 + f1v + f1t
    return torch.cat([_f4, f3i, f2i, f1i], dim=1)


def mstct_forward(sd: SD, x: torch.Tensor, loss_type: str, num_block=2, heads=8):
    """`VideoNas.forward` (`Temporal_mstct/network.py:75-101`), eval: x [B,D,T] -> ((y_i,f),(y_v,f),(y_t,f),(y_ivt,concat))"""
    feats = temporal_encoder(sd, "TemporalEncoder.", x, num_block, heads)
    concat = temporal_mixer(sd, "Temporal_Mixer.", feats)
    b, t = x.shape[0], x.shape[-1]
    ys = {"i": torch.zeros(b, t, 6), "v": torch.zeros(b, t, 10), "t": torch.zeros(b, t, 15), "ivt": torch.zeros(b, t, 100)}
    fs = {k: concat for k in ys}
    q = f"classifier_{loss_type}"
    feat = F.conv1d(concat, sd[q + ".linear_fuse.weight"], sd[q + ".linear_fuse.bias"])
    ys[loss_type] = F.conv1d(feat, sd[q + ".linear_pred.weight"], sd[q + ".linear_pred.bias"]).permute(0, 2, 1)
    fs[loss_type] = feat
    return (ys["i"], fs["i"]), (ys["v"], fs["v"]), (ys["t"], fs["t"]), (ys["ivt"], concat)
