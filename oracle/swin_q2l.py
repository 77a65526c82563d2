"""Oracle for Spatial_transformer: Swin backbone (`models/swin_transformer.py`), sine position encoding
(`models/position_encoding.py`), Q2L transformer (`models/transformer.py`) and `Qeruy2Label`/`Decoder`/
`GroupWiseLinear` (`network.py`), eval mode.  Test infrastructure only."""
from __future__ import annotations

import math
from typing import Dict

import torch
import torch.nn.functional as F

SD = Dict[str, torch.Tensor]

SWIN_CFG = {  # `swin_transformer.py:596-631`
    "swin_T_224_1k": dict(embed_dim=96, depths=(2, 2, 6, 2), num_heads=(3, 6, 12, 24), window_size=7),
    "swin_B_224_22k": dict(embed_dim=128, depths=(2, 2, 18, 2), num_heads=(4, 8, 16, 32), window_size=7),
    "swin_B_384_22k": dict(embed_dim=128, depths=(2, 2, 18, 2), num_heads=(4, 8, 16, 32), window_size=12),
    "swin_L_224_22k": dict(embed_dim=192, depths=(2, 2, 18, 2), num_heads=(6, 12, 24, 48), window_size=7),
    "swin_L_384_22k": dict(embed_dim=192, depths=(2, 2, 18, 2), num_heads=(6, 12, 24, 48), window_size=12),
}


def _ln(sd, p, x):
    return F.layer_norm(x, (x.shape[-1],), sd[p + ".weight"], sd[p + ".bias"], 1e-5)


def window_partition(x, ws):
    """`swin_transformer.py:34-47`"""
    b, h, w, c = x.shape
    x = x.view(b, h // ws, ws, w // ws, ws, c)
    return x.permute(0, 1, 3, 2, 4, 5).contiguous().view(-1, ws, ws, c)


def window_reverse(win, ws, h, w):
    """`swin_transformer.py:50-63`"""
    b = int(win.shape[0] / (h * w / ws / ws))
    x = win.view(b, h // ws, w // ws, ws, ws, -1)
    return x.permute(0, 1, 3, 2, 4, 5).contiguous().view(b, h, w, -1)


def relative_position_index(ws):
    """`swin_transformer.py:92-103`"""
    coords = torch.stack(torch.meshgrid([torch.arange(ws), torch.arange(ws)], indexing="ij"))
    cf = torch.flatten(coords, 1)
    rel = (cf[:, :, None] - cf[:, None, :]).permute(1, 2, 0).contiguous()
    rel[:, :, 0] += ws - 1
    rel[:, :, 1] += ws - 1
    rel[:, :, 0] *= 2 * ws - 1
    return rel.sum(-1)


def shift_attn_mask(h, w, ws, shift):
    """`swin_transformer.py:210-229` (fill value -100.0)"""
    img = torch.zeros((1, h, w, 1))
    cnt = 0
    for hs in (slice(0, -ws), slice(-ws, -shift), slice(-shift, None)):
        for wsl in (slice(0, -ws), slice(-ws, -shift), slice(-shift, None)):
            img[:, hs, wsl, :] = cnt
            cnt += 1
    mw = window_partition(img, ws).view(-1, ws * ws)
    m = mw.unsqueeze(1) - mw.unsqueeze(2)
    return m.masked_fill(m != 0, -100.0).masked_fill(m == 0, 0.0)


def window_attention(sd, p, x, nh, ws, mask):
    """`WindowAttention.forward` (`swin_transformer.py:114-145`)"""
    b_, n, c = x.shape
    qkv = F.linear(x, sd[p + ".qkv.weight"], sd[p + ".qkv.bias"]).reshape(b_, n, 3, nh, c // nh).permute(2, 0, 3, 1, 4)
    q, k, v = qkv[0], qkv[1], qkv[2]
    q = q * ((c // nh) ** -0.5)
    attn = q @ k.transpose(-2, -1)
    idx = relative_position_index(ws)
    bias = sd[p + ".relative_position_bias_table"][idx.view(-1)].view(n, n, -1).permute(2, 0, 1).contiguous()
    attn = attn + bias.unsqueeze(0)
    if mask is not None:
        nw = mask.shape[0]
        attn = attn.view(b_ // nw, nw, nh, n, n) + mask.unsqueeze(1).unsqueeze(0)
        attn = attn.view(-1, nh, n, n)
    attn = attn.softmax(dim=-1)
    x = (attn @ v).transpose(1, 2).reshape(b_, n, c)
    return F.linear(x, sd[p + ".proj.weight"], sd[p + ".proj.bias"])


def swin_block(sd, p, x, res, nh, ws, shift):
    """`SwinTransformerBlock.forward` (`swin_transformer.py:234-271`)"""
    h, w = res
    if min(res) <= ws:  # :193-196
        shift, ws = 0, min(res)
    b, l, c = x.shape
    shortcut = x
    x = _ln(sd, p + ".norm1", x).view(b, h, w, c)
    if shift > 0:
        x = torch.roll(x, shifts=(-shift, -shift), dims=(1, 2))
    xw = window_partition(x, ws).view(-1, ws * ws, c)
    mask = shift_attn_mask(h, w, ws, shift) if shift > 0 else None
    aw = window_attention(sd, p + ".attn", xw, nh, ws, mask).view(-1, ws, ws, c)
    x = window_reverse(aw, ws, h, w)
    if shift > 0:
        x = torch.roll(x, shifts=(shift, shift), dims=(1, 2))
    x = shortcut + x.view(b, h * w, c)
    y = _ln(sd, p + ".norm2", x)
    y = F.linear(F.gelu(F.linear(y, sd[p + ".mlp.fc1.weight"], sd[p + ".mlp.fc1.bias"])), sd[p + ".mlp.fc2.weight"],
                 sd[p + ".mlp.fc2.bias"])
    return x + y


def patch_merging(sd, p, x, res):
    """`PatchMerging.forward` (`swin_transformer.py:308-329`)"""
    h, w = res
    b, l, c = x.shape
    x = x.view(b, h, w, c)
    x = torch.cat([x[:, 0::2, 0::2], x[:, 1::2, 0::2], x[:, 0::2, 1::2], x[:, 1::2, 1::2]], -1).view(b, -1, 4 * c)
    return F.linear(_ln(sd, p + ".norm", x), sd[p + ".reduction.weight"])


def swin_forward_features(sd: SD, x: torch.Tensor, name: str, img_size: int, prefix: str = "") -> torch.Tensor:
    """`SwinTransformer.forward_features` (`swin_transformer.py:565-577`): [B,3,S,S] -> [B,8*embed,S/32,S/32]"""
    cfg = SWIN_CFG[name]
    x = F.conv2d(x, sd[prefix + "patch_embed.proj.weight"], sd[prefix + "patch_embed.proj.bias"], stride=4)
    x = x.flatten(2).transpose(1, 2)
    x = _ln(sd, prefix + "patch_embed.norm", x)
    r = img_size // 4
    for s, (depth, nh) in enumerate(zip(cfg["depths"], cfg["num_heads"])):
        res = (r // (2 ** s), r // (2 ** s))
        for bi in range(depth):
            shift = 0 if bi % 2 == 0 else cfg["window_size"] // 2
            x = swin_block(sd, f"{prefix}layers.{s}.blocks.{bi}", x, res, nh, cfg["window_size"], shift)
        if s < 3:
            x = patch_merging(sd, f"{prefix}layers.{s}.downsample", x, res)
    x = _ln(sd, prefix + "norm", x)
    b, l, c = x.shape
    hh = img_size // 32
    return x.transpose(1, 2).reshape(b, c, hh, hh)


def sine_position_encoding(hidden_dim: int, h: int, w: int) -> torch.Tensor:
    """`PositionEmbeddingSine._gen_pos_buffer` (`position_encoding.py:36-57`), normalize=True: [1,hidden,h,w]"""
    npf = hidden_dim // 2
    eyes = torch.ones((1, h, w))
    y = eyes.cumsum(1, dtype=torch.float32)
    x = eyes.cumsum(2, dtype=torch.float32)
    eps, scale = 1e-6, 2 * math.pi
    y = y / (y[:, -1:, :] + eps) * scale
    x = x / (x[:, :, -1:] + eps) * scale
    dim_t = torch.arange(npf, dtype=torch.float32)
    dim_t = 10000 ** (2 * (dim_t // 2) / npf)
    px = x[:, :, :, None] / dim_t
    py = y[:, :, :, None] / dim_t
    px = torch.stack((px[:, :, :, 0::2].sin(), px[:, :, :, 1::2].cos()), dim=4).flatten(3)
    py = torch.stack((py[:, :, :, 0::2].sin(), py[:, :, :, 1::2].cos()), dim=4).flatten(3)
    return torch.cat((py, px), dim=3).permute(0, 3, 1, 2)


def _mha(sd, p, q, k, v, nhead):
    """nn.MultiheadAttention forward (seq-first [L,B,E]), no masks, eval"""
    e = q.shape[-1]
    w, b = sd[p + ".in_proj_weight"], sd[p + ".in_proj_bias"]
    qq = F.linear(q, w[:e], b[:e])
    kk = F.linear(k, w[e:2 * e], b[e:2 * e])
    vv = F.linear(v, w[2 * e:], b[2 * e:])
    lq, bs, _ = qq.shape
    lk = kk.shape[0]
    hd = e // nhead
    qq = qq.reshape(lq, bs * nhead, hd).transpose(0, 1) * (hd ** -0.5)
    kk = kk.reshape(lk, bs * nhead, hd).transpose(0, 1)
    vv = vv.reshape(lk, bs * nhead, hd).transpose(0, 1)
    a = torch.softmax(qq @ kk.transpose(1, 2), dim=-1) @ vv
    a = a.transpose(0, 1).reshape(lq, bs, e)
    return F.linear(a, sd[p + ".out_proj.weight"], sd[p + ".out_proj.bias"])


def q2l_transformer(sd: SD, p: str, src, query_embed, pos, nhead=4):
    """`Transformer.forward` (`transformer.py:95-113`), 1 post-norm encoder layer, 2 decoder layers with
    self-attention removed (`:59-76`), decoder.norm, return_intermediate False."""
    bs, c, h, w = src.shape
    src = src.flatten(2).permute(2, 0, 1)
    pos = pos.flatten(2).permute(2, 0, 1)
    qe = query_embed.unsqueeze(1).repeat(1, bs, 1)
    e = p + "encoder.layers.0"
    qk = src + pos
    s2 = _mha(sd, e + ".self_attn", qk, qk, src, nhead)
    src = _ln(sd, e + ".norm1", src + s2)
    s2 = F.linear(F.relu(F.linear(src, sd[e + ".linear1.weight"], sd[e + ".linear1.bias"])), sd[e + ".linear2.weight"],
                  sd[e + ".linear2.bias"])
    memory = _ln(sd, e + ".norm2", src + s2)
    tgt = torch.zeros_like(qe)
    for li in range(2):
        d = f"{p}decoder.layers.{li}"
        t2 = _mha(sd, d + ".multihead_attn", tgt + qe, memory + pos, memory, nhead)
        tgt = _ln(sd, d + ".norm2", tgt + t2)
        t2 = F.linear(F.relu(F.linear(tgt, sd[d + ".linear1.weight"], sd[d + ".linear1.bias"])), sd[d + ".linear2.weight"],
                      sd[d + ".linear2.bias"])
        tgt = _ln(sd, d + ".norm3", tgt + t2)
    tgt = _ln(sd, p + "decoder.norm", tgt)
    hs = tgt.unsqueeze(0).transpose(1, 2)                       # [1,B,K,d]
    mem = memory[:h * w].permute(1, 2, 0).view(bs, c, h, w)
    return hs, mem


def q2l_decoder(sd: SD, p: str, src, pos):
    """`Decoder.forward` (`network.py:163-171`) + `GroupWiseLinear.forward` (`:40-45`)"""
    x = F.conv2d(src, sd[p + "input_proj.weight"], sd[p + "input_proj.bias"])
    hs, mem = q2l_transformer(sd, p + "transformer.", x, sd[p + "query_embed.weight"], pos)
    out = (sd[p + "fc.W"] * hs[-1]).sum(-1) + sd[p + "fc.b"]
    feat = mem.mean(dim=(2, 3))
    return feat, out


def q2l_forward(sd: SD, img: torch.Tensor, backbone: str, img_size: int, hidden_dim: int, loss_type: str, teacher=None):
    """`Qeruy2Label.forward` (`network.py:82-128`).  Single-task loss_type ('i'|'v'|'t'): one decoder, KD slots are 0.
    'all': four decoders over ONE shared transformer (its weights live under decoder_i.transformer.*, `network.py:66-73`),
    feat = decoder_ivt's pooled memory, and the always-on KD mixing (`:98-124`; reduced form, see oracle/spatial_cnn.kd_branch)
    with teacher = (tool, verb, target) features [B, teacher_dim]."""
    src = swin_forward_features(sd, img, backbone, img_size, prefix="backbone.0.")
    pos = sine_position_encoding(hidden_dim, img_size // 32, img_size // 32).repeat(src.shape[0], 1, 1, 1)
    b = img.shape[0]
    if loss_type != "all":
        feat, y = q2l_decoder(sd, f"decoder_{loss_type}.", src, pos)
        ys = {"i": torch.zeros(b, 6), "v": torch.zeros(b, 10), "t": torch.zeros(b, 15), "ivt": torch.zeros(b, 100)}
        ys[loss_type] = y
        return (0, ys["i"]), (0, ys["v"]), (0, ys["t"]), (feat, ys["ivt"])
    ys = {}
    for task in ("i", "v", "t", "ivt"):
        sdt = dict(sd)
        if task != "i":   # the shared Transformer: alias decoder_i's weights
            for k, v in sd.items():
                if k.startswith("decoder_i.transformer."):
                    sdt[k.replace("decoder_i.", f"decoder_{task}.", 1)] = v
        feat, ys[task] = q2l_decoder(sdt, f"decoder_{task}.", src, pos)
    s = feat
    c = s.shape[1]
    teas = [F.conv1d(t.unsqueeze(-1), sd[f"{m}.weight"], sd[f"{m}.bias"]).squeeze(-1) for m, t in zip(("mi", "mv", "mt"), teacher)]
    tsum = torch.stack([t.sum(dim=1) for t in teas], dim=-1)
    attn = torch.softmax((s / (c ** 0.5)).unsqueeze(-1) * tsum.unsqueeze(1), dim=-1)
    kd = [F.conv1d((s * attn[:, :, n]).unsqueeze(-1), sd[f"{w}.weight"], sd[f"{w}.bias"]).squeeze(-1) for n, w in enumerate(("wi", "wv", "wt"))]
    return (kd[0], ys["i"]), (kd[1], ys["v"]), (kd[2], ys["t"]), (feat, ys["ivt"])
