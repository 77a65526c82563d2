"""Spatial_transformer on MI355X: host-side mirror of `MT4MTLKD/Spatial_transformer/network.py`
(`Qeruy2Label`, `Decoder`, `GroupWiseLinear`, `build_q2l`) over the Swin backbone
(`models/swin_transformer.py`), sine position code (`models/position_encoding.py`) and the Q2L
transformer (`models/transformer.py`).  Same state-dict keys and return tuple as the reference.

Tokens stay row-major `[B*L][C]` end to end (the reference's `[B,C,h,w]` hand-off between backbone and
decoder, `swin_transformer.py:574-576` / `transformer.py:97`, is the same memory read as rows).
Per Swin block (`swin_transformer.py:234-271`): 7 launches --
  LN(norm1) fused with roll(-s)+window_partition as a row gather -> QKV GEMM -> attention core (dense
  relative-position bias + -100 shift mask) -> proj GEMM whose epilogue adds the shortcut and scatters rows
  back (window_reverse + roll(+s)) -> LN(norm2) -> fc1 GEMM + GELU -> fc2 GEMM + residual.
"""
from __future__ import annotations

import math
from typing import Dict, Optional

import torch

from . import ops
from .shapes import SWIN_BUFFER_SUFFIXES, SWIN_CFG, q2l_param_shapes, swin_window
from .synth import IMAGENET_MEAN, IMAGENET_STD

_K = {"i": 6, "v": 10, "t": 15, "ivt": 100}
MLP_CHAIN = True        # Swin blocks of width 128 / 256: Mlp + shortcut as ONE launch (`ops.chain_gemm`, bit-identical to fc1 -> fc2; tests switch it off)


def _window_row_map(res: int, ws: int, shift: int) -> torch.Tensor:
    """window-order row (win, pos) -> canonical token index h*res + w, with the cyclic shift folded in:
    shifted[hs][ws_] = x[(hs+shift) % res][(ws_+shift) % res]  (`torch.roll(x, -shift)`, :245-247) and
    windows enumerate (wh, ww, ph, pw) (`window_partition`, :34-47)."""
    nw = res // ws
    wh, ww, ph, pw = torch.meshgrid(torch.arange(nw), torch.arange(nw), torch.arange(ws), torch.arange(ws), indexing="ij")
    h = (wh * ws + ph + shift) % res
    w = (ww * ws + pw + shift) % res
    return (h * res + w).reshape(-1).to(torch.int32)


def _merge_row_map(res: int) -> torch.Tensor:
    """PatchMerging gather (`swin_transformer.py:320-324`): out (h2,w2) <- x0 (2h2,2w2), x1 (2h2+1,2w2), x2 (2h2,2w2+1), x3"""
    h2, w2 = torch.meshgrid(torch.arange(res // 2), torch.arange(res // 2), indexing="ij")
    parts = [(2 * h2) * res + 2 * w2, (2 * h2 + 1) * res + 2 * w2, (2 * h2) * res + 2 * w2 + 1, (2 * h2 + 1) * res + 2 * w2 + 1]
    return torch.stack(parts, -1).reshape(-1).to(torch.int32)


def _rel_pos_index(ws: int) -> torch.Tensor:
    """`swin_transformer.py:92-103`"""
    coords = torch.stack(torch.meshgrid(torch.arange(ws), torch.arange(ws), indexing="ij"))
    cf = coords.flatten(1)
    rel = (cf[:, :, None] - cf[:, None, :]).permute(1, 2, 0).contiguous()
    rel[:, :, 0] += ws - 1
    rel[:, :, 1] += ws - 1
    rel[:, :, 0] *= 2 * ws - 1
    return rel.sum(-1)


def _shift_regions(res: int, ws: int, shift: int) -> torch.Tensor:
    """`swin_transformer.py:210-221`: region id of every token of every window type, [nW, N] (ids 0 .. 8: `mt4_window_attention_rel_bf16` takes 0 .. 15)"""
    img = torch.zeros((res, res))
    cnt = 0
    for hs in (slice(0, -ws), slice(-ws, -shift), slice(-shift, None)):
        for wsl in (slice(0, -ws), slice(-ws, -shift), slice(-shift, None)):
            img[hs, wsl] = cnt
            cnt += 1
    nw = res // ws
    return img.view(nw, ws, nw, ws).permute(0, 2, 1, 3).reshape(nw * nw, ws * ws)


def _shift_mask(res: int, ws: int, shift: int) -> torch.Tensor:
    """`swin_transformer.py:210-229`: [nW, N, N], 0 / -100.0"""
    mw = _shift_regions(res, ws, shift)
    m = mw.unsqueeze(1) - mw.unsqueeze(2)
    return torch.where(m != 0, torch.full_like(m, -100.0), torch.zeros_like(m)).contiguous()


def sine_position_rows(hidden_dim: int, h: int, w: int) -> torch.Tensor:
    """`PositionEmbeddingSine._gen_pos_buffer` (`position_encoding.py:36-57`, normalize=True) as rows [h*w, hidden]"""
    npf = hidden_dim // 2
    eyes = torch.ones((1, h, w))
    y = eyes.cumsum(1, dtype=torch.float32)
    x = eyes.cumsum(2, dtype=torch.float32)
    y = y / (y[:, -1:, :] + 1e-6) * (2 * math.pi)
    x = x / (x[:, :, -1:] + 1e-6) * (2 * math.pi)
    dim_t = torch.arange(npf, dtype=torch.float32)
    dim_t = 10000 ** (2 * (dim_t // 2) / npf)
    px = x[:, :, :, None] / dim_t
    py = y[:, :, :, None] / dim_t
    px = torch.stack((px[:, :, :, 0::2].sin(), px[:, :, :, 1::2].cos()), dim=4).flatten(3)
    py = torch.stack((py[:, :, :, 0::2].sin(), py[:, :, :, 1::2].cos()), dim=4).flatten(3)
    return torch.cat((py, px), dim=3).reshape(h * w, hidden_dim).contiguous()


class Qeruy2Label:
    """Drop-in for `Spatial_transformer.network.Qeruy2Label` as built by `build_q2l` (eval path).

    args needs: backbone ('swin_{T,B,L}_{224,384}_*'), img_size, hidden_dim (= 8*embed_dim), loss_type ('i'|'v'|'t'|'all'),
    teacher_dim (KD adaptors, 'all' only; default 512 as `Spatial_transformer/test.py:82`).
    'all' = four decoders over ONE shared transformer (`network.py:66-73`) + the always-on KD mixing (`:98-124`), which
    needs the three teacher features as forward arguments exactly like the reference."""

    def __init__(self, args, dtype: torch.dtype = torch.float32, device: str = "cuda"):
        self.args = args
        self.backbone_name = args.backbone
        self.img_size = int(args.img_size)
        self.hidden = int(args.hidden_dim)
        self.loss_type = args.loss_type
        if self.loss_type not in ("i", "v", "t", "all"):
            raise ValueError(f"loss_type {self.loss_type}")
        self.tasks = ("i", "v", "t", "ivt") if self.loss_type == "all" else (self.loss_type,)
        self.batch_decoders = True     # loss_type all: the shared encoder layer once over the four tasks' tokens (decode_all)
        self.teacher_dim = int(getattr(args, "teacher_dim", 512))
        self.cfg = SWIN_CFG[self.backbone_name]
        self.dtype, self.device = dtype, torch.device(device)
        self.training = False
        self._table = q2l_param_shapes(self.backbone_name, self.img_size, self.hidden, self.loss_type, self.teacher_dim)
        self._sd: Dict[str, torch.Tensor] = {}
        self._p: Dict[str, object] = {}

    def eval(self):
        self.training = False
        return self

    def cuda(self):
        return self

    def state_dict(self):
        return dict(self._sd)

    def load_state_dict(self, sd: Dict[str, torch.Tensor], strict: bool = True):
        names = [k for k, _ in self._table]
        # a reference `loss_type all` checkpoint lists the shared transformer under every decoder: the copies are aliases
        extra = [k for k in sd if k not in names and not k.endswith(SWIN_BUFFER_SUFFIXES) and ".transformer." not in k]
        missing = [k for k in names if k not in sd]
        if strict and (missing or extra):
            raise KeyError(f"state dict mismatch: missing {missing[:4]}, unexpected {extra[:4]}")
        for k, shp in self._table:
            if k in sd:
                if tuple(sd[k].shape) != tuple(shp):
                    raise ValueError(f"{k}: shape {tuple(sd[k].shape)} != {shp}")
                self._sd[k] = sd[k].detach().float()
        self._pack()
        return self

    # ------------------------------------------------------------------ packing
    def _lin(self, key_w: str, key_b: Optional[str]):
        w = ops.pack_linear_weight(self._sd[key_w].to(self.device), self.dtype)
        b = self._sd[key_b].to(self.device).contiguous() if key_b else None
        return w, b

    def _ln(self, prefix: str):
        return self._sd[prefix + ".weight"].to(self.device).contiguous(), self._sd[prefix + ".bias"].to(self.device).contiguous()

    def _pack(self):
        dev, p = self.device, {}
        pre = "backbone.0."
        p["pe.proj"] = self._lin(pre + "patch_embed.proj.weight", pre + "patch_embed.proj.bias")
        p["pe.norm"] = self._ln(pre + "patch_embed.norm")
        self._stages = []
        for s, (depth, nh) in enumerate(zip(self.cfg["depths"], self.cfg["num_heads"])):
            ws, res = swin_window(self.backbone_name, self.img_size, s)
            idx = _rel_pos_index(ws).view(-1)
            blocks = []
            for b in range(depth):
                q = f"{pre}layers.{s}.blocks.{b}."
                shift = 0 if (b % 2 == 0 or res <= self.cfg["window_size"]) else self.cfg["window_size"] // 2
                tbl = self._sd[q + "attn.relative_position_bias_table"]
                n = ws * ws
                bias = tbl[idx].view(n, n, nh).permute(2, 0, 1).contiguous().to(dev)
                blocks.append(dict(
                    shift=shift, norm1=self._ln(q + "norm1"), norm2=self._ln(q + "norm2"),
                    qkv=self._lin(q + "attn.qkv.weight", q + "attn.qkv.bias"), proj=self._lin(q + "attn.proj.weight", q + "attn.proj.bias"),
                    fc1=self._lin(q + "mlp.fc1.weight", q + "mlp.fc1.bias"), fc2=self._lin(q + "mlp.fc2.weight", q + "mlp.fc2.bias"),
                    # Mlp + shortcut in one launch where the block's width allows it (`ops.chain_gemm`, MLP form: the 4C-wide hidden map never
                    # reaches memory; bit-identical to fc1 -> fc2): fragment-ordered copies of the two weight matrices
                    mlp_frag=self._mlp_frag(q, self.cfg["embed_dim"] * 2 ** s),
                    bias=bias, mask=_shift_mask(res, ws, shift).to(dev) if shift > 0 else None,
                    # MFMA core (bf16, head dim 32): the (2ws-1)^2 table per head and the region ids of the window types; the
                    # kernel gathers bias and mask from them in LDS
                    rel=tbl.t().contiguous().float().to(dev) if self._mfma_attn(s) else None,
                    region=_shift_regions(res, ws, shift).to(torch.int32).to(dev) if (shift > 0 and self._mfma_attn(s)) else None,
                    row_map=_window_row_map(res, ws, shift).to(dev)))
            st = dict(res=res, ws=ws, nh=nh, c=self.cfg["embed_dim"] * 2 ** s, blocks=blocks)
            if s < 3:
                q = f"{pre}layers.{s}.downsample."
                st["merge"] = dict(norm=self._ln(q + "norm"), red=self._lin(q + "reduction.weight", None), row_map=_merge_row_map(res).to(dev))
            self._stages.append(st)
        p["norm"] = self._ln(pre + "norm")
        # shared transformer (`network.py:66-73`: one object for all decoders), nn.MultiheadAttention in_proj split into q/k/v
        d = self.hidden
        t = f"decoder_{self.tasks[0]}.transformer."

        def mha(prefix):
            w, b = self._sd[prefix + ".in_proj_weight"], self._sd[prefix + ".in_proj_bias"]
            pk = lambda lo, hi: (ops.pack_linear_weight(w[lo:hi].to(dev), self.dtype), b[lo:hi].to(dev).contiguous())
            return dict(q=pk(0, d), k=pk(d, 2 * d), v=pk(2 * d, 3 * d), qk=pk(0, 2 * d),
                        out=self._lin(prefix + ".out_proj.weight", prefix + ".out_proj.bias"))

        e = t + "encoder.layers.0"
        p["enc"] = dict(attn=mha(e + ".self_attn"), l1=self._lin(e + ".linear1.weight", e + ".linear1.bias"),
                        l2=self._lin(e + ".linear2.weight", e + ".linear2.bias"), n1=self._ln(e + ".norm1"), n2=self._ln(e + ".norm2"))
        p["dec"] = []
        for li in range(2):
            dl = f"{t}decoder.layers.{li}"
            p["dec"].append(dict(attn=mha(dl + ".multihead_attn"), l1=self._lin(dl + ".linear1.weight", dl + ".linear1.bias"),
                                 l2=self._lin(dl + ".linear2.weight", dl + ".linear2.bias"), n2=self._ln(dl + ".norm2"), n3=self._ln(dl + ".norm3")))
        p["dec_norm"] = self._ln(t + "decoder.norm")
        hh = self.img_size // 32
        p["pos"] = sine_position_rows(d, hh, hh).to(dev, self.dtype)
        # per-task pieces (`Decoder.__init__`, `network.py:144-161`)
        p["task"] = {}
        for task in self.tasks:
            q = f"decoder_{task}."
            p["task"][task] = dict(in_proj=self._lin(q + "input_proj.weight", q + "input_proj.bias"),
                                   query=self._sd[q + "query_embed.weight"].to(dev, self.dtype).contiguous(),
                                   W=self._sd[q + "fc.W"][0].to(dev).contiguous(), b=self._sd[q + "fc.b"][0].to(dev).contiguous())
        if self.loss_type == "all":   # KD adaptors, fp32 (`network.py:75-80`)
            for n in ("wi", "wv", "wt", "mi", "mv", "mt"):
                p[n] = (ops.pack_linear_weight(self._sd[n + ".weight"].to(dev), torch.float32), self._sd[n + ".bias"].to(dev).contiguous())
        self._p = p

    def _mlp_frag(self, q: str, c: int):
        if self.dtype != torch.bfloat16 or not MLP_CHAIN or not ops.chain_gemm_supported(c, 4 * c, c, False):
            return None
        return ops.pack_fragments(self._lin(q + "mlp.fc1.weight", None)[0]), ops.pack_fragments(self._lin(q + "mlp.fc2.weight", None)[0])

    def _mfma_attn(self, stage: int) -> bool:
        c = self.cfg["embed_dim"] * 2 ** stage
        return self.dtype == torch.bfloat16 and c // self.cfg["num_heads"][stage] == 32

    # ------------------------------------------------------------------ backbone
    def _block(self, x, st, blk, batch):
        c, nh, ws, res = st["c"], st["nh"], st["ws"], st["res"]
        L, n, nwin = res * res, ws * ws, (res // ws) ** 2
        xw = ops.layernorm(x, *blk["norm1"], row_map=blk["row_map"], group=1, l_out=L, l_in=L)
        qkv = ops.linear(xw, *blk["qkv"])
        if blk["rel"] is not None:
            a = ops.window_attention_rel_bf16(qkv[:, :c], qkv[:, c:2 * c], qkv[:, 2 * c:], batch=batch * nwin, heads=nh, ws=ws, q_stride=3 * c,
                                              k_stride=3 * c, v_stride=3 * c, scale=(c // nh) ** -0.5, rel_table=blk["rel"],
                                              region=blk["region"])
        else:
            a = ops.attention(qkv[:, :c], qkv[:, c:2 * c], qkv[:, 2 * c:], batch=batch * nwin, heads=nh, nq=n, nk=n, hd=c // nh,
                              q_stride=3 * c, k_stride=3 * c, v_stride=3 * c, scale=(c // nh) ** -0.5, bias=blk["bias"], mask=blk["mask"])
        x = ops.linear(a, *blk["proj"], residual=x, out_row_map=blk["row_map"])
        y = ops.layernorm(x, *blk["norm2"])
        if blk["mlp_frag"] is not None and ops.chain_gemm_pays(y.shape[0]):
            return ops.chain_gemm(y, blk["mlp_frag"][0], blk["fc1"][1], blk["mlp_frag"][1], blk["fc2"][1], r2=x)[1]
        h = ops.linear(y, *blk["fc1"], act="gelu")
        return ops.linear(h, *blk["fc2"], residual=x)

    def forward_features(self, img: torch.Tensor) -> torch.Tensor:
        """img: normalised float32 NCHW or uint8 NHWC frames -> final-norm tokens [B*h*h, 8*embed] (rows of the
        reference's [B,C,h,h], `swin_transformer.py:565-577`)."""
        p = self._p
        batch = img.shape[0]
        x = ops.patchify(img, 4, self.dtype, IMAGENET_MEAN, IMAGENET_STD)
        x = ops.layernorm(ops.linear(x, *p["pe.proj"]), *p["pe.norm"])
        for st in self._stages:
            for blk in st["blocks"]:
                x = self._block(x, st, blk, batch)
            if "merge" in st:
                mg, res = st["merge"], st["res"]
                xm = ops.layernorm(x, *mg["norm"], row_map=mg["row_map"], group=4, l_out=(res // 2) ** 2, l_in=res * res,
                                   m_out=batch * (res // 2) ** 2)
                x = ops.linear(xm, *mg["red"])
        return ops.layernorm(x, *p["norm"])

    # ------------------------------------------------------------------ Q2L decoder
    def _mha(self, a, q_in, k_in, v_in, batch, nq, nk, residual):
        d, nhead = self.hidden, 4
        if q_in is k_in:  # encoder self-attention: q and k share their input -> one [2d] projection
            qk = ops.linear(q_in, *a["qk"])
            qq, kk, qs, ks = qk[:, :d], qk[:, d:], 2 * d, 2 * d
        else:
            qq, kk, qs, ks = ops.linear(q_in, *a["q"]), ops.linear(k_in, *a["k"]), d, d
        vv = ops.linear(v_in, *a["v"])
        o = ops.attention(qq, kk, vv, batch=batch, heads=nhead, nq=nq, nk=nk, hd=d // nhead, q_stride=qs, k_stride=ks, v_stride=d,
                          scale=(d // nhead) ** -0.5)
        return ops.linear(o, *a["out"], residual=residual)

    def _encode(self, s: torch.Tensor, batch: int, L: int):
        """the transformer's encoder layer + pooled feature on projected tokens s [batch * L, d] (`transformer.py:64-98`): -> (memory, feat [batch, d] fp32)"""
        p = self._p
        e = p["enc"]
        sp = ops.add_rowbcast(s, p["pos"])
        s = ops.layernorm(self._mha(e["attn"], sp, sp, s, batch, L, L, residual=s), *e["n1"])
        s = ops.layernorm(ops.linear(ops.linear(s, *e["l1"], act="relu"), *e["l2"], residual=s), *e["n2"])
        return s, ops.global_avgpool(s.view(batch, L, 1, self.hidden))

    def _decode_queries(self, memory: torch.Tensor, batch: int, L: int, tp):
        """the decoder layers of one task on its encoder memory [batch * L, d] -> logits [batch, K] fp32"""
        p = self._p
        kq = tp["query"].shape[0]
        mem_pos = ops.add_rowbcast(memory, p["pos"])
        tgt = torch.zeros((batch * kq, self.hidden), dtype=self.dtype, device=memory.device)
        for dl in p["dec"]:
            qin = ops.add_rowbcast(tgt, tp["query"])
            tgt = ops.layernorm(self._mha(dl["attn"], qin, mem_pos, memory, batch, kq, L, residual=tgt), *dl["n2"])
            tgt = ops.layernorm(ops.linear(ops.linear(tgt, *dl["l1"], act="relu"), *dl["l2"], residual=tgt), *dl["n3"])
        hs = ops.layernorm(tgt, *p["dec_norm"])
        return ops.groupwise_linear(hs, tp["W"], tp["b"], batch, kq)

    def decode(self, src_tokens: torch.Tensor, batch: int, task: Optional[str] = None):
        """`Decoder.forward` (`network.py:163-171`): tokens [B*L, C] -> (feat [B,d] fp32, logits [B,K] fp32)"""
        tp = self._p["task"][task or self.tasks[0]]
        L = src_tokens.shape[0] // batch
        memory, feat = self._encode(ops.linear(src_tokens, *tp["in_proj"]), batch, L)
        return feat, self._decode_queries(memory, batch, L, tp)

    def decode_all(self, src_tokens: torch.Tensor, batch: int):
        """the four decoders of `loss_type all` (`network.py:66-73,90-100`).  They share ONE transformer: its encoder layer runs once over the four
        tasks' projected tokens stacked along the batch axis (rows, attention batches and LayerNorm rows are independent, so every task's memory
        is bit-identical to its own `decode` call: 4 x fewer, 4 x larger launches); the query side runs per task (6 / 10 / 15 / 100 queries).
        -> (feat of the last task, {task: logits})"""
        p = self._p
        L = src_tokens.shape[0] // batch
        n = batch * L
        s = torch.empty((len(self.tasks) * n, self.hidden), dtype=self.dtype, device=src_tokens.device)
        for i, task in enumerate(self.tasks):
            ops.linear(src_tokens, *p["task"][task]["in_proj"], out=s[i * n:(i + 1) * n])
        memory, feat = self._encode(s, len(self.tasks) * batch, L)
        ys = {task: self._decode_queries(memory[i * n:(i + 1) * n], batch, L, p["task"][task]) for i, task in enumerate(self.tasks)}
        return feat[(len(self.tasks) - 1) * batch:].contiguous(), ys

    def forward(self, input: torch.Tensor, tool=None, verb=None, target=None):
        if self.training:
            raise NotImplementedError("training path is a later row")
        if not self._p:
            raise RuntimeError("load_state_dict first")
        b = input.shape[0]
        src = self.forward_features(input)
        if self.loss_type != "all":
            feat, y = self.decode(src, b)
            ys = {k: torch.zeros((b, n), device=feat.device) for k, n in _K.items()}  # `network.py:85-88`
            ys[self.loss_type] = y
            return (0, ys["i"]), (0, ys["v"]), (0, ys["t"]), (feat, ys["ivt"])
        if self.batch_decoders:
            feat, ys = self.decode_all(src, b)          # four decoders, shared transformer weights; feat = the last one's (`:100`)
        else:
            ys = {}
            for task in self.tasks:
                feat, ys[task] = self.decode(src, b, task)
        if tool is None or verb is None or target is None:
            raise TypeError("loss_type 'all' runs the KD mixing unconditionally (network.py:98-124): pass tool, verb, target features")
        p = self._p
        teas = [ops.linear(t.contiguous().float(), *p[m]) for m, t in zip(("mi", "mv", "mt"), (tool, verb, target))]
        mixed = ops.kd_mix(feat, *teas)
        kd = [ops.linear(mx, *p[w]) for mx, w in zip(mixed, ("wi", "wv", "wt"))]
        return (kd[0], ys["i"]), (kd[1], ys["v"]), (kd[2], ys["t"]), (feat, ys["ivt"])

    __call__ = forward


def extraction_batch(img_size: int, device_batch: int) -> int:
    """frames per extraction pass (a frame's features do not depend on it): the GEMMs run 256 x 256 tiles on 256 CUs, and with 227 frames of 384^2 /
    668 of 224^2 the stage-2 / stage-3 / decoder launches with 2 or 4 column tiles are whole rounds of the chip (at 256 frames of 384^2 they run
    2.25 rounds: profiles/r04_swin_batch_sweep.txt); capped by --device_batch"""
    return max(1, min(device_batch, {384: 227, 224: 668}.get(int(img_size), 256)))


def build_q2l(args, dtype: torch.dtype = torch.float32, device: str = "cuda") -> Qeruy2Label:
    """`network.py:187-204`"""
    return Qeruy2Label(args, dtype=dtype, device=device)
