"""Temporal_mstct (MS-TCT teacher) on MI355X: host-side mirror of `MT4MTLKD/Temporal_mstct/network.py`
(VideoNas, Classifier) over `MSTCT/Temporal_Encoder.py` and `MSTCT/TS_Mixer.py`.  Same constructor,
state-dict keys and return tuple as the reference.

Frames stay row-major `[B*T][C]`; the reference's `[B,C,T]` tensors are permuted views of those rows.
Per GLRBlock (`Temporal_Encoder.py:123-126`): LN -> q GEMM + kv GEMM -> attention core -> proj GEMM (+x)
-> LN -> linear1 GEMM -> depthwise conv k3 + GELU -> linear2 GEMM (+x).  The mixer's nine 1x1 convs
collapse algebraically: `_f3_ivt = linear7(_f4)+_f3+(linear1(_f4)+_f3)+(linear4(_f4)+_f3)`
(`TS_Mixer.py:70-81`) = `(W7+W1+W4) _f4 + 3 _f3`, so each scale is ONE GEMM with summed weights whose
residual is the (3x pre-scaled) `linear_f` output; results go straight into their column slice of the
`[B*T][4*E]` concat buffer.
"""
from __future__ import annotations

from typing import Dict

import torch

from . import ops
from .shapes import mstct_shapes

_K = {"i": 6, "v": 10, "t": 15, "ivt": 100}


class VideoNas:
    """Drop-in for `Temporal_mstct.network.VideoNas` (eval path).  args needs: loss_type ('i'|'v'|'t'|'ivt')."""

    def __init__(self, args, inter_channels, num_block, head, mlp_ratio, in_feat_dim, final_embedding_dim, num_tool=6, num_verb=10,
                 num_target=15, num_triplet=100, dtype: torch.dtype = torch.float32, device: str = "cuda"):
        self.args = args
        self.loss_type = args.loss_type
        self.inter = tuple(inter_channels)
        self.num_block, self.head, self.mlp_ratio = num_block, head, mlp_ratio
        self.D, self.E = in_feat_dim, final_embedding_dim
        self.dtype, self.device = dtype, torch.device(device)
        self.training = False
        self.fold_layernorm = True      # fp32, one short window: norm1 / norm2 ride in the launch of the nn.Linear behind them (ops.linear_ln)
        self.fold_linear1_max_c = 256   # ... norm2 -> linear1 only up to this width (profiles/r04_mstct_ln_fold_ab.txt)
        self._table = mstct_shapes(in_feat_dim, self.inter, num_block, mlp_ratio, final_embedding_dim, self.loss_type)
        self._sd: Dict[str, torch.Tensor] = {}
        self._p: Dict[str, object] = {}

    def eval(self):
        self.training = False
        return self

    def cuda(self):
        return self

    def state_dict(self):
        return dict(self._sd)

    def load_state_dict(self, sd, strict: bool = True):
        names = [k for k, _ in self._table]
        missing = [k for k in names if k not in sd]
        if strict and (missing or len(sd) != len(names)):
            raise KeyError(f"state dict mismatch: missing {missing[:4]}, unexpected {[k for k in sd if k not in names][:4]}")
        for k, shp in self._table:
            if k in sd:
                if tuple(sd[k].shape) != tuple(shp):
                    raise ValueError(f"{k}: shape {tuple(sd[k].shape)} != {shp}")
                self._sd[k] = sd[k].detach().float()
        self._pack()
        return self

    def _pack(self):
        dev, sd, dt = self.device, self._sd, self.dtype
        lin = lambda k, scale=1.0: (ops.pack_linear_weight((sd[k + ".weight"] * scale).to(dev), dt), (sd[k + ".bias"] * scale).to(dev).contiguous())
        ln = lambda k: (sd[k + ".weight"].to(dev).contiguous(), sd[k + ".bias"].to(dev).contiguous())
        p = {"stages": []}
        for s in range(1, 5):
            m = f"TemporalEncoder.Temporal_Merging_Block{s}"
            w = sd[m + ".proj.weight"].to(dev)  # [C, Cin, 3] -> OIHW [C, Cin, 1, 3]
            st = dict(merge=(ops.pack_conv_weight(w.unsqueeze(2), None, dt), sd[m + ".proj.bias"].to(dev).contiguous()),
                      merge_norm=ln(m + ".norm"), norm=ln(f"TemporalEncoder.norm{s}"), blocks=[])
            for b in range(self.num_block):
                q = f"TemporalEncoder.block{s}.{b}"
                g, l = q + ".Global_Relational_Block", q + ".Local_Relational_Block"
                # q and kv (`Temporal_Encoder.py:80-84`: two nn.Linear on the same normalised input) as ONE GEMM over the stacked weights
                qkv = (ops.pack_linear_weight(torch.cat([sd[g + ".q.weight"], sd[g + ".kv.weight"]], 0).to(dev), dt),
                       torch.cat([sd[g + ".q.bias"], sd[g + ".kv.bias"]], 0).to(dev).contiguous())
                blk = dict(n1=ln(q + ".norm1"), n2=ln(q + ".norm2"), qkv=qkv, proj=lin(g + ".proj"),
                           l1=lin(l + ".linear1"), l2=lin(l + ".linear2"),
                           tc=(sd[l + ".TC.weight"][:, 0, :].to(dev).contiguous(), sd[l + ".TC.bias"].to(dev).contiguous()))
                if dt == torch.float32:
                    # one short window (latency context): norm1 -> q | kv and norm2 -> linear1 run as ONE launch each, the LayerNorm folded into
                    # the GEMM (`ops.linear_ln`): gamma into the weight, mean / rstd from the operand fragments, beta into the bias
                    def folded(w, b, n):
                        wf, cs, bf = ops.fold_layernorm(w, b, sd[n + ".weight"], sd[n + ".bias"])
                        return ops.pack_linear_weight(wf.to(dev), dt), cs.to(dev), bf.to(dev)
                    blk["qkv_ln"] = folded(torch.cat([sd[g + ".q.weight"], sd[g + ".kv.weight"]], 0), torch.cat([sd[g + ".q.bias"], sd[g + ".kv.bias"]], 0),
                                           q + ".norm1")
                    blk["l1_ln"] = folded(sd[l + ".linear1.weight"], sd[l + ".linear1.bias"], q + ".norm2")
                st["blocks"].append(blk)
            p["stages"].append(st)
        mx = "Temporal_Mixer."
        p["f4"] = lin(mx + "linear_f4.proj")
        # scales 3,2,1: residual term is 3 * linear_f(f_s); GEMM term uses the summed 1x1 weights
        for name, fkey, ids in (("s3", "linear_f3", (7, 1, 4)), ("s2", "linear_f2", (8, 2, 5)), ("s1", "linear_f1", (9, 3, 6))):
            p[name + ".f"] = lin(mx + fkey + ".proj", 3.0)
            wsum = sum(sd[f"{mx}linear{i}.weight"] for i in ids)
            bsum = sum(sd[f"{mx}linear{i}.bias"] for i in ids)
            p[name + ".mix"] = (ops.pack_linear_weight(wsum.to(dev), dt), bsum.to(dev).contiguous())
        c = f"classifier_{self.loss_type}"
        p["fuse"] = lin(c + ".linear_fuse")
        p["pred"] = lin(c + ".linear_pred")
        self._p = p

    def _block(self, x, blk, b, t, c, stats=None):
        """one encoder block on rows x [B*T, C]; `stats` = the LayerNorm partials of x if the launch that produced x left them.  Returns (x, stats)"""
        hd = c // self.head
        fold = self.fold_layernorm and "qkv_ln" in blk and ops.latency_linear_ok(b * t, c, self.dtype) and c % 16 == 0
        if not fold:
            qkv = ops.linear(ops.layernorm(x, *blk["n1"]), *blk["qkv"])       # [B*T, 3C]: q | k | v column slices
            a = ops.attention(qkv[:, :c], qkv[:, c:2 * c], qkv[:, 2 * c:], batch=b, heads=self.head, nq=t, nk=t, hd=hd, q_stride=3 * c, k_stride=3 * c,
                              v_stride=3 * c, scale=hd ** -0.5)
            x = ops.linear(a, *blk["proj"], residual=x)
            h = ops.linear(ops.layernorm(x, *blk["n2"]), *blk["l1"])
            h = ops.dwconv1d_k3(h.view(b, t, -1), *blk["tc"], act="gelu").view(b * t, -1)
            return ops.linear(h, *blk["l2"], residual=x), None
        # one short window: norm1 / norm2 ride in the nn.Linear behind them (gamma in the weight, beta in the bias), their row statistics come as
        # partial sums from the epilogue of the launch that wrote the row (proj / linear2 of the block before); a block whose input has none
        # (behind the stage's merge norm) normalises in a launch of its own
        qkv = ops.linear_ln(x, *blk["qkv_ln"], stats_in=stats) if stats is not None else ops.linear(ops.layernorm(x, *blk["n1"]), *blk["qkv"])
        a = ops.attention(qkv[:, :c], qkv[:, c:2 * c], qkv[:, 2 * c:], batch=b, heads=self.head, nq=t, nk=t, hd=hd, q_stride=3 * c, k_stride=3 * c,
                          v_stride=3 * c, scale=hd ** -0.5)
        if c <= self.fold_linear1_max_c:
            x, st = ops.linear_stats(a, *blk["proj"], residual=x)
            h = ops.linear_ln(x, *blk["l1_ln"], stats_in=st)
        else:      # C -> 8 C from C = 384: C / 16 partial sums re-read by each of the 8 C / 16 channel tiles cost more than the LayerNorm launch, and
            # the GEMM itself is faster on the generic kernel's 32 x 32 tiles (ops.linear)
            x = ops.linear(a, *blk["proj"], residual=x)
            h = ops.linear(ops.layernorm(x, *blk["n2"]), *blk["l1"])
        h = ops.dwconv1d_k3(h.view(b, t, -1), *blk["tc"], act="gelu").view(b * t, -1)
        return ops.linear_stats(h, *blk["l2"], residual=x)

    @ops.with_latency_tiles
    def forward_btd(self, x_btd: torch.Tensor):
        """x [B,T,D] (frame-major, the feature-file layout).  Returns like the reference's forward."""
        if self.training:
            raise NotImplementedError("training path (input Dropout, network.py:76) is a later row")
        if not self._p:
            raise RuntimeError("load_state_dict first")
        p = self._p
        b, t, d = x_btd.shape
        assert d == self.D
        x = x_btd.contiguous().to(self.dtype)
        feats = []
        cin = d
        for st, c in zip(p["stages"], self.inter):
            if ops.latency_linear_ok(b * t, cin, self.dtype):      # short window: the temporal head's latency kernel (3 taps, dilation 1)
                y = ops.tcn_conv(x.view(b, t, cin), st["merge"][0], st["merge"][1], taps=3, dilation=1).view(b * t, c)
            else:
                y = ops.conv_nhwc(x.view(b, 1, t, cin), st["merge"][0], st["merge"][1], kh=1, kw=3, pad=(0, 1)).view(b * t, c)
            y = ops.layernorm(y, *st["merge_norm"])
            stats = None
            for i, blk in enumerate(st["blocks"]):
                y, stats = self._block(y, blk, b, t, c, stats)
            y = ops.layernorm(y, *st["norm"])
            feats.append(y)
            x, cin = y, c
        f1, f2, f3, f4 = feats
        e = self.E
        concat = torch.empty((b * t, 4 * e), dtype=self.dtype, device=x.device)
        _f4 = ops.linear(f4, *p["f4"])                               # interpolate to equal length == identity (TS_Mixer.py:57-64)
        concat[:, 0:e].copy_(_f4)                                    # device copy into the concat slot
        for slot, (name, f) in enumerate((("s3", f3), ("s2", f2), ("s1", f1)), start=1):
            r = ops.linear(f, *p[name + ".f"])                       # 3 * linear_f(f)
            ops.linear(_f4, *p[name + ".mix"], residual=r, out=concat[:, slot * e:(slot + 1) * e])
        feat = ops.linear(concat, *p["fuse"])                        # Classifier (network.py:104-118), dropout = identity
        y = ops.linear(feat, *p["pred"], out_dtype=torch.float32).view(b, t, -1)
        concat_ref = concat.view(b, t, 4 * e).permute(0, 2, 1)       # [B, 4E, T] view
        zeros = torch.zeros(b * t * sum(n for k, n in _K.items() if k != self.loss_type), device=x.device)      # (one fill for the unused heads)
        ys, at = {}, 0
        for k, n in _K.items():
            if k != self.loss_type:
                ys[k] = zeros[at:at + b * t * n].view(b, t, n)
                at += b * t * n
        fs = {k: concat_ref for k in _K}
        ys[self.loss_type] = y
        fs[self.loss_type] = feat.view(b, t, e).permute(0, 2, 1)
        return (ys["i"], fs["i"]), (ys["v"], fs["v"]), (ys["t"], fs["t"]), (ys["ivt"], concat_ref)

    def forward(self, inputs: torch.Tensor):
        """inputs [B,D,T] as the reference passes them (`Temporal_mstct/run.py:149-152`); transposed once to rows."""
        return self.forward_btd(inputs.permute(0, 2, 1))

    __call__ = forward
