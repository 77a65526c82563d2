"""One training step of the MS-TCT temporal teacher on MI355X: forward (train mode), BCE-with-logits, backward and SGD, as
`Temporal_mstct/run.py:147-235` does with torch autograd over `Temporal_mstct/network.py:75-101` -- here as explicit HIP launches.

* every nn.Linear / Conv1d: the implicit-GEMM kernel forward (`mt4_conv_nhwc`), the same kernel with transposed weights for the data
  gradient, `mt4_wgrad_conv1d_f32` + `mt4_colsum_f32` for the parameter gradients;
* Global_Relational_Block (`Temporal_Encoder.py:76-88`): q.k^T, softmax, P.v and their four backward products through the strided batched
  GEMM `mt4_bgemm_f32` on head slices of the packed q / kv buffers (no permutes), `mt4_softmax_rows_f32` / `mt4_softmax_bwd_rows_f32`;
  the probabilities P [B,8,T,T] are kept for the backward;
* nn.LayerNorm: `mt4_layernorm` / `mt4_layernorm_bwd_f32`; Local_Relational_Block (`:34-43`): `mt4_dwconv1d_k3` (+GELU) and
  `mt4_gelu_bwd_f32`, `mt4_dwconv1d_k3_bwd_f32`;
* Temporal_Mixer (`TS_Mixer.py:50-84`): at equal lengths `interpolate` is the identity and the three 1x1 convs that meet in each scale's
  output apply to the same input, so the forward uses their summed weight (rebuilt on the device every step) and each of the three
  receives the same gradient -- exactly what autograd gives;
* loss `mt4_bce_logits_pw_f32` (pos_weight for i / v / t, none for ivt: `run.py:331-334`), optimizer `mt4_sgd_step_f32` on ONE flat
  parameter buffer, so DDP is one all-reduce of the flat gradient buffer over RCCL (windows shard over ranks).

The two nn.Dropout(0.5) of `network.py:58,76,108,113` are drawn on the host (`draw_masks`) or passed in, so that parity tests feed the
oracle the same draw.  float32 throughout.
"""
from __future__ import annotations

from typing import Dict, List, Optional, Sequence

import torch

from . import ops
from .shapes import mstct_shapes
from .tenco_train import allreduce_sum_flat

F32 = torch.float32
NCLS = {"i": 6, "v": 10, "t": 15, "ivt": 100}
# `Temporal_mstct/run.py:322-328` (the class weights of the CholecT45 label statistics; same tables as Spatial_cnn/run.py:306-311)
POS_W = {"i": [0.93487068, 0.94234964, 0.93487068, 1.18448115, 1.02368339, 0.97974447],
         "v": [0.60002400, 0.60002400, 0.60002400, 0.61682467, 0.67082683, 0.80163207, 0.70562823, 2.11208448, 2.69230769, 0.60062402],
         "t": [0.49752894, 0.52041527, 0.49752894, 0.51394739, 2.71899565, 1.75577963, 0.58509403, 1.25228034, 0.49752894, 2.42993134,
               0.49802647, 0.87266576, 1.36074165, 0.50150917, 0.49802647],
         "ivt": None}


class _Lin:
    """nn.Linear / Conv1d parameter pair with its gradient views and the transposed copy for the data gradient"""
    __slots__ = ("name", "cout", "cout_real", "cin", "taps", "w", "b", "gw", "gb", "wt", "kpad", "conv_shape", "w16", "wt16")


class _Vec:
    __slots__ = ("name", "p", "g", "shape")


def _r4(n: int) -> int:
    return (n + 3) // 4 * 4


class MstctTrainer:
    def __init__(self, inter_channels: Sequence[int] = (256, 384, 576, 864), num_block: int = 2, head: int = 8, mlp_ratio: int = 8,
                 in_feat_dim: int = 1536, final_embedding_dim: int = 512, loss_type: str = "i", lr: float = 0.1, weight_decay: float = 1e-5,
                 device: str = "cuda", process_group=None, operand_dtype: torch.dtype = torch.float32):
        assert loss_type in NCLS
        assert operand_dtype in (torch.float32, torch.bfloat16)
        # bfloat16: the nn.Linear GEMMs (forward, data and weight gradients) read bf16 copies of their operands -- activations are cast on the
        # way in (`mt4_cast_f32_bf16`), weights re-derived from the fp32 masters after every step -- and accumulate / store in fp32; everything
        # between the GEMMs (LayerNorm, softmax, GELU, attention products, the k = 3 merge convolutions, the loss) stays fp32
        self.op16 = operand_dtype == torch.bfloat16
        self.inter, self.nb, self.H, self.ratio = tuple(inter_channels), num_block, head, mlp_ratio
        self.D, self.E, self.loss_type, self.K = in_feat_dim, final_embedding_dim, loss_type, NCLS[loss_type]
        self.KP = _r4(self.K)
        self.lr, self.wd = lr, weight_decay
        self.dev, self.pg = torch.device(device), process_group
        self.exchange = True
        self._table = mstct_shapes(in_feat_dim, self.inter, num_block, mlp_ratio, final_embedding_dim, loss_type)
        self.lins: Dict[str, _Lin] = {}
        self.vecs: Dict[str, _Vec] = {}
        self._graphs: Dict[tuple, object] = {}

    # ------------------------------------------------------------------ parameters
    def _specs(self):
        """(kind, name, ...) in flat-buffer order.  lin: (cout, cin, taps, conv_shape flag); vec: shape"""
        lin, vec = [], []
        cin = self.D
        for s, c in enumerate(self.inter, start=1):
            m = f"TemporalEncoder.Temporal_Merging_Block{s}"
            lin.append((m + ".proj", c, cin, 3, True))
            vec += [(m + ".norm.weight", (c,)), (m + ".norm.bias", (c,))]
            for b in range(self.nb):
                q = f"TemporalEncoder.block{s}.{b}"
                g, l = q + ".Global_Relational_Block", q + ".Local_Relational_Block"
                vec += [(q + ".norm1.weight", (c,)), (q + ".norm1.bias", (c,)), (q + ".norm2.weight", (c,)), (q + ".norm2.bias", (c,)),
                        (l + ".TC.weight", (self.ratio * c, 3)), (l + ".TC.bias", (self.ratio * c,))]
                lin += [(g + ".q", c, c, 1, False), (g + ".kv", 2 * c, c, 1, False), (g + ".proj", c, c, 1, False),
                        (l + ".linear1", self.ratio * c, c, 1, False), (l + ".linear2", c, self.ratio * c, 1, False)]
            vec += [(f"TemporalEncoder.norm{s}.weight", (c,)), (f"TemporalEncoder.norm{s}.bias", (c,))]
            cin = c
        for i, c in zip((4, 3, 2, 1), reversed(self.inter)):
            lin.append((f"Temporal_Mixer.linear_f{i}.proj", self.E, c, 1, False))
        for i in range(1, 10):
            lin.append((f"Temporal_Mixer.linear{i}", self.E, self.E, 1, True))
        q = f"classifier_{self.loss_type}"
        lin += [(q + ".linear_fuse", self.E, 4 * self.E, 1, True), (q + ".linear_pred", self.K, self.E, 1, True)]
        return lin, vec

    def load_state_dict(self, sd: Dict[str, torch.Tensor]):
        names = [k for k, _ in self._table]
        assert all(k in sd for k in names), "state dict incomplete"
        lin, vec = self._specs()
        total = 0
        for name, cout, cin, taps, _ in lin:
            total += _r4(cout) * ops.packed_k(cin, 1, taps, F32) + _r4(cout)
        for name, shp in vec:
            total += _r4(int(torch.tensor(shp).prod()))
        self.P = torch.zeros(total, dtype=F32, device=self.dev)
        self.G = torch.zeros(total, dtype=F32, device=self.dev)
        off = 0
        for name, cout, cin, taps, conv_shape in lin:
            c = _Lin()
            c.name, c.cout_real, c.cout, c.cin, c.taps, c.conv_shape = name, cout, _r4(cout), cin, taps, conv_shape
            c.kpad = ops.packed_k(cin, 1, taps, F32)
            nw = c.cout * c.kpad
            c.w, c.gw = self.P[off:off + nw].view(c.cout, c.kpad), self.G[off:off + nw].view(c.cout, c.kpad)
            c.b, c.gb = self.P[off + nw:off + nw + c.cout], self.G[off + nw:off + nw + c.cout]
            off += nw + c.cout
            w = sd[name + ".weight"].float()
            w3 = w.reshape(cout, cin, taps)                                   # Linear [N,K] -> [N,K,1]; Conv1d [N,K,taps]
            if c.cout != cout:
                w3 = torch.cat([w3, torch.zeros(c.cout - cout, cin, taps)], 0)
            c.w.copy_(ops.pack_conv_weight(w3.to(self.dev).unsqueeze(2), None, F32))
            c.b[:cout].copy_(sd[name + ".bias"].float().to(self.dev))
            c.wt = None if name.endswith("Block1.proj") else True          # (the first merge conv needs no data gradient; filled below)
            c.w16 = c.wt16 = None
            self.lins[name] = c
        for name, shp in vec:
            v = _Vec()
            n = int(torch.tensor(shp).prod())
            v.name, v.shape = name, shp
            v.p, v.g = self.P[off:off + n].view(*shp), self.G[off:off + n].view(*shp)
            off += _r4(n)
            src = sd[name].float()
            v.p.copy_((src[:, 0, :] if name.endswith("TC.weight") else src).to(self.dev))
            self.vecs[name] = v
        assert off == total
        E = self.E
        kp = ops.packed_k(E, 1, 1, F32)
        self._wsum = torch.zeros((3, E, kp), dtype=F32, device=self.dev)       # summed 1x1 mixer weights of scales 3, 2, 1
        self._bsum = torch.zeros((3, E), dtype=F32, device=self.dev)
        self._wsum_t = torch.zeros((3, E, ops.packed_k(E, 1, 1, F32)), dtype=F32, device=self.dev)
        # every derived matrix -- the fp32 transposed (tap-flipped) weights of the data gradients and, in the bf16-operand mode, the bf16 forward /
        # transposed copies of the eligible nn.Linear layers -- comes from ONE launch over a table after each parameter change
        self._tab = ops.RefreshTable(self.dev)
        for c in self.lins.values():
            if c.wt is not None:
                c.wt = self._tab.add(c.w, c.cout, c.cin, F32, True, [c.taps - 1 - i for i in range(c.taps)])
            if self.op16 and c.taps == 1 and c.cin % 8 == 0 and c.cout % 8 == 0 and c.cout >= 64:
                c.w16 = self._tab.add(c.w, c.cout, c.cin, torch.bfloat16, False, [0])
                if c.wt is not None:
                    c.wt16 = self._tab.add(c.w, c.cout, c.cin, torch.bfloat16, True, [0])
        self._c16: Dict[tuple, tuple] = {}
        self._dy16 = None
        self._refresh()
        return self

    _MIX = ((3, (7, 1, 4)), (2, (8, 2, 5)), (1, (9, 3, 6)))     # scale -> the linear_k that meet in its output (`TS_Mixer.py:66-79`)

    def _refresh(self):
        """derived copies after a parameter change: transposed weights for the data gradients, summed mixer weights"""
        self._tab.run()
        for j, (_, ids) in enumerate(self._MIX):
            for n, i in enumerate(ids):
                l = self.lins[f"Temporal_Mixer.linear{i}"]
                ops.axpby_(l.w, self._wsum[j], 1.0, 0.0 if n == 0 else 1.0)
                ops.axpby_(l.b, self._bsum[j], 1.0, 0.0 if n == 0 else 1.0)
            ops.transpose_pack_conv1d(self._wsum[j], self.E, self.E, 1, out=self._wsum_t[j])

    def _unpack(self, c: _Lin, packed: torch.Tensor, bias: torch.Tensor):
        w = packed[:c.cout_real, :c.taps * c.cin].reshape(c.cout_real, c.taps, c.cin).permute(0, 2, 1).contiguous().cpu()
        if not c.conv_shape:
            w = w[:, :, 0].contiguous()
        return w, bias[:c.cout_real].clone().cpu()

    def state_dict(self) -> Dict[str, torch.Tensor]:
        """reference layout and key names (`Temporal_mstct/network.py`), on the CPU"""
        out = {}
        for name, c in self.lins.items():
            out[name + ".weight"], out[name + ".bias"] = self._unpack(c, c.w, c.b)
        for name, v in self.vecs.items():
            t = v.p.clone().cpu()
            out[name] = t.unsqueeze(1) if name.endswith("TC.weight") else t
        return {k: out[k] for k, _ in self._table}

    def grads(self) -> Dict[str, torch.Tensor]:
        out = {}
        for name, c in self.lins.items():
            out[name + ".weight"], out[name + ".bias"] = self._unpack(c, c.gw, c.gb)
        for name, v in self.vecs.items():
            t = v.g.clone().cpu()
            out[name] = t.unsqueeze(1) if name.endswith("TC.weight") else t
        return out

    # ------------------------------------------------------------------ randomness
    def draw_masks(self, b: int, t: int, generator: Optional[torch.Generator] = None) -> dict:
        """the two nn.Dropout(0.5) draws as reference-shaped tensors: input [B,D,T], feat [B,E,T]; values 0 or 2"""
        g = generator
        return {"input": (torch.rand(b, self.D, t, generator=g) >= 0.5).float() * 2.0,
                "feat": (torch.rand(b, self.E, t, generator=g) >= 0.5).float() * 2.0}

    def draw_masks_device(self, b: int, t: int, seed: int, step: int) -> dict:
        """the same two draws made on the device, already in row layout ([B*T,D] / [B*T,E]): what the training driver uses (a host draw of
        31 x 1536 x 256 values and its upload cost more than the step).  Element i of stream 2*step (+1) of `synth.uniform01(seed, .)`."""
        return {"input_rows": ops.dropout_mask((b * t, self.D), seed, 2 * step, 0.5, self.dev),
                "feat_rows": ops.dropout_mask((b * t, self.E), seed, 2 * step + 1, 0.5, self.dev)}

    # ------------------------------------------------------------------ building blocks
    def _cast(self, x2d, grad=False):
        """bf16 copy of a GEMM operand.  Forward activations: made once per step and kept with their source (the copy also serves the weight
        gradient; holding the source keeps the allocator from handing its address to another tensor).  Gradients are short-lived: only the
        latest one is remembered (a layer's data and weight gradient ask for the same tensor back to back)."""
        key = (x2d.data_ptr(), tuple(x2d.shape))
        if grad:
            if self._dy16 is not None and self._dy16[0] == key:
                return self._dy16[2]
            y = ops.cast_bf16(x2d)
            self._dy16 = (key, x2d, y)
            return y
        hit = self._c16.get(key)
        if hit is None:
            hit = self._c16[key] = (x2d, ops.cast_bf16(x2d))
        return hit[1]

    def _fwd(self, x2d, c: _Lin, residual=None, act=None, out=None):
        if c.w16 is not None:
            x2d = x2d if x2d.is_contiguous() else x2d.contiguous()
            return ops.linear(self._cast(x2d), c.w16, c.b, residual=residual, act=act, out=out, out_dtype=F32)
        return ops.linear(x2d, c.w, c.b, residual=residual, act=act, out=out)

    def _dgrad(self, dy2d, c: _Lin, residual=None):
        if c.wt16 is not None:
            dy2d = dy2d if dy2d.is_contiguous() else dy2d.contiguous()
            return ops.linear(self._cast(dy2d, grad=True), c.wt16, None, residual=residual, out_dtype=F32)
        return ops.linear(dy2d, c.wt, None, residual=residual)

    def _wgrad(self, dy2d, x2d, c: _Lin):
        m = dy2d.shape[0]
        if c.w16 is not None and m % 16 == 0 and dy2d.is_contiguous() and x2d.is_contiguous():    # rows as a [M / 16, 16] pixel grid of one image
            ops.wgrad_conv2d_bf16(self._cast(dy2d, grad=True).view(1, m // 16, 16, c.cout), self._cast(x2d).view(1, m // 16, 16, c.cin), c.gw, 1, 1)
            ops.colsum(dy2d, c.gb, accumulate=True)
        else:   # (G is zeroed once per step; the bias gradient rides in the weight gradient's launch)
            ops.wgrad_conv1d(dy2d, x2d, c.gw, batch=1, t=m, taps=1, dil=1, pad=0, accumulate=True, bias_grad=c.gb)

    def _ln(self, x, name):
        return ops.layernorm(x, self.vecs[name + ".weight"].p, self.vecs[name + ".bias"].p)

    def _ln_bwd(self, dy, x, name, dx=None, accumulate=False):
        return ops.layernorm_bwd(dy, x, self.vecs[name + ".weight"].p, self.vecs[name + ".weight"].g, self.vecs[name + ".bias"].g, dx=dx,
                                 accumulate_dx=accumulate)

    def _attention_fwd(self, q, kv, b, t, c):
        hd, H = c // self.H, self.H
        P = torch.empty((b, H, t, t), dtype=F32, device=self.dev)
        ops.bgemm(q, kv, P, m=t, n=t, k=hd, nb0=H, nb1=b, a_strides=(hd, t * c, c, 1), b_strides=(hd, t * 2 * c, 1, 2 * c),
                  c_strides=(t * t, H * t * t, t, 1))
        ops.softmax_rows_(P, hd ** -0.5)
        o = torch.empty((b * t, c), dtype=F32, device=self.dev)
        ops.bgemm(P, kv[:, c:], o, m=t, n=hd, k=t, nb0=H, nb1=b, a_strides=(t * t, H * t * t, t, 1), b_strides=(hd, t * 2 * c, 2 * c, 1),
                  c_strides=(hd, t * c, c, 1))
        return P, o

    def _attention_bwd(self, do, q, kv, P, b, t, c):
        hd, H = c // self.H, self.H
        dq = torch.empty((b * t, c), dtype=F32, device=self.dev)
        dkv = torch.empty((b * t, 2 * c), dtype=F32, device=self.dev)
        sp, sq, skv = (t * t, H * t * t), (hd, t * c), (hd, t * 2 * c)
        # dV = P^T dO
        ops.bgemm(P, do, dkv[:, c:], m=t, n=hd, k=t, nb0=H, nb1=b, a_strides=sp + (1, t), b_strides=sq + (c, 1), c_strides=skv + (2 * c, 1))
        # dP = dO V^T, then dS in place
        dP = torch.empty_like(P)
        ops.bgemm(do, kv[:, c:], dP, m=t, n=t, k=hd, nb0=H, nb1=b, a_strides=sq + (c, 1), b_strides=skv + (1, 2 * c), c_strides=sp + (t, 1))
        ops.softmax_bwd_rows_(P, dP, hd ** -0.5)
        # dQ = dS K, dK = dS^T Q
        ops.bgemm(dP, kv, dq, m=t, n=hd, k=t, nb0=H, nb1=b, a_strides=sp + (t, 1), b_strides=skv + (2 * c, 1), c_strides=sq + (c, 1))
        ops.bgemm(dP, q, dkv, m=t, n=hd, k=t, nb0=H, nb1=b, a_strides=sp + (1, t), b_strides=sq + (c, 1), c_strides=skv + (2 * c, 1))
        return dq, dkv

    # ------------------------------------------------------------------ forward + backward (enqueue only)
    def _fwd_bwd(self, x_btd: torch.Tensor, z: torch.Tensor, mask_in: Optional[torch.Tensor], mask_feat: Optional[torch.Tensor]):
        """x [B,T,D] frame-major, z [B*T,K] multi-hot fp32, masks as ROWS ([B*T,D] / [B*T,E]) or None.  Returns per-column loss sums [K]."""
        b, t, d = x_btd.shape
        M, E, L = b * t, self.E, self.lins
        self.G.zero_()
        self._c16.clear()
        self._dy16 = None
        x = x_btd.contiguous().view(M, d)
        if mask_in is not None:
            x = ops.mul_add(x, mask_in)
        saved, feats = [], []
        cin = d
        for s, c in enumerate(self.inter, start=1):
            m = f"TemporalEncoder.Temporal_Merging_Block{s}"
            pm = ops.conv_nhwc(x.view(b, 1, t, cin), L[m + ".proj"].w, L[m + ".proj"].b, kh=1, kw=3, pad=(0, 1)).view(M, c)
            y = self._ln(pm, m + ".norm")
            st = dict(x_in=x, pm=pm, blocks=[])
            for bi in range(self.nb):
                q_ = f"TemporalEncoder.block{s}.{bi}"
                g_, l_ = q_ + ".Global_Relational_Block", q_ + ".Local_Relational_Block"
                n1 = self._ln(y, q_ + ".norm1")
                q = self._fwd(n1, L[g_ + ".q"])
                kv = self._fwd(n1, L[g_ + ".kv"])
                P, o = self._attention_fwd(q, kv, b, t, c)
                x_mid = self._fwd(o, L[g_ + ".proj"], residual=y)
                n2 = self._ln(x_mid, q_ + ".norm2")
                h1 = self._fwd(n2, L[l_ + ".linear1"])
                h2 = ops.dwconv1d_k3(h1.view(b, t, -1), self.vecs[l_ + ".TC.weight"].p, self.vecs[l_ + ".TC.bias"].p, act="none").view(M, -1)
                h3 = ops.gelu(h2)
                y_out = self._fwd(h3, L[l_ + ".linear2"], residual=x_mid)
                st["blocks"].append(dict(x_in=y, n1=n1, q=q, kv=kv, P=P, o=o, x_mid=x_mid, n2=n2, h1=h1, h2=h2, h3=h3))
                y = y_out
            st["pre_norm"] = y
            f = self._ln(y, f"TemporalEncoder.norm{s}")
            feats.append(f)
            saved.append(st)
            x, cin = f, c
        f1, f2, f3, f4 = feats
        concat = torch.empty((M, 4 * E), dtype=F32, device=self.dev)
        p4 = self._fwd(f4, L["Temporal_Mixer.linear_f4.proj"])
        concat[:, 0:E].copy_(p4)
        for slot, (j, (sc, _)) in enumerate(zip(range(3), self._MIX), start=1):
            r = self._fwd(feats[sc - 1], L[f"Temporal_Mixer.linear_f{sc}.proj"])
            ops.axpby_(r, r, 3.0, 0.0)                                                   # _f + (_f) + (_f): three residual copies
            ops.linear(p4, self._wsum[j], self._bsum[j], residual=r, out=concat[:, slot * E:(slot + 1) * E])
        cq = f"classifier_{self.loss_type}"
        zf = self._fwd(concat, L[cq + ".linear_fuse"])
        zd = ops.mul_add(zf, mask_feat) if mask_feat is not None else zf
        logits = self._fwd(zd, L[cq + ".linear_pred"])                                   # [M, KP]
        # ---- loss + its gradient
        dy = torch.zeros((M, self.KP), dtype=F32, device=self.dev)
        col_loss = torch.zeros(self.K, dtype=F32, device=self.dev)
        ops.bce_logits_pw(logits, z, self._pos_w, self._col_scale(M), dy, col_loss)
        # ---- backward: classifier
        self._wgrad(dy, zd, L[cq + ".linear_pred"])
        dzd = self._dgrad(dy, L[cq + ".linear_pred"])
        dzf = ops.mul_add(dzd, mask_feat) if mask_feat is not None else dzd
        self._wgrad(dzf, concat, L[cq + ".linear_fuse"])
        wt_fuse = L[cq + ".linear_fuse"].wt                                              # [4E rows (its cin)][Kpad(E)]
        dslots = [ops.linear(dzf, wt_fuse[sidx * E:(sidx + 1) * E], None) for sidx in range(4)]   # d concat, one contiguous block per slot
        # ---- backward: mixer
        dp4 = dslots[0]
        dfeat: List[Optional[torch.Tensor]] = [None, None, None, None]
        for slot, (j, (sc, ids)) in enumerate(zip(range(3), self._MIX), start=1):
            dout = dslots[slot]
            first = L[f"Temporal_Mixer.linear{ids[0]}"]
            self._wgrad(dout, p4, first)                                                 # the three 1x1 convs of this scale see the same
            for i in ids[1:]:                                                            # input and output gradient
                o_ = L[f"Temporal_Mixer.linear{i}"]
                ops.axpby_(first.gw, o_.gw, 1.0, 0.0)
                ops.axpby_(first.gb, o_.gb, 1.0, 0.0)
            dp4 = ops.linear(dout, self._wsum_t[j], None, residual=dp4)
            dr = ops.axpby_(dout, torch.empty_like(dout), 3.0, 0.0)
            lf = L[f"Temporal_Mixer.linear_f{sc}.proj"]
            self._wgrad(dr, feats[sc - 1], lf)
            dfeat[sc - 1] = self._dgrad(dr, lf)
        lf4 = L["Temporal_Mixer.linear_f4.proj"]
        self._wgrad(dp4, f4, lf4)
        dfeat[3] = self._dgrad(dp4, lf4)
        # ---- backward: encoder stages 4 .. 1
        gnext = None                                                                     # gradient arriving through the next stage's merge conv
        for s in range(4, 0, -1):
            c, st = self.inter[s - 1], saved[s - 1]
            gf = dfeat[s - 1]
            if gnext is not None:
                ops.axpby_(gnext, gf, 1.0, 1.0)
            g = self._ln_bwd(gf, st["pre_norm"], f"TemporalEncoder.norm{s}")
            for bi in range(self.nb - 1, -1, -1):
                blk = st["blocks"][bi]
                q_ = f"TemporalEncoder.block{s}.{bi}"
                g_, l_ = q_ + ".Global_Relational_Block", q_ + ".Local_Relational_Block"
                # local branch: y_out = x_mid + linear2(gelu(dwconv(linear1(LN2(x_mid)))))
                self._wgrad(g, blk["h3"], L[l_ + ".linear2"])
                dh3 = self._dgrad(g, L[l_ + ".linear2"])
                dh2 = ops.gelu_bwd(dh3, blk["h2"], out=dh3)
                dh1 = ops.dwconv1d_k3_bwd(dh2.view(b, t, -1), blk["h1"].view(b, t, -1), self.vecs[l_ + ".TC.weight"].p,
                                          self.vecs[l_ + ".TC.weight"].g, self.vecs[l_ + ".TC.bias"].g).view(M, -1)
                self._wgrad(dh1, blk["n2"], L[l_ + ".linear1"])
                dn2 = self._dgrad(dh1, L[l_ + ".linear1"])
                self._ln_bwd(dn2, blk["x_mid"], q_ + ".norm2", dx=g, accumulate=True)
                # global branch: x_mid = x_in + proj(attention(LN1(x_in)))
                self._wgrad(g, blk["o"], L[g_ + ".proj"])
                do = self._dgrad(g, L[g_ + ".proj"])
                dq, dkv = self._attention_bwd(do, blk["q"], blk["kv"], blk["P"], b, t, c)
                self._wgrad(dq, blk["n1"], L[g_ + ".q"])
                self._wgrad(dkv, blk["n1"], L[g_ + ".kv"])
                dn1 = self._dgrad(dkv, L[g_ + ".kv"], residual=self._dgrad(dq, L[g_ + ".q"]))
                self._ln_bwd(dn1, blk["x_in"], q_ + ".norm1", dx=g, accumulate=True)
            m = f"TemporalEncoder.Temporal_Merging_Block{s}"
            dpm = self._ln_bwd(g, st["pm"], m + ".norm")
            pc = L[m + ".proj"]
            xin = st["x_in"]
            ops.wgrad_conv1d(dpm.view(b, t, c), xin.view(b, t, -1), pc.gw, batch=b, t=t, taps=3, dil=1, pad=1, accumulate=True, bias_grad=pc.gb)
            gnext = ops.conv_nhwc(dpm.view(b, 1, t, c), pc.wt, None, kh=1, kw=3, pad=(0, 1)).view(M, -1) if s > 1 else None
        return col_loss

    def _col_scale(self, M: int) -> torch.Tensor:
        key = ("cs", M)
        if key not in self._graphs:
            self._graphs[key] = torch.full((self.K,), 1.0 / (M * self.K), dtype=F32, device=self.dev)
        return self._graphs[key]

    @property
    def _pos_w(self):
        if not hasattr(self, "_pw"):
            self._pw = torch.tensor(POS_W[self.loss_type], dtype=F32, device=self.dev) if POS_W[self.loss_type] is not None else None
        return self._pw

    # ------------------------------------------------------------------ one step
    def prepare_labels(self, labels: torch.Tensor) -> torch.Tensor:
        """[B,T,K] multi-hot -> fp32 [B*T,K] on the device (through pinned memory, see tenco_train.prepare_labels)"""
        z = labels.reshape(-1, self.K).to(F32).contiguous()
        return z.pin_memory().to(self.dev, non_blocking=True) if not z.is_cuda else z

    @ops.with_latency_tiles
    def train_step(self, x: torch.Tensor, labels: torch.Tensor, masks: Optional[dict] = None, apply_update: bool = True, use_graph: bool = False):
        """x [B,D,T] as the reference's loader hands it over (`dataloader.py:236-245`) or frame-major [B,T,D] (pass `frame_major=True` via
        x.is_frame_major -- see `train_step_btd`); labels [B,T,K] (or prepared [B*T,K]); masks from `draw_masks` or None.  Returns the loss."""
        return self.train_step_btd(x.permute(0, 2, 1).contiguous(), labels, masks, apply_update, use_graph)

    @ops.with_latency_tiles
    def train_step_btd(self, x_btd: torch.Tensor, labels: torch.Tensor, masks: Optional[dict] = None, apply_update: bool = True,
                       use_graph: bool = False):
        b, t, d = x_btd.shape
        assert d == self.D and x_btd.is_cuda and x_btd.dtype == F32
        z = labels if (labels.dim() == 2 and labels.is_cuda) else self.prepare_labels(labels)
        assert tuple(z.shape) == (b * t, self.K)
        rows = lambda m: m.permute(0, 2, 1).reshape(b * t, -1).contiguous().to(self.dev)     # [B,C,T] -> rows
        masks = masks or {}
        mi = masks["input_rows"] if "input_rows" in masks else rows(masks["input"]) if masks.get("input") is not None else None
        mf = masks["feat_rows"] if "feat_rows" in masks else rows(masks["feat"]) if masks.get("feat") is not None else None
        if use_graph:
            key = (b, t, mi is not None, mf is not None)
            g = self._graphs.get(key)
            ins = [x_btd, z] + ([mi] if mi is not None else []) + ([mf] if mf is not None else [])
            if g is None:
                from .graph import GraphedForward

                def fn(xx, zz, *ms):
                    ms = list(ms)
                    return self._fwd_bwd(xx, zz, ms.pop(0) if mi is not None else None, ms.pop(0) if mf is not None else None)
                g = self._graphs[key] = GraphedForward(fn, ins)
            col_loss = g(*ins)
        else:
            col_loss = self._fwd_bwd(x_btd, z, mi, mf)
        loss = float(col_loss.sum().item()) / (b * t * self.K)
        if apply_update:
            self.apply_update()
        return loss

    def apply_update(self):
        """DDP exchange (one all-reduce of the flat gradient buffer, mean over ranks) + SGD + refresh of the derived copies"""
        scale = allreduce_sum_flat(self.G, self.pg) if self.exchange else 1.0
        ops.sgd_step(self.P, self.G, self.lr, self.wd, scale)
        self._refresh()


# ------------------------------------------------------------------------------------------------ Temporal_mstct/run.py -t
def draw_windows(lengths: Dict[str, int], rng, num_clips: int = 256) -> Dict[str, int]:
    """one random `num_clips`-frame window per training video and epoch (`Temporal_mstct/dataloader.py:225-245`: the dataset of a video has
    length 1 in train mode and its item starts at `random.choice(range(0, len - num_clips))`)"""
    return {v: rng.choice(range(0, n - num_clips)) for v, n in lengths.items()}


def train_driver(argv=None):
    """`Temporal_mstct/run.py -t` (:147-235, :345-420): every epoch one random 256-frame window per training video, windows shuffled into
    batches of --batch (31 in Scripts/train_fold1.sh), SGD without momentum under LinearLR warm-up -> ExponentialLR, checkpoint
    `..._lowreslatest.pth` (no underscore: `run.py:268`) in run_<version>[_<task>].  Under torchrun every rank takes its own batch of a
    step (window-DDP, global batch = world x --batch) and the flat gradient buffer is all-reduced over RCCL once per step."""
    import argparse
    import os
    import random
    import time

    from . import cholect, featfile, shapes, synth
    from .drivers import _barrier, _common, _dist, _log
    from .tenco_train import lr_at_epoch
    p = argparse.ArgumentParser()
    _common(p)
    p.add_argument("--input_dim", type=int, default=1536)
    p.add_argument("--final_embedding_dim", type=int, default=512)
    p.add_argument("--epochs", type=int, default=100)
    p.add_argument("-w", "--warmups", type=int, nargs="+", default=[9, 18, 58])
    p.add_argument("-l", "--initial_learning_rates", type=float, nargs="+", default=[0.01, 0.01, 0.01])
    p.add_argument("--weight_decay", type=float, default=1e-5)
    p.add_argument("--decay_rate", type=float, default=0.99)
    p.add_argument("--power", type=float, default=0.1)
    p.add_argument("--val_interval", type=int, default=1)
    p.add_argument("--num_clips", type=int, default=256, help="window length (the reference hard-codes 256, dataloader.py:237)")
    p.add_argument("--operand_dtype", type=str, default="fp32", choices=["fp32", "bf16"],
                   help="bf16: the nn.Linear GEMMs on bf16 operand copies (fp32 activations, accumulation and master weights)")
    F, _ = p.parse_known_args(argv)
    rank, world = _dist()
    lt = F.loss_type
    if lt not in NCLS:
        raise ValueError("Temporal_mstct trains one task at a time: --loss_type i | v | t | ivt (Scripts/train_fold1.sh:16)")
    model_dir = f"./__checkpoint__/run_{F.version}" + ("_" + lt if lt != "all" else "")          # `run.py:88-90,131`
    modelname = f"{F.model}_l8_cholect{F.dataset_variant}_k{F.kfold}_batchnorm_lowres"
    logfile = os.path.join(model_dir, modelname + ".log")
    latest = os.path.join(model_dir, modelname + "latest.pth")
    tr = MstctTrainer((256, 384, 576, 864), 2, 8, 8, F.input_dim, F.final_embedding_dim, lt, lr=F.initial_learning_rates[2],
                      weight_decay=F.weight_decay, operand_dtype=torch.bfloat16 if F.operand_dtype == "bf16" else torch.float32)
    if os.path.exists(latest):
        tr.load_state_dict(torch.load(latest, map_location="cpu"))
    else:   # no torch.nn init here: deterministic synthetic start (the reference starts from its trunc_normal_ init)
        tr.load_state_dict(synth.fill_from_shapes(shapes.mstct_shapes(F.input_dim, (256, 384, 576, 864), 2, 8, F.final_embedding_dim, lt), seed=F.seed))
    train_videos, val_videos, _ = cholect.split_videos(F.dataset_variant, F.kfold)
    val_interval = F.epochs - 1 if F.val_interval == -1 else max(1, F.val_interval)
    best, best_path, vmodel = 0.0, os.path.join(model_dir, modelname + ".pth"), None
    feats = featfile.read_feats(featfile.feats_path("..", F.version1, F.kfold, lt))                # `dataloader.py:220-222`
    lab_name = {"i": "i", "v": "v", "t": "t", "ivt": "ivt"}[lt]
    xs, zs = {}, {}
    for v in train_videos:                                                                        # uploaded ONCE; windows are device slices
        key = featfile.video_key(v)
        if key not in feats:
            key = v[3:]                                                                           # Spatial_transformer's key style
        xs[v] = torch.from_numpy(feats[key]).to(tr.dev)
        zs[v] = torch.from_numpy(cholect.load_labels(F.data_dir, v)[lab_name][:, 1:]).to(F32).to(tr.dev)
    lengths = {v: int(xs[v].shape[0]) for v in train_videos}
    short = [v for v, n in lengths.items() if n <= F.num_clips]
    if short:
        raise ValueError(f"videos shorter than the {F.num_clips}-frame training window: {short[:3]} (the reference's sampler fails on them too)")
    order_rng, win_rng = random.Random(F.seed), random.Random(F.seed * 7919 + 1)
    for epoch in range(F.epochs):
        tr.lr = lr_at_epoch(epoch, F.initial_learning_rates[2], F.power, F.warmups[2], F.decay_rate)
        starts = draw_windows(lengths, win_rng, F.num_clips)                                       # same draw on every rank
        order = list(train_videos)
        order_rng.shuffle(order)
        batches = [order[i:i + F.batch] for i in range(0, len(order), F.batch)]                   # drop_last False
        steps = (len(batches) + world - 1) // world
        t0, tot = time.time(), 0.0
        for s in range(steps):
            vids = batches[(s * world + rank) % len(batches)]
            x = torch.stack([xs[v][starts[v]:starts[v] + F.num_clips] for v in vids])             # [B,T,D] frame-major
            z = torch.cat([zs[v][starts[v]:starts[v] + F.num_clips] for v in vids])               # [B*T,K]
            tot += tr.train_step_btd(x, z, masks=tr.draw_masks_device(len(vids), F.num_clips, F.seed + rank, epoch * steps + s))
        if rank == 0:
            _log(logfile, f"Traning | lr: {tr.lr:.6f} | epoch {epoch} | loss {tot / steps:.4f} | {time.time() - t0:.2f} secs")
            os.makedirs(model_dir, exist_ok=True)
            state = tr.state_dict()
            torch.save(state, latest + ".tmp")
            os.replace(latest + ".tmp", latest)
            if epoch % val_interval == 0:                          # validation + `weight_mgt` (`run.py:416-452,265-277`): best `.pth` by the task's mAP
                from .drivers import _chlg, _mstct_scores
                from .metrics import recognition_from
                t1 = time.time()
                if vmodel is None:
                    from .temporal_mstct import VideoNas
                    vmodel = VideoNas(F, [256, 384, 576, 864], 2, 8, 8, F.input_dim, F.final_embedding_dim).eval()
                vmodel.load_state_dict(state)
                vm = recognition_from(_mstct_scores(vmodel, feats, val_videos, F.data_dir, lt), val_videos) if val_videos else None
                score = float(vm[lt].compute_video_AP(ignore_null=_chlg(F))["mAP"]) if vm else 0.0
                if score > best or not os.path.exists(best_path):
                    best = max(best, score)
                    torch.save(state, best_path + ".tmp")
                    os.replace(best_path + ".tmp", best_path)
                    _log(logfile, f">>> Saving checkpoint for epoch {epoch + 1} at {best_path}, time {time.ctime()} ")
                _log(logfile, f"\t\t\t\t\t\t\t video-wise | eta {time.time() - t1:.2f} secs | mAP => {lt}: [{score:.5f}] ")
    _barrier()
    return tr
