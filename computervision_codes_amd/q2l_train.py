"""One training step of the Swin + Query2Label teacher on MI355X: forward (train mode), BCE-with-logits(pos_weight), backward and SGD, as
`Spatial_transformer/run.py:150-229` does with torch autograd over `Spatial_transformer/network.py:82-128` -- here as explicit HIP launches.
Single-task teachers (`--loss_type i|v|t`: what `Scripts/train_fold1.sh:12-14` trains): one `Decoder` over the backbone, the loss is that
head's BCE alone (`run.py:168-182`).  `--loss_type all` (`run.py:183-197`, the Res -> Swin direction of MT4MTL-KD): four `Decoder`s over ONE
shared `Transformer` (`network.py:66-73`: its parameters exist once, their gradient is the sum over the four passes), the always-on KD mixing
on decoder_ivt's pooled memory (`network.py:98-124`; `mt4_kd_mix` / `mt4_kd_mix_bwd_f32`), loss = rates[0] * (4 x BCE) + rates[1] * mean
DistillKL + rates[2] * mean MSE (`mt4_distill_kl_f32`, `mt4_mse_f32`).

* every nn.Linear / 1x1 conv / the 4x4 patch-embedding conv: `mt4_conv_nhwc` forward, the same kernel with transposed weights for data
  gradients, `mt4_wgrad_conv1d_f32` + `mt4_colsum_f32` for parameter gradients;
* `SwinTransformerBlock` (`swin_transformer.py:234-271`): roll + window_partition and their inverses are row gathers / scatters
  (`mt4_gather_rows_f32`), `WindowAttention` (`:114-145`) runs as strided batched GEMMs on head slices of the packed qkv buffer
  (`mt4_bgemm_f32`), the relative-position table is read through its index (`mt4_add_bias_mask_f32`) and receives its gradient by the
  transposed scatter (`mt4_relpos_table_grad_f32`), softmax / LayerNorm / GELU forward and backward as in the MS-TCT trainer, DropPath
  per sample (`mt4_rowscale_add_f32`); `PatchMerging` (`:308-329`) = gather(group 4) + LayerNorm + GEMM;
* the Q2L transformer (`models/transformer.py:95-113,160-196,260-302`; one post-norm encoder layer, two decoder layers without self-attention,
  `nn.MultiheadAttention` with 4 heads) with its nn.Dropout(0.1) masks, `GroupWiseLinear` (`network.py:40-45`);
* loss `mt4_bce_logits_pw_f32`, optimizer `mt4_sgd_step_f32` on one flat parameter buffer (DDP: one all-reduce of the flat gradient buffer).

Randomness (DropPath per block and sample, the transformer's dropouts) is passed in as masks (`draw_masks` for reference-shaped host draws,
`draw_masks_device` for training), so that parity tests feed the oracle the same draw.  float32 throughout.
"""
from __future__ import annotations

from typing import Dict, List, Optional

import torch

from . import ops
from .shapes import SWIN_CFG, q2l_param_shapes, swin_window
from .spatial_transformer import _merge_row_map, _rel_pos_index, _shift_mask, _window_row_map, sine_position_rows
from .synth import IMAGENET_MEAN, IMAGENET_STD
from .spatial_cnn_train import TARGET_W, TOOL_W, VERB_W
from .tenco_train import allreduce_sum_flat

F32 = torch.float32
NCLS = {"i": 6, "v": 10, "t": 15, "ivt": 100}
TASKS_ALL = ("i", "v", "t", "ivt")
POS_W = {"i": TOOL_W, "v": VERB_W, "t": TARGET_W}                                   # `Spatial_transformer/run.py:312-316,339-341`
NHEAD, FFN = 4, 8192                                                               # `transformer.py:347-359` (build_transformer)


def _r4(n: int) -> int:
    return (n + 3) // 4 * 4


class _Lin:
    """a GEMM weight (packed) + bias with gradient views; `wt` = transposed packed copy for the data gradient"""
    __slots__ = ("name", "wkey", "bkey", "cout", "cin", "w", "b", "gw", "gb", "wt", "shape", "w16", "wt16")


class _Vec:
    __slots__ = ("key", "p", "g", "shape")


class FlatParams:
    """all trained tensors in ONE flat parameter buffer P and ONE flat gradient buffer G (packed GEMM layouts), plus the derived transposed
    copies; state-dict in / out in the reference's key names and shapes"""

    def __init__(self, device, op16: bool = False):
        self.dev, self.op16 = device, op16
        self.tab = ops.RefreshTable(device)          # every derived matrix (transposed fp32, bf16 copies) from one launch per refresh
        self._lin: List[tuple] = []
        self._vec: List[tuple] = []
        self.L: Dict[str, _Lin] = {}
        self.V: Dict[str, _Vec] = {}
        self._slices: List[tuple] = []

    def lin(self, name, wkey, bkey, cout, cin, need_dgrad=True):
        self._lin.append((name, wkey, bkey, cout, cin, need_dgrad))

    def vec(self, key, shape):
        self._vec.append((key, tuple(shape)))

    def build(self, sd):
        total = sum(c * ops.packed_k(ci, 1, 1, F32) + _r4(c) for _, _, _, c, ci, _ in self._lin) + sum(_r4(int(torch.tensor(s).prod())) for _, s in self._vec)
        self.P = torch.zeros(total, dtype=F32, device=self.dev)
        self.G = torch.zeros(total, dtype=F32, device=self.dev)
        off = 0
        for name, wkey, bkey, cout, cin, need_dgrad in self._lin:
            assert cout % 4 == 0 and cin % 4 == 0, (name, cout, cin)
            l = _Lin()
            l.name, l.wkey, l.bkey, l.cout, l.cin = name, wkey, bkey, cout, cin
            kp = ops.packed_k(cin, 1, 1, F32)
            n = cout * kp
            l.w, l.gw = self.P[off:off + n].view(cout, kp), self.G[off:off + n].view(cout, kp)
            l.b, l.gb = (self.P[off + n:off + n + cout], self.G[off + n:off + n + cout]) if bkey else (None, None)
            off += n + _r4(cout)
            w = sd[wkey].float()
            l.shape = tuple(w.shape)
            l.w.copy_(ops.pack_linear_weight(w.reshape(cout, cin).to(self.dev), F32))
            if bkey:
                l.b.copy_(sd[bkey].float().to(self.dev))
            self._derive(l, need_dgrad)
            self.L[name] = l
        for key, shape in self._vec:
            v = _Vec()
            n = int(torch.tensor(shape).prod())
            v.key, v.shape = key, tuple(sd[key].shape)
            v.p, v.g = self.P[off:off + n].view(*shape), self.G[off:off + n].view(*shape)
            off += _r4(n)
            v.p.copy_(sd[key].float().reshape(shape).to(self.dev))
            self.V[key] = v
        assert off == total
        self.refresh()
        return self

    def _derive(self, l: _Lin, need_dgrad: bool):
        """the matrices the kernels read besides the master weight: the transposed copy for the data gradient (fp32) and, in the bf16-operand
        mode, bf16 copies of both for the GEMMs whose channel counts allow it"""
        l.wt = self.tab.add(l.w, l.cout, l.cin, F32, True, [0]) if need_dgrad else None
        l.w16 = l.wt16 = None
        if self.op16 and l.cin % 8 == 0 and l.cout % 8 == 0 and l.cout >= 64:
            l.w16 = self.tab.add(l.w, l.cout, l.cin, torch.bfloat16, False, [0])
            if need_dgrad:
                l.wt16 = self.tab.add(l.w, l.cout, l.cin, torch.bfloat16, True, [0])

    def rows(self, name: str, lo: int, hi: int) -> _Lin:
        """a row slice of a packed weight (nn.MultiheadAttention's in_proj split into q / k / v) with its own transposed copy"""
        src = self.L[name]
        l = _Lin()
        l.name, l.cout, l.cin = f"{name}[{lo}:{hi}]", hi - lo, src.cin
        l.w, l.gw, l.b, l.gb = src.w[lo:hi], src.gw[lo:hi], src.b[lo:hi], src.gb[lo:hi]
        self._derive(l, True)
        self._slices.append(l)
        return l

    def refresh(self):
        self.tab.run()

    def _export(self, which: str) -> Dict[str, torch.Tensor]:
        out = {}
        for l in self.L.values():
            wsrc, bsrc = (l.w, l.b) if which == "p" else (l.gw, l.gb)
            out[l.wkey] = wsrc[:, :l.cin].reshape(l.shape).clone().cpu()
            if l.bkey:
                out[l.bkey] = bsrc.clone().cpu()
        for v in self.V.values():
            out[v.key] = (v.p if which == "p" else v.g).reshape(v.shape).clone().cpu()
        return out


class Q2LTrainer:
    def __init__(self, backbone: str = "swin_L_384_22k", img_size: int = 384, hidden_dim: int = 1536, loss_type: str = "i", lr: float = 0.01,
                 weight_decay: float = 1e-5, drop_path_rate: float = 0.1, device: str = "cuda", process_group=None,
                 operand_dtype: torch.dtype = torch.float32, teacher_dim: int = 512, rates=(1.0, 0.0, 0.1), temp: float = 4.0):
        if loss_type not in ("i", "v", "t", "all"):
            raise ValueError("loss_type: i | v | t | all (Spatial_transformer/run.py:168-197)")
        self.backbone, self.S, self.d, self.loss_type = backbone, int(img_size), int(hidden_dim), loss_type
        self.tasks = TASKS_ALL if loss_type == "all" else (loss_type,)
        self.task, self.K = self.tasks[0], NCLS[self.tasks[0]]          # (single-task mode: the one decoder)
        self.teacher_dim, self.rates, self.temp = int(teacher_dim), tuple(float(r) for r in rates), float(temp)   # `run.py:66,70` (--rates default 1 0 0.1)
        self.cfg = SWIN_CFG[backbone]
        assert self.d == self.cfg["embed_dim"] * 8, "hidden_dim is the backbone's final width (backbone.py:188-201)"
        self.lr, self.wd, self.dev, self.pg = lr, weight_decay, torch.device(device), process_group
        assert operand_dtype in (torch.float32, torch.bfloat16)
        # bfloat16: the nn.Linear / patch-embedding GEMMs (forward, data and weight gradients) read bf16 copies of their operands; activations,
        # accumulation, everything between the GEMMs and the master weights stay fp32 (as in `MstctTrainer`)
        self.op16 = operand_dtype == torch.bfloat16
        self._c16: Dict[tuple, tuple] = {}
        self._dy16 = None
        self.exchange = True
        self._table = q2l_param_shapes(backbone, self.S, self.d, loss_type, teacher_dim=self.teacher_dim)
        nblk = sum(self.cfg["depths"])
        self.drop_probs = [drop_path_rate * i / max(1, nblk - 1) for i in range(nblk)]     # `swin_transformer.py:517` (linspace 0 .. rate)

    # ------------------------------------------------------------------ parameters
    def load_state_dict(self, sd: Dict[str, torch.Tensor]):
        assert all(k in sd for k, _ in self._table), "state dict incomplete"
        fp, pre, C0, d = FlatParams(self.dev, self.op16), "backbone.0.", self.cfg["embed_dim"], self.d
        fp.lin("pe", pre + "patch_embed.proj.weight", pre + "patch_embed.proj.bias", C0, 48, need_dgrad=False)
        fp.vec(pre + "patch_embed.norm.weight", (C0,)); fp.vec(pre + "patch_embed.norm.bias", (C0,))
        self.stages = []
        for s, (depth, nh) in enumerate(zip(self.cfg["depths"], self.cfg["num_heads"])):
            ws, res = swin_window(self.backbone, self.S, s)
            C = C0 * 2 ** s
            blocks = []
            for b in range(depth):
                q = f"{pre}layers.{s}.blocks.{b}."
                shift = 0 if (b % 2 == 0 or res <= self.cfg["window_size"]) else self.cfg["window_size"] // 2
                for n_ in ("norm1", "norm2"):
                    fp.vec(q + n_ + ".weight", (C,)); fp.vec(q + n_ + ".bias", (C,))
                fp.vec(q + "attn.relative_position_bias_table", ((2 * ws - 1) ** 2, nh))
                fp.lin(q + "qkv", q + "attn.qkv.weight", q + "attn.qkv.bias", 3 * C, C)
                fp.lin(q + "proj", q + "attn.proj.weight", q + "attn.proj.bias", C, C)
                fp.lin(q + "fc1", q + "mlp.fc1.weight", q + "mlp.fc1.bias", 4 * C, C)
                fp.lin(q + "fc2", q + "mlp.fc2.weight", q + "mlp.fc2.bias", C, 4 * C)
                blocks.append(dict(q=q, shift=shift, row_map=_window_row_map(res, ws, shift).to(self.dev),
                                   mask=_shift_mask(res, ws, shift).to(self.dev) if shift > 0 else None))
            st = dict(res=res, ws=ws, nh=nh, C=C, blocks=blocks, idx=_rel_pos_index(ws).reshape(-1).to(torch.int32).to(self.dev))
            if s < 3:
                q = f"{pre}layers.{s}.downsample."
                fp.vec(q + "norm.weight", (4 * C,)); fp.vec(q + "norm.bias", (4 * C,))
                fp.lin(q + "red", q + "reduction.weight", None, 2 * C, 4 * C)
                st["merge"] = dict(q=q, row_map=_merge_row_map(res).to(self.dev))
            self.stages.append(st)
        fp.vec(pre + "norm.weight", (d,)); fp.vec(pre + "norm.bias", (d,))
        for task in self.tasks:
            dq, K = f"decoder_{task}.", NCLS[task]
            fp.lin("in_proj." + task, dq + "input_proj.weight", dq + "input_proj.bias", d, d)
            fp.vec(dq + "query_embed.weight", (K, d)); fp.vec(dq + "fc.W", (K, d)); fp.vec(dq + "fc.b", (K,))
        if self.loss_type == "all":              # KD adaptors (`network.py:75-80`): Conv1d(k = 1) on [B, C, 1] = linear layers
            for n in ("wi", "wv", "wt"):
                fp.lin(n, n + ".weight", n + ".bias", self.teacher_dim, d)
            for n in ("mi", "mv", "mt"):
                fp.lin(n, n + ".weight", n + ".bias", d, self.teacher_dim, need_dgrad=False)
        t = self.tprefix = f"decoder_{self.tasks[0]}.transformer."     # the ONE Transformer (`named_parameters()` lists it under its first owner)
        layers = [("enc", t + "encoder.layers.0", "self_attn", ("norm1", "norm2"))] + \
                 [(f"dec{i}", f"{t}decoder.layers.{i}", "multihead_attn", ("norm2", "norm3")) for i in range(2)]
        for tag, lp, att, norms in layers:
            fp.lin(tag + ".in", f"{lp}.{att}.in_proj_weight", f"{lp}.{att}.in_proj_bias", 3 * d, d, need_dgrad=False)
            fp.lin(tag + ".out", f"{lp}.{att}.out_proj.weight", f"{lp}.{att}.out_proj.bias", d, d)
            fp.lin(tag + ".l1", lp + ".linear1.weight", lp + ".linear1.bias", FFN, d)
            fp.lin(tag + ".l2", lp + ".linear2.weight", lp + ".linear2.bias", d, FFN)
            for n_ in norms:
                fp.vec(f"{lp}.{n_}.weight", (d,)); fp.vec(f"{lp}.{n_}.bias", (d,))
        fp.vec(t + "decoder.norm.weight", (d,)); fp.vec(t + "decoder.norm.bias", (d,))
        fp.build(sd)
        self.fp, self.P, self.G = fp, fp.P, fp.G
        self.att = {tag: tuple(fp.rows(tag + ".in", i * d, (i + 1) * d) for i in range(3)) for tag, *_ in layers}
        fp.refresh()                                   # (the q / k / v row slices joined the table after build())
        self.layer_prefix = {tag: lp for tag, lp, _, _ in layers}
        self.layer_norms = {tag: norms for tag, _, _, norms in layers}
        hh = self.S // 32
        self.pos = sine_position_rows(d, hh, hh).to(self.dev, F32)
        self.pos_w = {t: torch.tensor(POS_W[t], dtype=F32, device=self.dev) for t in self.tasks if t in POS_W}     # ivt: plain BCE (`run.py:342`)
        self._graphs: Dict[tuple, object] = {}
        return self

    def state_dict(self) -> Dict[str, torch.Tensor]:
        """the reference's parameter names and shapes; loss_type all: followed by the shared transformer's three alias entries per tensor
        (decoder_v / _t / _ivt.transformer.*), as the reference module's own `state_dict()` lists them (`run.py:266-277` saves that)"""
        out = self.fp._export("p")
        sd = {k: out[k] for k, _ in self._table}
        if self.loss_type == "all":
            from .shapes import q2l_state_dict_aliases
            for alias, src in q2l_state_dict_aliases(self.d):
                sd[alias] = sd[src]
        return sd

    def grads(self) -> Dict[str, torch.Tensor]:
        return self.fp._export("g")

    # ------------------------------------------------------------------ randomness
    def mask_specs(self, b: int):
        """(key, shape) of every nn.Dropout(0.1) draw of the Q2L transformer in a step, row layout (rows image-major)"""
        L, d = (self.S // 32) ** 2, self.d
        specs = []
        for task in self.tasks:               # loss_type all: every decoder pass draws its own masks, keys 'i/enc.attn' ... 'ivt/dec1.d3'
            K, pk = NCLS[task], (task + "/" if self.loss_type == "all" else "")
            specs += [(pk + "enc.attn", (b, NHEAD, L, L)), (pk + "enc.d1", (b * L, d)), (pk + "enc.ffn", (b * L, FFN)), (pk + "enc.d2", (b * L, d))]
            for i in range(2):
                specs += [(f"{pk}dec{i}.attn", (b, NHEAD, K, L)), (f"{pk}dec{i}.d2", (b * K, d)), (f"{pk}dec{i}.ffn", (b * K, FFN)),
                          (f"{pk}dec{i}.d3", (b * K, d))]
        return specs

    def draw_masks(self, b: int, generator: Optional[torch.Generator] = None) -> dict:
        """host draw (tests): DropPath keep / keep_prob per block and sample, nn.Dropout(0.1) masks of the transformer"""
        g = generator
        dp = []
        for p in self.drop_probs:
            keep = 1.0 - p
            dp.append(tuple((torch.rand(b, generator=g) < keep).float() / keep for _ in range(2)))
        tx = {k: (torch.rand(*shp, generator=g) >= 0.1).float() / 0.9 for k, shp in self.mask_specs(b)}
        return {"droppath": dp, "tx": tx}

    def draw_masks_device(self, b: int, seed: int, step: int) -> dict:
        """the same draws on the device from the counter generator (`mt4_dropout_mask_f32`)"""
        stream = step * 4096
        dp = []
        for i, p in enumerate(self.drop_probs):
            dp.append(tuple(ops.dropout_mask((b,), seed, stream + 2 * i + j, p, self.dev) for j in range(2)))
        tx = {k: ops.dropout_mask(shp, seed, stream + 2048 + n, 0.1, self.dev) for n, (k, shp) in enumerate(self.mask_specs(b))}
        return {"droppath": dp, "tx": tx}

    # ------------------------------------------------------------------ building blocks
    def _cast(self, x2d, grad=False):
        """bf16 copy of a GEMM operand: forward activations once per step, kept with their source (the copy also serves the weight gradient, and
        holding the source keeps the allocator from reusing its address); of the short-lived gradients only the latest is remembered"""
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

    def _fwd(self, x, l: _Lin, residual=None, act=None, out_row_map=None):
        if l.w16 is not None:
            x = x if x.is_contiguous() else x.contiguous()
            return ops.linear(self._cast(x), l.w16, l.b, residual=residual, act=act, out_row_map=out_row_map, out_dtype=F32)
        return ops.linear(x, l.w, l.b, residual=residual, act=act, out_row_map=out_row_map)

    def _bwd(self, dy, x, l: _Lin, need_dx=True, residual=None, gate=None):
        """parameter gradients of y = x W^T + b from dy; returns dx (+ residual; gate: ReLU of the layer below, `act=relu_gate`)"""
        m = dy.shape[0]
        mixed = l.w16 is not None and m % 16 == 0
        if mixed:
            dy = dy if dy.is_contiguous() else dy.contiguous()
            x = x if x.is_contiguous() else x.contiguous()
            dy16 = self._cast(dy, grad=True)
            ops.wgrad_conv2d_bf16(dy16.view(1, m // 16, 16, l.cout), self._cast(x).view(1, m // 16, 16, l.cin), l.gw, 1, 1)   # (adds to gw)
            if l.gb is not None:
                ops.colsum(dy, l.gb, accumulate=True)
        else:   # (the bias gradient rides in the weight gradient's launch)
            ops.wgrad_conv1d(dy, x, l.gw, batch=1, t=m, taps=1, dil=1, pad=0, accumulate=True, bias_grad=l.gb)
        if not need_dx:
            return None
        if mixed and l.wt16 is not None:
            if gate is not None:
                return ops.linear(dy16, l.wt16, None, residual=gate, act="relu_gate", out_dtype=F32)
            return ops.linear(dy16, l.wt16, None, residual=residual, out_dtype=F32)
        if gate is not None:
            return ops.linear(dy, l.wt, None, residual=gate, act="relu_gate")
        return ops.linear(dy, l.wt, None, residual=residual)

    def _ln(self, x, key):
        return ops.layernorm(x, self.fp.V[key + ".weight"].p, self.fp.V[key + ".bias"].p)

    def _ln_bwd(self, dy, x, key, dx=None, accumulate=False):
        V = self.fp.V
        return ops.layernorm_bwd(dy, x, V[key + ".weight"].p, V[key + ".weight"].g, V[key + ".bias"].g, dx=dx, accumulate_dx=accumulate)

    def _attn_fwd(self, q, k, v, nb, nh, nq, nk, sq, sk, sv, dout, scale, bias=None, index=None, mask=None, drop=None):
        """softmax(scale q k^T [+ bias + mask]) v on head slices: q rows [nb*nq] of pitch sq, k / v rows [nb*nk] of pitch sk / sv, head h at
        column h*hd.  Returns (P kept for the backward, P after dropout or None, output [nb*nq, dout])"""
        hd = dout // nh
        P = torch.empty((nb, nh, nq, nk), dtype=F32, device=self.dev)
        sp = (nq * nk, nh * nq * nk)
        ops.bgemm(q, k, P, m=nq, n=nk, k=hd, nb0=nh, nb1=nb, a_strides=(hd, nq * sq, sq, 1), b_strides=(hd, nk * sk, 1, sk), c_strides=sp + (nk, 1),
                  alpha=scale)
        if bias is not None:
            ops.add_bias_mask_(P, bias, mask, index)
        ops.softmax_rows_(P, 1.0)
        Pd = ops.mul_add(P, drop) if drop is not None else None
        o = torch.empty((nb * nq, dout), dtype=F32, device=self.dev)
        ops.bgemm(Pd if Pd is not None else P, v, o, m=nq, n=hd, k=nk, nb0=nh, nb1=nb, a_strides=sp + (nk, 1), b_strides=(hd, nk * sv, sv, 1),
                  c_strides=(hd, nq * dout, dout, 1))
        return P, Pd, o

    def _attn_bwd(self, do, q, k, v, P, Pd, drop, dq, dk, dv, nb, nh, nq, nk, sq, sk, sv, dout, scale, sdq, sdk, sdv):
        """gradients into dq / dk / dv (views with row pitches sdq / sdk / sdv); returns dS (for the relative-position table)"""
        hd = dout // nh
        sp = (nq * nk, nh * nq * nk)
        so = (hd, nq * dout)
        ops.bgemm(Pd if Pd is not None else P, do, dv, m=nk, n=hd, k=nq, nb0=nh, nb1=nb, a_strides=sp + (1, nk), b_strides=so + (dout, 1),
                  c_strides=(hd, nk * sdv, sdv, 1))                                                     # dV = P^T dO
        dP = torch.empty_like(P)
        ops.bgemm(do, v, dP, m=nq, n=nk, k=hd, nb0=nh, nb1=nb, a_strides=so + (dout, 1), b_strides=(hd, nk * sv, 1, sv), c_strides=sp + (nk, 1))
        if drop is not None:
            dP = ops.mul_add(dP, drop)
        ops.softmax_bwd_rows_(P, dP, 1.0)                                                               # dS
        ops.bgemm(dP, k, dq, m=nq, n=hd, k=nk, nb0=nh, nb1=nb, a_strides=sp + (nk, 1), b_strides=(hd, nk * sk, sk, 1), c_strides=(hd, nq * sdq, sdq, 1),
                  alpha=scale)                                                                          # dQ = scale dS K
        ops.bgemm(dP, q, dk, m=nk, n=hd, k=nq, nb0=nh, nb1=nb, a_strides=sp + (1, nk), b_strides=(hd, nq * sq, sq, 1), c_strides=(hd, nk * sdk, sdk, 1),
                  alpha=scale)                                                                          # dK = scale dS^T Q
        return dP

    # ------------------------------------------------------------------ Swin block
    def _block_fwd(self, x, st, blk, B, dp):
        C, nh, ws, res = st["C"], st["nh"], st["ws"], st["res"]
        L, N, nwin, q_ = res * res, ws * ws, (res // ws) ** 2, blk["q"]
        fp = self.fp
        xn = self._ln(x, q_ + "norm1")
        xw = ops.gather_rows(xn, blk["row_map"], l_out=L, l_in=L)
        qkv = self._fwd(xw, fp.L[q_ + "qkv"])
        P, _, a = self._attn_fwd(qkv, qkv[:, C:], qkv[:, 2 * C:], B * nwin, nh, N, N, 3 * C, 3 * C, 3 * C, C, (C // nh) ** -0.5,
                                 bias=fp.V[q_ + "attn.relative_position_bias_table"].p, index=st["idx"], mask=blk["mask"])
        if dp is None:
            x1 = self._fwd(a, fp.L[q_ + "proj"], residual=x, out_row_map=blk["row_map"])
        else:
            oc = ops.scatter_rows(self._fwd(a, fp.L[q_ + "proj"]), blk["row_map"], l_out=L, l_in=L)
            x1 = ops.rowscale_add(oc, dp[0], x, L)
        y = self._ln(x1, q_ + "norm2")
        h1 = self._fwd(y, fp.L[q_ + "fc1"])
        h2 = ops.gelu(h1)
        x2 = self._fwd(h2, fp.L[q_ + "fc2"], residual=x1) if dp is None else ops.rowscale_add(self._fwd(h2, fp.L[q_ + "fc2"]), dp[1], x1, L)
        return x2, dict(x=x, xw=xw, qkv=qkv, P=P, a=a, x1=x1, y=y, h1=h1, h2=h2)

    def _block_bwd(self, g, sv, st, blk, B, dp):
        """g = gradient w.r.t. the block's output (owned, updated in place to the gradient w.r.t. its input)"""
        C, nh, ws, res = st["C"], st["nh"], st["ws"], st["res"]
        L, N, nwin, q_ = res * res, ws * ws, (res // ws) ** 2, blk["q"]
        fp = self.fp
        d2 = g if dp is None else ops.rowscale_add(g, dp[1], None, L)
        dh2 = self._bwd(d2, sv["h2"], fp.L[q_ + "fc2"])
        dh1 = ops.gelu_bwd(dh2, sv["h1"], out=dh2)
        dy = self._bwd(dh1, sv["y"], fp.L[q_ + "fc1"])
        self._ln_bwd(dy, sv["x1"], q_ + "norm2", dx=g, accumulate=True)                      # g = d x1
        d1 = g if dp is None else ops.rowscale_add(g, dp[0], None, L)
        do = ops.gather_rows(d1, blk["row_map"], l_out=L, l_in=L)                            # window order
        da = self._bwd(do, sv["a"], fp.L[q_ + "proj"])
        qkv = sv["qkv"]
        dqkv = torch.empty_like(qkv)
        dS = self._attn_bwd(da, qkv, qkv[:, C:], qkv[:, 2 * C:], sv["P"], None, None, dqkv, dqkv[:, C:], dqkv[:, 2 * C:], B * nwin, nh, N, N,
                            3 * C, 3 * C, 3 * C, C, (C // nh) ** -0.5, 3 * C, 3 * C, 3 * C)
        ops.relpos_table_grad(dS, st["idx"], fp.V[q_ + "attn.relative_position_bias_table"].g)
        dxw = self._bwd(dqkv, sv["xw"], fp.L[q_ + "qkv"])
        dxn = ops.scatter_rows(dxw, blk["row_map"], l_out=L, l_in=L)
        self._ln_bwd(dxn, sv["x"], q_ + "norm1", dx=g, accumulate=True)                      # g = d x
        return g

    # ------------------------------------------------------------------ Q2L transformer layer pieces
    def _mha_fwd(self, tag, q_in, k_in, v_in, B, nq, nk, drop):
        lq, lk, lv = self.att[tag]
        q, k, v = self._fwd(q_in, lq), self._fwd(k_in, lk), self._fwd(v_in, lv)
        d = self.d
        P, Pd, a = self._attn_fwd(q, k, v, B, NHEAD, nq, nk, d, d, d, d, (d // NHEAD) ** -0.5, drop=drop)
        o = self._fwd(a, self.fp.L[tag + ".out"])
        return o, dict(q_in=q_in, k_in=k_in, v_in=v_in, q=q, k=k, v=v, P=P, Pd=Pd, a=a, drop=drop)

    def _mha_bwd(self, tag, do, sv, B, nq, nk):
        """returns (d q_in, d k_in, d v_in)"""
        lq, lk, lv = self.att[tag]
        d = self.d
        da = self._bwd(do, sv["a"], self.fp.L[tag + ".out"])
        dq, dk, dv = torch.empty_like(sv["q"]), torch.empty_like(sv["k"]), torch.empty_like(sv["v"])
        self._attn_bwd(da, sv["q"], sv["k"], sv["v"], sv["P"], sv["Pd"], sv["drop"], dq, dk, dv, B, NHEAD, nq, nk, d, d, d, d, (d // NHEAD) ** -0.5, d, d, d)
        return self._bwd(dq, sv["q_in"], lq), self._bwd(dk, sv["k_in"], lk), self._bwd(dv, sv["v_in"], lv)

    def _ffn_fwd(self, tag, x, m_ffn, m_out):
        fp = self.fp
        f = self._fwd(x, fp.L[tag + ".l1"], act="relu")
        fd = ops.mul_add(f, m_ffn) if m_ffn is not None else f
        o = self._fwd(fd, fp.L[tag + ".l2"])
        if m_out is not None:
            o = ops.mul_add(o, m_out)
        return o, dict(x=x, f=f, fd=fd)

    def _ffn_bwd(self, tag, do, sv, m_ffn, m_out, residual):
        """d x (+ residual) of x -> linear2(drop(relu(linear1 x)))"""
        fp = self.fp
        if m_out is not None:
            do = ops.mul_add(do, m_out)
        dfd = self._bwd(do, sv["fd"], fp.L[tag + ".l2"], gate=sv["f"])        # ReLU gate fused: (f > 0) ? grad : 0
        if m_ffn is not None:
            dfd = ops.mul_add(dfd, m_ffn)
        return self._bwd(dfd, sv["x"], fp.L[tag + ".l1"], residual=residual)

    # ------------------------------------------------------------------ forward + backward (enqueue only)
    def _backbone_fwd(self, img, B, dps):
        fp, pre = self.fp, "backbone.0."
        rows = ops.patchify(img, 4, F32, IMAGENET_MEAN, IMAGENET_STD)
        pe = self._fwd(rows, fp.L["pe"])
        x = self._ln(pe, pre + "patch_embed.norm")
        saved, bi_all = [], 0
        for st in self.stages:
            sts = dict(blocks=[])
            for blk in st["blocks"]:
                dp = dps[bi_all] if dps is not None else None
                x, sv = self._block_fwd(x, st, blk, B, dp)
                sts["blocks"].append((sv, dp))
                bi_all += 1
            if "merge" in st:
                res, C, mg = st["res"], st["C"], st["merge"]
                xm = ops.gather_rows(x, mg["row_map"], l_out=(res // 2) ** 2, l_in=res * res, group=4, m_out=B * (res // 2) ** 2)
                xmn = self._ln(xm, mg["q"] + "norm")
                sts.update(xm=xm, xmn=xmn, m_in=x.shape[0])
                x = self._fwd(xmn, fp.L[mg["q"] + "red"])
            saved.append(sts)
        feats = self._ln(x, pre + "norm")                                                    # [B*L, d] rows of the reference's [B,d,h,h]
        return feats, dict(rows=rows, pe=pe, saved=saved, x_last=x)

    def _backbone_bwd(self, dfeats, bb, B):
        fp, pre = self.fp, "backbone.0."
        g = self._ln_bwd(dfeats, bb["x_last"], pre + "norm")
        for si in range(len(self.stages) - 1, -1, -1):
            st, sts = self.stages[si], bb["saved"][si]
            if "merge" in st:
                mg, res = st["merge"], st["res"]
                dxmn = self._bwd(g, sts["xmn"], fp.L[mg["q"] + "red"])
                dxm = self._ln_bwd(dxmn, sts["xm"], mg["q"] + "norm")
                g = ops.scatter_rows(dxm, mg["row_map"], l_out=(res // 2) ** 2, l_in=res * res, group=4, m_in=sts["m_in"])
            for blk, (sv, dp) in zip(reversed(st["blocks"]), reversed(sts["blocks"])):
                g = self._block_bwd(g, sv, st, blk, B, dp)
        dpe = self._ln_bwd(g, bb["pe"], pre + "patch_embed.norm")
        self._bwd(dpe, bb["rows"], fp.L["pe"], need_dx=False)

    def _decoder_fwd(self, task, feats, B, tm):
        """`Decoder.forward` (`network.py:163-171`) of one task over the shared transformer: logits [B, K]; everything the backward needs"""
        fp, d, K = self.fp, self.d, NCLS[task]
        L = feats.shape[0] // B
        s0 = self._fwd(feats, fp.L["in_proj." + task])
        sp = ops.add_rowbcast(s0, self.pos)
        o, enc_att = self._mha_fwd("enc", sp, sp, s0, B, L, L, tm("enc.attn"))
        if tm("enc.d1") is not None:
            o = ops.mul_add(o, tm("enc.d1"))
        lp = self.layer_prefix["enc"]
        u1 = ops.axpby_(s0, o, 1.0, 1.0)
        s1 = self._ln(u1, lp + ".norm1")
        f2, enc_ffn = self._ffn_fwd("enc", s1, tm("enc.ffn"), tm("enc.d2"))
        u2 = ops.axpby_(s1, f2, 1.0, 1.0)
        memory = self._ln(u2, lp + ".norm2")
        mem_pos = ops.add_rowbcast(memory, self.pos)
        query = fp.V[f"decoder_{task}.query_embed.weight"]
        tgt = torch.zeros((B * K, d), dtype=F32, device=self.dev)
        dec_saved = []
        for i in range(2):
            tag, lpd = f"dec{i}", self.layer_prefix[f"dec{i}"]
            qin = ops.add_rowbcast(tgt, query.p)
            o, att = self._mha_fwd(tag, qin, mem_pos, memory, B, K, L, tm(tag + ".attn"))
            if tm(tag + ".d2") is not None:
                o = ops.mul_add(o, tm(tag + ".d2"))
            ua = ops.axpby_(tgt, o, 1.0, 1.0)
            t1 = self._ln(ua, lpd + ".norm2")
            f2, ffn = self._ffn_fwd(tag, t1, tm(tag + ".ffn"), tm(tag + ".d3"))
            ub = ops.axpby_(t1, f2, 1.0, 1.0)
            tgt = self._ln(ub, lpd + ".norm3")
            dec_saved.append(dict(att=att, ua=ua, ffn=ffn, ub=ub))
        hs = self._ln(tgt, self.tprefix + "decoder.norm")
        Wk, bk = fp.V[f"decoder_{task}.fc.W"], fp.V[f"decoder_{task}.fc.b"]
        logits = ops.groupwise_linear(hs, Wk.p, bk.p, B, K)                                   # [B, K]
        return logits, dict(feats=feats, L=L, enc_att=enc_att, u1=u1, enc_ffn=enc_ffn, u2=u2, memory=memory, dec=dec_saved, tgt=tgt, hs=hs)

    def _decoder_bwd(self, task, dy, sv, B, tm, dmem_extra=None):
        """gradients of one decoder pass from dy = dL/dlogits [B, K] (+ dmem_extra = dL/dmemory rows from the KD branch): parameter gradients
        are ADDED (the shared transformer collects all four passes); returns dL/dfeats"""
        fp, K, L = self.fp, NCLS[task], sv["L"]
        query = fp.V[f"decoder_{task}.query_embed.weight"]
        Wk, bk = fp.V[f"decoder_{task}.fc.W"], fp.V[f"decoder_{task}.fc.b"]
        dhs = ops.groupwise_linear_bwd(dy, sv["hs"], Wk.p, Wk.g, bk.g)
        dtgt = self._ln_bwd(dhs, sv["tgt"], self.tprefix + "decoder.norm")
        dmem = dmem_extra                                                                     # gradient w.r.t. the encoder memory
        for i in (1, 0):
            tag, lpd, ds = f"dec{i}", self.layer_prefix[f"dec{i}"], sv["dec"][i]
            dub = self._ln_bwd(dtgt, ds["ub"], lpd + ".norm3")
            dt1 = self._ffn_bwd(tag, dub, ds["ffn"], tm(tag + ".ffn"), tm(tag + ".d3"), residual=dub)
            dua = self._ln_bwd(dt1, ds["ua"], lpd + ".norm2")
            do = ops.mul_add(dua, tm(tag + ".d2")) if tm(tag + ".d2") is not None else dua
            dqin, dkin, dvin = self._mha_bwd(tag, do, ds["att"], B, K, L)
            ops.sum_over_batch(dqin, query.g, B, accumulate=True)                             # query embedding: added to every image's queries
            dtgt = ops.axpby_(dua, dqin, 1.0, 1.0)                                            # previous tgt: residual path + through q
            dm = ops.axpby_(dkin, dvin, 1.0, 1.0)                                             # memory: through k (memory + pos) and v
            dmem = dm if dmem is None else ops.axpby_(dm, dmem, 1.0, 1.0)
        lp = self.layer_prefix["enc"]
        du2 = self._ln_bwd(dmem, sv["u2"], lp + ".norm2")
        ds1 = self._ffn_bwd("enc", du2, sv["enc_ffn"], tm("enc.ffn"), tm("enc.d2"), residual=du2)
        du1 = self._ln_bwd(ds1, sv["u1"], lp + ".norm1")
        do = ops.mul_add(du1, tm("enc.d1")) if tm("enc.d1") is not None else du1
        dq_in, dk_in, dv_in = self._mha_bwd("enc", do, sv["enc_att"], B, L, L)
        ds0 = ops.axpby_(du1, dv_in, 1.0, 1.0)
        ops.axpby_(dq_in, ds0, 1.0, 1.0)
        ops.axpby_(dk_in, ds0, 1.0, 1.0)
        return self._bwd(ds0, sv["feats"], fp.L["in_proj." + task])

    def _fwd_bwd(self, img: torch.Tensor, z: torch.Tensor, masks: Optional[dict], tp=None, tf=None):
        """img uint8 NHWC or normalised float32 NCHW; z [B, sum K] multi-hot fp32 (the tasks' label blocks side by side); loss_type all: tp = 3 x
        raw teacher logits, tf = 3 x teacher features [B, teacher_dim].  Returns (per-column BCE sums [sum K], soft sum [1], kd sum [1])."""
        fp, B, d = self.fp, img.shape[0], self.d
        masks = masks or {}
        tx = masks.get("tx") or {}
        allm = self.loss_type == "all"
        self.G.zero_()
        self._c16.clear()
        self._dy16 = None
        feats, bb = self._backbone_fwd(img, B, masks.get("droppath"))
        # ---- decoders forward
        tms = {t: (lambda k, pk=(t + "/" if allm else ""): tx.get(pk + k)) for t in self.tasks}
        logits, svs = {}, {}
        for t in self.tasks:
            logits[t], svs[t] = self._decoder_fwd(t, feats, B, tms[t])
        self.last_logits = logits
        # ---- losses + their gradients (`run.py:164-197`)
        r0, r1, r2 = self.rates if allm else (1.0, 0.0, 0.0)
        col_loss = torch.zeros(z.shape[1], dtype=F32, device=self.dev)
        soft = torch.zeros(1, dtype=F32, device=self.dev)
        kdl = torch.zeros(1, dtype=F32, device=self.dev)
        dys, o = {}, 0
        for t in self.tasks:
            K = NCLS[t]
            dys[t] = torch.zeros((B, K), dtype=F32, device=self.dev)
            ops.bce_logits_pw(logits[t], z[:, o:o + K].contiguous(), self.pos_w.get(t), self._col_scale(B, K, r0), dys[t], col_loss[o:o + K])
            o += K
        dmem_kd = None
        if allm:
            for t, tpn in zip(("i", "v", "t"), tp):
                ops.distill_kl(logits[t], tpn, dys[t], soft, self.temp, r1 / 3.0, accumulate=True)
            # KD mixing on decoder_ivt's pooled memory (`network.py:96-124`): feat [B, d]
            mem, L = svs["ivt"]["memory"], svs["ivt"]["L"]
            feat = ops.global_avgpool(mem.view(B, L, 1, d))
            self.last_feat = feat
            teas = [self._fwd(tn, fp.L[m]) for m, tn in zip(("mi", "mv", "mt"), tf)]
            mixed = ops.kd_mix(feat, *teas)
            cams = [self._fwd(mx, fp.L[w]) for w, mx in zip(("wi", "wv", "wt"), mixed)]
            self.last_cams = cams
            dcams = [ops.mse(c, tn, kdl, r2 / 3.0) for c, tn in zip(cams, tf)]
            gs = [self._bwd(dc, mx, fp.L[w]) for w, dc, mx in zip(("wi", "wv", "wt"), dcams, mixed)]
            ds_kd, dtau = ops.kd_mix_bwd(feat, teas, gs)
            for n, (m, tn) in enumerate(zip(("mi", "mv", "mt"), tf)):
                dte = dtau[:, n:n + 1].expand(B, d).contiguous()                              # d(tea_n)[b][:] = dtau[b][n]
                self._bwd(dte, tn, fp.L[m], need_dx=False)
            dmem_kd = ops.avgpool_bwd(ds_kd, B, L, d).view(B * L, d)
        # ---- decoders backward -> dL/dfeats (summed over the passes), backbone backward
        dfeats = None
        for t in reversed(self.tasks):
            df = self._decoder_bwd(t, dys[t], svs[t], B, tms[t], dmem_extra=dmem_kd if t == "ivt" else None)
            dfeats = df if dfeats is None else ops.axpby_(df, dfeats, 1.0, 1.0)
            svs[t] = None
        self._backbone_bwd(dfeats, bb, B)
        return col_loss, soft, kdl

    def _col_scale(self, B: int, K: int, r0: float = 1.0) -> torch.Tensor:
        key = ("cs", B, K, r0)
        if key not in self._graphs:
            self._graphs[key] = torch.full((K,), r0 / (B * K), dtype=F32, device=self.dev)
        return self._graphs[key]

    # ------------------------------------------------------------------ one step
    def _prep_masks(self, masks: Optional[dict]) -> Optional[dict]:
        if not masks:
            return None
        out = {}
        if masks.get("droppath") is not None:
            out["droppath"] = [tuple(m.to(self.dev, F32).contiguous() for m in pair) for pair in masks["droppath"]]
        if masks.get("tx"):
            out["tx"] = {k: v.to(self.dev, F32).contiguous() for k, v in masks["tx"].items()}
        return out

    @ops.with_latency_tiles
    def train_step(self, img: torch.Tensor, labels, masks: Optional[dict] = None, apply_update: bool = True, teacher_pred=None, teacher_feat=None):
        """img: uint8 NHWC [B,S,S,3] or normalised float32 NCHW [B,3,S,S] on the GPU.  Single task: labels [B,K] multi-hot of the task, returns
        the loss.  loss_type all: labels (y_i, y_v, y_t, y_ivt), teacher_pred 3 x raw logits [B,K], teacher_feat 3 x [B,teacher_dim]
        (`run.py:152-157`), returns the dict of loss terms (`run.py:199-214`)."""
        B = img.shape[0]
        if self.loss_type != "all":
            z = labels.to(self.dev, F32).contiguous()
            assert img.is_cuda and tuple(z.shape) == (B, self.K)
            col_loss, _, _ = self._fwd_bwd(img, z, self._prep_masks(masks))
            loss = float(col_loss.sum().item()) / (B * self.K)
            if apply_update:
                self.apply_update()
            return loss
        z = torch.cat([l.to(self.dev, F32) for l in labels], 1).contiguous()
        tp = [t.to(self.dev, F32).contiguous() for t in teacher_pred]
        tf = [t.to(self.dev, F32).contiguous() for t in teacher_feat]
        assert img.is_cuda and tuple(z.shape) == (B, sum(NCLS[t] for t in self.tasks)) and all(t.shape == (B, self.teacher_dim) for t in tf)
        col_loss, soft, kdl = self._fwd_bwd(img, z, self._prep_masks(masks), tp, tf)
        cl, sk = col_loss.cpu(), torch.cat([soft, kdl]).cpu()
        r0, r1, r2 = self.rates
        terms, o, hard = {}, 0, 0.0
        for t in self.tasks:                         # (the kernels scale the GRADIENT by the rates; the sums they return are unscaled)
            K = NCLS[t]
            terms["hard_" + t] = float(cl[o:o + K].sum() / (B * K))
            hard += terms["hard_" + t]
            o += K
        terms.update(hard=hard, soft=float(sk[0]) / 3.0, kd=float(sk[1]) / 3.0)
        terms["loss"] = r0 * terms["hard"] + r1 * terms["soft"] + r2 * terms["kd"]
        if apply_update:
            self.apply_update()
        return terms

    def apply_update(self):
        scale = allreduce_sum_flat(self.G, self.pg) if self.exchange else 1.0
        ops.sgd_step(self.P, self.G, self.lr, self.wd, scale)
        self.fp.refresh()
