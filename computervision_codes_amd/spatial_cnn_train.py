"""One training step of the spatial student on MI355X (`Spatial_cnn/run.py:145-224`): train-mode ResNet-18/50 trunk
(BatchNorm on batch statistics), KD branch (`network.py:47-71`), hard BCE (pos_weight) + soft DistillKL + feature MSE
(`run.py:159-192,284-295,322-328`), backward and SGD (`:342-351`) as explicit HIP launches -- no autograd.  fp32, NHWC.

* conv forward / data gradient: `mt4_conv_nhwc` (data gradient = conv of dY with the transposed, tap-reversed weights;
  stride-2 3x3 convs as 4 sub-pixel phases and the stride-2 1x1 downsample as one phase, scattered through the output row
  map into dX);  weight gradient: `mt4_wgrad_conv2d_f32` / `mt4_wgrad_conv1d_f32`;
* BatchNorm: `mt4_bn_stats_f32` + `mt4_bn_apply_f32` forward, `mt4_bn_backward_f32` (ReLU gate and residual fan-out fused);
* losses: `mt4_bce_logits_pw_f32`, `mt4_distill_kl_f32`, `mt4_mse_f32`; KD mixing `mt4_kd_mix` / `mt4_kd_mix_bwd_f32`;
* one flat fp32 parameter buffer and one flat gradient buffer (packed layouts) -> ONE all-reduce per step for DDP, one
  `mt4_sgd_step_f32`.  BatchNorm statistics stay per GPU, like the reference's plain `BatchNorm2d`.
"""
from __future__ import annotations

import os
from typing import Dict, List, Sequence

import torch

from . import ops
from .shapes import resnet_feat_dim, spatial_cnn_shapes
from .tenco_train import allreduce_sum_flat

_DEPTHS = {"resnet18": (2, 2, 2, 2), "resnet50": (3, 4, 6, 3)}
_ALL_HEADS = (("i", 6), ("v", 10), ("t", 15), ("ivt", 100))
# `Spatial_cnn/run.py:306-311`
TOOL_W = [0.93487068, 0.94234964, 0.93487068, 1.18448115, 1.02368339, 0.97974447]
VERB_W = [0.60002400, 0.60002400, 0.60002400, 0.61682467, 0.67082683, 0.80163207, 0.70562823, 2.11208448, 2.69230769, 0.60062402]
TARGET_W = [0.49752894, 0.52041527, 0.49752894, 0.51394739, 2.71899565, 1.75577963, 0.58509403, 1.25228034, 0.49752894, 2.42993134,
            0.49802647, 0.87266576, 1.36074165, 0.50150917, 0.49802647]
F32 = torch.float32
_POS_W = {"i": TOOL_W, "v": VERB_W, "t": TARGET_W, "ivt": [1.0] * 100}


class _Unit:
    """conv + BatchNorm (+ReLU) with its parameter / gradient views"""
    __slots__ = ("name", "bn", "cin", "cout", "k", "stride", "pad", "w", "gw", "gamma", "beta", "ggamma", "gbeta", "rmean", "rvar", "wt", "phase_w",
                 "sums_f", "sums_b", "w16", "wt16", "phase_w16")


class SpatialCnnTrainer:
    op16 = False                # (class defaults: GEMM operands fp32; no derived-weight table built yet)
    _refresh_table = None
    def __init__(self, network: str = "resnet50", lr: float = 0.01, weight_decay: float = 1e-5, rates: Sequence[float] = (1.0, 1.0, 1.0),
                 temp: float = 4.0, device: str = "cuda", process_group=None, overlap: bool = True, teacher_dim: int = 1536,
                 loss_type: str = "all", operand_dtype: torch.dtype = torch.float32, epilogue_stats: bool = True):
        """loss_type 'all': the distillation recipe (four heads, KD branch, hard + soft + feature losses: `run.py:180-192`);
        'i' | 'v' | 't': a single-task student -- only that classifier exists (`network.py:34-41`) and the loss is its BCE alone
        (`run.py:165-179`)"""
        assert loss_type in ("all", "i", "v", "t")
        assert operand_dtype in (torch.float32, torch.bfloat16)
        # bfloat16: the convolutions' GEMM operands (activations, activation gradients, weight copies) are bf16, sums fp32 / fp64, master
        # weights + gradients + SGD fp32 (csrc/train2d_bf16.hip); the stem's 7x7x3 convolution and the heads / KD branch stay fp32
        self.op16 = operand_dtype == torch.bfloat16
        # BatchNorm statistics from the producing convolution's epilogue (False: the separate pass over the map -- same bits, tested)
        self.epilogue_stats = bool(epilogue_stats)
        self.loss_type = loss_type
        self.heads = _ALL_HEADS if loss_type == "all" else tuple(h for h in _ALL_HEADS if h[0] == loss_type)
        self.NH = sum(k for _, k in self.heads)
        self.NHP = (self.NH + 3) // 4 * 4
        self.network, self.lr, self.wd, self.rates, self.temp = network, lr, weight_decay, tuple(rates), float(temp)
        self.overlap = overlap            # DDP: all-reduce each gradient bucket as soon as the backward has written it (eager steps)
        self._pending: list = []
        self._capturing = False
        self.dev, self.pg = torch.device(device), process_group
        self.C = resnet_feat_dim(network)
        self.TD = int(teacher_dim)          # `--teacher_dim` (`Spatial_cnn/run.py:82`): width of the teachers' frame features
        self._table = spatial_cnn_shapes(network, self.C, self.TD, loss_type)
        self.units: Dict[str, _Unit] = {}
        self.nbt: Dict[str, int] = {}
        self._extra: Dict[str, torch.Tensor] = {}

    # ------------------------------------------------------------------ parameters
    def _unit_specs(self):
        pre = "basemodel.basemodel."
        specs = [(pre + "conv1", pre + "bn1", 4, 64, 7, 2, 0)]   # stem on the physically padded 4-channel image
        bott = self.network == "resnet50"
        cin = 64
        for li, (planes, n) in enumerate(zip((64, 128, 256, 512), _DEPTHS[self.network]), start=1):
            for b in range(n):
                s = 2 if (b == 0 and li > 1) else 1
                q = f"{pre}layer{li}.{b}."
                cout = planes * (4 if bott else 1)
                if bott:
                    specs += [(q + "conv1", q + "bn1", cin, planes, 1, 1, 0), (q + "conv2", q + "bn2", planes, planes, 3, s, 1),
                              (q + "conv3", q + "bn3", planes, cout, 1, 1, 0)]
                else:
                    specs += [(q + "conv1", q + "bn1", cin, planes, 3, s, 1), (q + "conv2", q + "bn2", planes, planes, 3, 1, 1)]
                if b == 0 and (s != 1 or cin != cout):
                    specs.append((q + "downsample.0", q + "downsample.1", cin, cout, 1, s, 0))
                cin = cout
        return specs

    def load_state_dict(self, sd: Dict[str, torch.Tensor]):
        assert all(k in sd for k, _ in self._table), "state dict incomplete"
        dev, C = self.dev, self.C
        specs = self._unit_specs()
        TD = self.TD
        NH, NHP, _HEADS = self.NH, self.NHP, self.heads
        lin = [("heads", NHP, C)]
        if self.loss_type == "all":
            lin += [("wi", TD, C), ("wv", TD, C), ("wt", TD, C), ("mi", C, TD), ("mv", C, TD), ("mt", C, TD)]
        r4 = lambda n: (n + 3) // 4 * 4
        total = sum(co * ops.packed_k(ci, k, k, F32) + 2 * r4(co) for _, _, ci, co, k, _, _ in specs)
        total += sum(co * ops.packed_k(ci, 1, 1, F32) + r4(co) for _, co, ci in lin)
        self.P, self.G = torch.zeros(total, dtype=F32, device=dev), torch.zeros(total, dtype=F32, device=dev)
        off = 0

        def take(n, shape=None):
            nonlocal off
            p, g = self.P[off:off + n], self.G[off:off + n]
            off += r4(n)
            return (p.view(shape), g.view(shape)) if shape else (p, g)

        trained = set()
        self._ranges: Dict[str, list] = {}     # flat-buffer range of every gradient bucket: "stem", "layer1".."layer4", "heads"
        for conv, bn, ci, co, k, s, pad in specs:
            bucket = conv.split(".")[2] if ".layer" in conv else "stem"
            self._ranges.setdefault(bucket, [off, off])
            u = _Unit()
            u.name, u.bn, u.cin, u.cout, u.k, u.stride, u.pad = conv, bn, ci, co, k, s, pad
            kp = ops.packed_k(ci, k, k, F32)
            u.w, u.gw = take(co * kp, (co, kp))
            u.gamma, u.ggamma = take(co)
            u.beta, u.gbeta = take(co)
            w = sd[conv + ".weight"].float().to(dev)
            if ci == 4 and k == 7:
                w = torch.cat([w, torch.zeros(co, 1, 7, 7, device=dev)], 1)
            u.w.copy_(ops.pack_conv_weight(w, None, F32))
            u.gamma.copy_(sd[bn + ".weight"].float())
            u.beta.copy_(sd[bn + ".bias"].float())
            u.rmean, u.rvar = sd[bn + ".running_mean"].float().to(dev).clone(), sd[bn + ".running_var"].float().to(dev).clone()
            self.nbt[bn] = int(sd[bn + ".num_batches_tracked"])
            u.wt, u.phase_w = None, None
            u.w16, u.wt16, u.phase_w16 = None, None, None
            self.units[conv] = u
            self._ranges[bucket][1] = off
            trained |= {conv + ".weight", bn + ".weight", bn + ".bias", bn + ".running_mean", bn + ".running_var", bn + ".num_batches_tracked"}
        self.lin: Dict[str, tuple] = {}
        self._ranges["heads"] = [off, total]
        for name, co, ci in lin:
            kp = ops.packed_k(ci, 1, 1, F32)
            w, gw = take(co * kp, (co, kp))
            b, gb = take(co)
            if name == "heads":
                wsrc = torch.cat([sd[f"classifier_{t}.fc.weight"].float() for t, _ in _HEADS] + [torch.zeros(NHP - NH, C)], 0)
                bsrc = torch.cat([sd[f"classifier_{t}.fc.bias"].float() for t, _ in _HEADS] + [torch.zeros(NHP - NH)], 0)
                trained |= {f"classifier_{t}.fc.{p}" for t, _ in _HEADS for p in ("weight", "bias")}
            else:
                wsrc, bsrc = sd[name + ".weight"].float()[:, :, 0], sd[name + ".bias"].float()
                trained |= {name + ".weight", name + ".bias"}
            w.copy_(ops.pack_linear_weight(wsrc.to(dev), F32))
            b.copy_(bsrc.to(dev))
            self.lin[name] = (w, b, gw, gb, co, ci)
        assert off == total
        # float64 scratch of every BatchNorm reduction of a step (forward 2C -- or its STAT_REPLICAS copies when the convolution's epilogue
        # fills them -- and backward 2C), zeroed once per step
        rep = ops.STAT_REPLICAS if self.epilogue_stats else 1
        self._arena = torch.zeros(sum((2 * rep + 2) * u.cout for u in self.units.values()), dtype=torch.float64, device=dev)
        o = 0
        for u in self.units.values():
            u.sums_f, u.sums_b = self._arena[o:o + 2 * rep * u.cout], self._arena[o + 2 * rep * u.cout:o + (2 * rep + 2) * u.cout]
            o += (2 * rep + 2) * u.cout
        self._row_maps: Dict[tuple, torch.Tensor] = {}
        self._col_scales: Dict[int, torch.Tensor] = {}
        self._graphs: Dict[tuple, object] = {}
        self._extra = {k: sd[k].detach().clone() for k, _ in self._table if k not in trained}   # the trunk's unused 1000-way fc
        self.pos_weight = torch.tensor(sum((_POS_W[t] for t, _ in self.heads), []), dtype=F32, device=dev)
        self._refresh_table = None
        self._refresh_transposed()
        return self

    def _refresh_transposed(self):
        """every weight matrix derived from the master weights -- data-gradient operators (transposed with flipped taps; the four sub-pixel phase
        kernels of a stride-2 3x3), in bf16 mode also the bf16 copies the forward convolutions read -- rebuilt by ONE launch over a table
        (`mt4_refresh_weights`; a launch per matrix was ~350 launches = 1.5 ms per step).  Destinations are allocated once: captured graphs keep
        their addresses."""
        if getattr(self, "_refresh_table", None) is None:
            dt = torch.bfloat16 if self.op16 else F32
            tab = ops.RefreshTable(self.dev)
            for u in self.units.values():
                if u.cin == 4:
                    continue   # stem: the image needs no gradient; it stays fp32
                taps = u.k * u.k
                if self.op16:
                    u.w16 = tab.add(u.w, u.cout, u.cin, dt, False, range(taps), (u.k, u.k))
                if u.stride == 1 or u.k == 1:
                    wt = tab.add(u.w, u.cout, u.cin, dt, True, [taps - 1 - t for t in range(taps)], (u.k, u.k))
                    if self.op16:
                        u.wt16 = wt
                    else:
                        u.wt = wt
                else:   # 3x3 stride 2 pad 1: sub-pixel phases, phase parity -> original taps in the order of the data-gradient offsets 0, +1
                    sel = {0: [1], 1: [2, 0]}
                    ph_w = {}
                    for ph in (0, 1):
                        for pw in (0, 1):
                            khs, kws = len(sel[ph]), len(sel[pw])
                            dst = tab.add(u.w, u.cout, u.cin, dt, True, [sel[ph][a_] * 3 + sel[pw][b_] for a_ in range(khs) for b_ in range(kws)], (khs, kws))
                            ph_w[(ph, pw)] = (dst, khs, kws)
                    if self.op16:
                        u.phase_w16 = ph_w
                    else:
                        u.phase_w = ph_w
            self._refresh_table = tab
        self._refresh_table.run()
        # (the linear layers' data gradients transpose their small weights on the fly in _linear_bwd)

    def running_stats(self) -> Dict[str, torch.Tensor]:
        """the BatchNorm buffers only (what a train-mode forward changes)"""
        out = {}
        for u in self.units.values():
            out[u.bn + ".running_mean"], out[u.bn + ".running_var"] = u.rmean.clone().cpu(), u.rvar.clone().cpu()
            out[u.bn + ".num_batches_tracked"] = torch.tensor(self.nbt[u.bn], dtype=torch.int64)
        return out

    def state_dict(self) -> Dict[str, torch.Tensor]:
        out = dict(self._extra)
        for u in self.units.values():
            taps = u.k * u.k
            tapw = (u.cin + 3) // 4 * 4
            w = u.w[:, :taps * tapw].reshape(u.cout, u.k, u.k, tapw)[..., :u.cin].permute(0, 3, 1, 2).contiguous().cpu()
            if u.cin == 4:
                w = w[:, :3].contiguous()
            out[u.name + ".weight"] = w
            out[u.bn + ".weight"], out[u.bn + ".bias"] = u.gamma.clone().cpu(), u.beta.clone().cpu()
            out[u.bn + ".running_mean"], out[u.bn + ".running_var"] = u.rmean.clone().cpu(), u.rvar.clone().cpu()
            out[u.bn + ".num_batches_tracked"] = torch.tensor(self.nbt[u.bn], dtype=torch.int64)
        for name, (w, b, gw, gb, co, ci) in self.lin.items():
            ww, bb = w[:, :ci].clone().cpu(), b.clone().cpu()
            if name == "heads":
                o = 0
                for t, k in self.heads:
                    out[f"classifier_{t}.fc.weight"], out[f"classifier_{t}.fc.bias"] = ww[o:o + k].clone(), bb[o:o + k].clone()
                    o += k
            else:
                out[name + ".weight"], out[name + ".bias"] = ww.unsqueeze(-1), bb
        return {k: out[k] for k, _ in self._table}

    def grads(self) -> Dict[str, torch.Tensor]:
        out = {}
        for u in self.units.values():
            taps, tapw = u.k * u.k, (u.cin + 3) // 4 * 4
            g = u.gw[:, :taps * tapw].reshape(u.cout, u.k, u.k, tapw)[..., :u.cin].permute(0, 3, 1, 2).contiguous().cpu()
            out[u.name + ".weight"] = g[:, :3].contiguous() if u.cin == 4 else g
            out[u.bn + ".weight"], out[u.bn + ".bias"] = u.ggamma.clone().cpu(), u.gbeta.clone().cpu()
        for name, (w, b, gw, gb, co, ci) in self.lin.items():
            gg, gbb = gw[:, :ci].clone().cpu(), gb.clone().cpu()
            if name == "heads":
                o = 0
                for t, k in self.heads:
                    out[f"classifier_{t}.fc.weight"], out[f"classifier_{t}.fc.bias"] = gg[o:o + k].clone(), gbb[o:o + k].clone()
                    o += k
            else:
                out[name + ".weight"], out[name + ".bias"] = gg.unsqueeze(-1), gbb
        return out

    # ------------------------------------------------------------------ building blocks
    def _fwd_unit(self, u: _Unit, x, residual=None, relu=True, saved=None):
        w = u.w16 if (self.op16 and u.cin != 4) else u.w            # (bf16 mode: the stem reads the fp32 image with its fp32 weights)
        if self.epilogue_stats:
            # the BatchNorm statistics ride in the convolution's epilogue, mean / invstd are evaluated inside the apply launch: no pass over z
            z = ops.conv_nhwc(x, w, None, kh=u.k, kw=u.k, stride=(u.stride, u.stride), pad=(u.pad, u.pad), stat_sums=u.sums_f)
            b, ho, wo, c = z.shape
            apply = ops.bn_apply_sums_t if self.op16 else ops.bn_apply_sums
            a, mean, invstd = apply(z.view(-1, c), u.sums_f, u.gamma, u.beta, residual.view(-1, c) if residual is not None else None, relu, u.rmean, u.rvar)
            a = a.view(b, ho, wo, c)
            if saved is not None:
                saved.append((u, x, z, mean, invstd, a, relu, residual is not None))
            return a
        z = ops.conv_nhwc(x, w, None, kh=u.k, kw=u.k, stride=(u.stride, u.stride), pad=(u.pad, u.pad))
        b, ho, wo, c = z.shape
        z2 = z.view(-1, c)
        if self.op16:
            mean, invstd = ops.bn_stats_t(z2, u.rmean, u.rvar, sums=u.sums_f)
            a = ops.bn_apply_t(z2, mean, invstd, u.gamma, u.beta, residual.view(-1, c) if residual is not None else None, relu).view(b, ho, wo, c)
        else:
            mean, invstd = ops.bn_stats(z2, u.rmean, u.rvar, sums=u.sums_f)
            a = ops.bn_apply(z2, mean, invstd, u.gamma, u.beta, residual.view(-1, c) if residual is not None else None, relu).view(b, ho, wo, c)
        if saved is not None:
            saved.append((u, x, z, mean, invstd, a, relu, residual is not None))
        return a

    def _dgrad(self, u: _Unit, dz, x_shape, residual=None):
        """gradient w.r.t. the unit's input [B,H,W,Cin] from dz [B,Ho,Wo,Cout]"""
        b, h, w, cin = x_shape
        wt, phase_w, dt = (u.wt16, u.phase_w16, dz.dtype) if self.op16 else (u.wt, u.phase_w, F32)
        if u.stride == 1:
            p = u.k - 1 - u.pad
            return ops.conv_nhwc(dz, wt, None, kh=u.k, kw=u.k, pad=(p, p), residual=residual)
        assert h % 2 == 0 and w % 2 == 0, "stride-2 data gradient is built for even input sizes"
        ho, wo = dz.shape[1], dz.shape[2]
        dx = torch.zeros((b, h, w, cin), dtype=dt, device=dz.device) if (u.k == 1 and residual is None) else \
            (residual.clone() if u.k == 1 else torch.empty((b, h, w, cin), dtype=dt, device=dz.device))
        if u.k == 1:   # dX[2a][2b] = W^T dY[a][b]; every other position keeps the residual (or zero)
            rm = self._row_map(ho, wo, w, 0, 0)
            ops.conv_nhwc(dz, wt, None, kh=1, kw=1, residual=dx if residual is not None else None, out=dx, out_row_map=rm, out_rows_per_image=h * w)
            return dx
        for (ph, pw), (wsub, khs, kws) in phase_w.items():
            rm = self._row_map(ho, wo, w, ph, pw)
            ops.conv_nhwc(dz, wsub, None, kh=khs, kw=kws, out_hw=(ho, wo), residual=residual, out=dx, out_row_map=rm, out_rows_per_image=h * w)
        return dx

    def _row_map(self, ho, wo, w, ph, pw) -> torch.Tensor:
        """output rows (2a+ph, 2b+pw) of a [*, 2*ho, w] image for the ho x wo results of one sub-pixel phase (cached on the device)"""
        key = (ho, wo, w, ph, pw)
        if key not in self._row_maps:
            aa, bb = torch.meshgrid(torch.arange(ho), torch.arange(wo), indexing="ij")
            self._row_maps[key] = ((2 * aa + ph) * w + 2 * bb + pw).reshape(-1).to(torch.int32).to(self.dev)
        return self._row_maps[key]

    def _bwd_unit(self, rec, dy, residual_for_dx=None, want_dres=False, need_dx=True):
        u, x, z, mean, invstd, a, relu, has_res = rec
        c = z.shape[-1]
        if self.op16:   # (units without a residual input recompute their ReLU gate from z: the stored activation is not read)
            dz, dres = ops.bn_backward_t(dy.reshape(-1, c), a.view(-1, c) if relu else None, z.view(-1, c), mean, invstd, u.gamma, u.ggamma, u.gbeta,
                                         relu=relu, want_dres=want_dres, sums=u.sums_b, beta=u.beta if (relu and not has_res) else None)
        else:
            dz, dres = ops.bn_backward(dy.reshape(-1, c), a.view(-1, c) if relu else None, z.view(-1, c), mean, invstd, u.gamma, u.ggamma, u.gbeta,
                                       relu=relu, want_dres=want_dres, sums=u.sums_b, beta=u.beta if (relu and not has_res) else None)
        dz = dz.view(z.shape)
        if self.op16 and u.cin != 4:
            ops.wgrad_conv2d_bf16(dz, x, u.gw, u.k, u.stride)                                         # (adds to G, zeroed once per step)
        else:
            ops.wgrad_conv2d(dz, x, u.gw, u.k, u.k, (u.stride, u.stride), (u.pad, u.pad), zero=False)     # G is zeroed once per step
        dx = self._dgrad(u, dz, x.shape, residual_for_dx) if need_dx else None
        return dx, (dres.view(z.shape) if dres is not None else None)

    def _linear_fwd(self, name, x):
        w, b, _, _, co, ci = self.lin[name]
        return ops.linear(x, w, b)

    def _linear_bwd(self, name, dy, x, need_dx=True):
        w, b, gw, gb, co, ci = self.lin[name]
        ops.wgrad_conv1d(dy, x, gw, batch=1, t=x.shape[0], taps=1, dil=1, pad=0, accumulate=True, bias_grad=gb)   # (G is zeroed once per step)
        if not need_dx:
            return None
        wt = ops.transpose_pack_conv1d(w, co, ci, 1)
        return ops.linear(dy, wt, None)

    # ------------------------------------------------------------------ one step
    def train_step(self, frames, labels, teacher_pred, teacher_feat, apply_update: bool = True, use_graph: bool = False):
        """frames: normalised float32 NCHW [B,3,H,W] or uint8 NHWC [B,H,W,3] on the GPU; labels (y_i, y_v, y_t, y_ivt) multi-hot [B,K]
        (or one prepared fp32 [B,131] device tensor); teacher_pred 3 x raw logits [B,K]; teacher_feat 3 x [B,1536].
        Returns the dict of loss terms.  use_graph: replay a hipGraph of the whole forward+backward captured for this input shape
        (~900 launches become one; running statistics and gradients are updated by the replay exactly as by the eager step)."""
        dev = self.dev
        if not torch.is_tensor(labels) and self.loss_type != "all":       # (y_i, y_v, y_t, y_ivt) as the loader yields them: keep the task's
            labels = [labels["ivt".index(self.loss_type)]] if len(labels) == 4 else labels
        z = labels if torch.is_tensor(labels) else torch.cat([l.to(dev, F32) for l in labels], 1).contiguous()
        tp = [t.to(dev, F32).contiguous() for t in teacher_pred]
        tf = [t.to(dev, F32).contiguous() for t in teacher_feat]
        B = frames.shape[0]
        assert frames.is_cuda and tuple(z.shape) == (B, self.NH)
        self.bucket_order = []            # gradient buckets in the order their all-reduce was issued this step (DDP)
        if use_graph:
            # data-parallel steps with bucket overlap: the backward is captured in SEGMENTS cut where a gradient bucket is complete
            # (`_reduce_bucket`), and the bucket's all-reduce is issued between two replays -- one graph would leave ONE flat all-reduce
            # behind the whole backward (171 MB for the ResNet-50 student)
            seg = self.overlap and getattr(self, "exchange", True) and self._ddp_world() > 1
            key = (tuple(frames.shape), frames.dtype, bool(seg))
            g = self._graphs.get(key)
            if g is None:
                from .graph import GraphedForward, SegmentedGraph
                keep = {n: (u.rmean.clone(), u.rvar.clone()) for n, u in self.units.items()}   # warm-up + capture runs must not count
                self._capturing = True
                try:
                    fn = lambda f, zz, *t: self._fwd_bwd(f, zz, t[:3], t[3:])
                    if seg:
                        def fn_cut(cut, *a):
                            self._cut = cut
                            try:
                                return fn(*a)
                            finally:
                                self._cut = None
                        g = SegmentedGraph(fn_cut, [frames, z, *tp, *tf])
                    else:
                        g = GraphedForward(fn, [frames, z, *tp, *tf])
                    self._graphs[key] = g
                finally:
                    self._capturing = False
                    self._cut = None
                for n, u in self.units.items():
                    u.rmean.copy_(keep[n][0])
                    u.rvar.copy_(keep[n][1])
            if seg:
                col_loss, soft, kdl = g(frames, z, *tp, *tf, on_cut=self._issue_bucket)
            else:
                col_loss, soft, kdl = g(frames, z, *tp, *tf)
        else:
            col_loss, soft, kdl = self._fwd_bwd(frames, z, tp, tf)
        for bn in self.nbt:
            self.nbt[bn] += 1
        r0, r1, r2 = self.rates
        cl, sk = col_loss.cpu(), torch.cat([soft, kdl]).cpu()
        terms, o = {}, 0
        hard = 0.0
        for t, k in self.heads:
            terms["hard_" + t] = float(cl[o:o + k].sum() / (B * k))
            hard += terms["hard_" + t]
            o += k
        terms.update(hard=hard, soft=float(sk[0]) / 3.0, kd=float(sk[1]) / 3.0)
        terms["loss"] = (r0 * terms["hard"] + r1 * terms["soft"] + r2 * terms["kd"]) if self.loss_type == "all" else hard   # `run.py:165-192`
        if apply_update:
            self.apply_update()
        return terms

    def forward_train(self, frames, teacher_feats=None):
        """The reference module call under `model.train()` (`Spatial_cnn/network.py:43-92`): BatchNorm on batch statistics (running statistics and
        num_batches_tracked advance, as in torch), the KD branch when teacher features are given (`loss_type all` and `args.train`, `network.py:47`).  frames uint8 NHWC or normalised float32 NCHW on the
        GPU; teacher_feats = (feat_i, feat_v, feat_t) [B, teacher_dim].  Returns ((kd_i | 0, logit_i), (kd_v | 0, logit_v), (kd_t | 0, logit_t),
        (feat, logit_ivt)) with zeros for heads the model does not have (`network.py:79-82`).  Forward only: gradients come from `train_step`."""
        from .synth import IMAGENET_MEAN, IMAGENET_STD
        if frames.dtype == torch.uint8:
            xp = ops.preprocess_u8(frames, IMAGENET_MEAN, IMAGENET_STD, F32)
        else:
            xp = ops.pad_nchw(frames, F32)
        B = frames.shape[0]
        self._arena.zero_()
        pre, U = "basemodel.basemodel.", self.units
        x = ops.maxpool3x3s2(self._fwd_unit(U[pre + "conv1"], xp))
        bott = self.network == "resnet50"
        for li, n in enumerate(_DEPTHS[self.network], start=1):
            for bi in range(n):
                q = f"{pre}layer{li}.{bi}."
                idt = self._fwd_unit(U[q + "downsample.0"], x, relu=False) if (q + "downsample.0") in U else x
                o = self._fwd_unit(U[q + "conv1"], x)
                if bott:
                    o = self._fwd_unit(U[q + "conv2"], o)
                    x = self._fwd_unit(U[q + "conv3"], o, residual=idt)
                else:
                    x = self._fwd_unit(U[q + "conv2"], o, residual=idt)
        for bn in self.nbt:
            self.nbt[bn] += 1
        feat = ops.global_avgpool(x)
        logits = self._linear_fwd("heads", feat)
        outs, o = {}, 0
        for t, k in self.heads:
            outs[t] = logits[:, o:o + k]
            o += k
        for t, k in _ALL_HEADS:
            outs.setdefault(t, torch.zeros((B, k), device=self.dev))
        cams = [0, 0, 0]
        if teacher_feats is not None:
            assert self.loss_type == "all" and len(teacher_feats) == 3
            teas = [self._linear_fwd(m, t.to(self.dev, F32).contiguous()) for m, t in zip(("mi", "mv", "mt"), teacher_feats)]
            cams = [self._linear_fwd(wn, mx) for wn, mx in zip(("wi", "wv", "wt"), ops.kd_mix(feat, *teas))]
        return (cams[0], outs["i"]), (cams[1], outs["v"]), (cams[2], outs["t"]), (feat, outs["ivt"])

    def _fwd_bwd(self, frames, z, tp, tf):
        """device part of a step (enqueue only): forward, losses, backward into self.G.  Returns (per-column BCE sums, soft, kd)."""
        from .synth import IMAGENET_MEAN, IMAGENET_STD
        dev, C = self.dev, self.C
        if frames.dtype == torch.uint8:
            B, H, W = frames.shape[0], frames.shape[1], frames.shape[2]
            xp = ops.preprocess_u8(frames, IMAGENET_MEAN, IMAGENET_STD, F32)
        else:
            B, H, W = frames.shape[0], frames.shape[2], frames.shape[3]
            xp = ops.pad_nchw(frames, F32)
        self._arena.zero_()
        self.G.zero_()
        saved: List[tuple] = []
        self.last_saved = saved          # (unit, input, conv output, mean, invstd, post-activation, relu, has residual) per conv+BN, forward order
        pre = "basemodel.basemodel."
        U = self.units
        a0 = self._fwd_unit(U[pre + "conv1"], xp, saved=saved)
        x = ops.maxpool3x3s2(a0)
        blocks = []
        bott = self.network == "resnet50"
        for li, n in enumerate(_DEPTHS[self.network], start=1):
            for bi in range(n):
                q = f"{pre}layer{li}.{bi}."
                first = len(saved)
                has_ds = (q + "downsample.0") in U
                idt = self._fwd_unit(U[q + "downsample.0"], x, relu=False, saved=saved) if has_ds else x
                o = self._fwd_unit(U[q + "conv1"], x, saved=saved)
                if bott:
                    o = self._fwd_unit(U[q + "conv2"], o, saved=saved)
                    x = self._fwd_unit(U[q + "conv3"], o, residual=idt, saved=saved)
                else:
                    x = self._fwd_unit(U[q + "conv2"], o, residual=idt, saved=saved)
                blocks.append((first, has_ds, bott, li))
        Bh, Hh, Wh, _ = x.shape
        feat = ops.global_avgpool(x)                                               # [B, C]
        # ---- heads, KD branch
        NH, NHP, kd_on = self.NH, self.NHP, self.loss_type == "all"
        logits = self._linear_fwd("heads", feat)                                   # [B, 132] (single task: its K rounded up to 4)
        # ---- losses and their gradients
        r0, r1, r2 = self.rates
        col_scale = self._col_scale(B)
        col_loss = torch.zeros(NH, device=dev)
        dlog = torch.zeros((B, NHP), device=dev)
        ops.bce_logits_pw(logits[:, :NH], z, self.pos_weight, col_scale, dlog, col_loss)
        soft = torch.zeros(1, device=dev)
        kdl = torch.zeros(1, device=dev)
        if kd_on:
            teas = [self._linear_fwd(m, t) for m, t in zip(("mi", "mv", "mt"), tf)]
            mixed = ops.kd_mix(feat, *teas)
            cams = [self._linear_fwd(wn, mx) for wn, mx in zip(("wi", "wv", "wt"), mixed)]
            o = 0
            for (t, k), tpn in zip(self.heads[:3], tp):
                ops.distill_kl(logits[:, o:o + k], tpn, dlog[:, o:o + k], soft, self.temp, r1 / 3.0, accumulate=True)
                o += k
            dcams = [ops.mse(c, t, kdl, r2 / 3.0) for c, t in zip(cams, tf)]
        # ---- backward: heads + KD branch -> dfeat
        dfeat = self._linear_bwd("heads", dlog, feat)
        if kd_on:
            gs = [self._linear_bwd(wn, dc, mx) for wn, dc, mx in zip(("wi", "wv", "wt"), dcams, mixed)]
            ds_kd, dtau = ops.kd_mix_bwd(feat, teas, gs)
            dfeat = ops.mul_add(dfeat, torch.ones_like(dfeat), ds_kd)
            for n, (m, t) in enumerate(zip(("mi", "mv", "mt"), tf)):
                dte = dtau[:, n:n + 1].expand(B, C).contiguous()                    # d(tea_n)[b][:] = dtau[b][n]
                self._linear_bwd(m, dte, t, need_dx=False)
        self._reduce_bucket("heads")
        # ---- backward through the trunk
        dx = (ops.avgpool_bwd_bf16 if self.op16 else ops.avgpool_bwd)(dfeat, Bh, Hh * Wh, C).view(Bh, Hh, Wh, C)
        for bi_, (first, has_ds, bott, li) in reversed(list(enumerate(blocks))):
            recs = saved[first:first + (1 if has_ds else 0) + (3 if bott else 2)]
            main = recs[1:] if has_ds else recs
            d, dres = self._bwd_unit(main[-1], dx, want_dres=True)                  # last conv: ReLU gate after the residual add
            for rec in reversed(main[1:-1]):
                d, _ = self._bwd_unit(rec, d)
            if has_ds:
                d_id, _ = self._bwd_unit(recs[0], dres)                             # identity path through downsample conv+bn
                dx, _ = self._bwd_unit(main[0], d, residual_for_dx=d_id)
            else:
                dx, _ = self._bwd_unit(main[0], d, residual_for_dx=dres)
            if bi_ == 0 or blocks[bi_ - 1][3] != li:      # first block of the layer done: the layer's gradients are complete
                self._reduce_bucket(f"layer{li}")
        da0 = (ops.maxpool3x3s2_bwd_bf16 if self.op16 else ops.maxpool3x3s2_bwd)(a0, dx)
        self._bwd_unit(saved[0], da0, need_dx=False)
        self._reduce_bucket("stem")
        return col_loss, soft, kdl

    def _ddp_world(self) -> int:
        import torch.distributed as dist
        return dist.get_world_size(self.pg) if (dist.is_available() and dist.is_initialized()) else 1

    def _reduce_bucket(self, name: str):
        """DDP overlap: the bucket's all-reduce is enqueued behind the kernels that wrote it and runs while the backward
        of the earlier layers continues; `apply_update` waits for all of them.  Not inside a hipGraph capture."""
        if self._capturing:
            if getattr(self, "_cut", None) is not None:     # segmented capture: the graph is cut here, the all-reduce is issued at replay
                self._cut(name)
            return
        if not self.overlap or not getattr(self, "exchange", True) or self._ddp_world() == 1:
            return
        self._issue_bucket(name)

    def _issue_bucket(self, name: str):
        import torch.distributed as dist
        a, b = self._ranges[name]
        self.bucket_order = getattr(self, "bucket_order", []) + [name]
        if b > a:
            self._pending.append(dist.all_reduce(self.G[a:b], op=dist.ReduceOp.SUM, group=self.pg, async_op=True))

    def _col_scale(self, B: int) -> torch.Tensor:
        if B not in self._col_scales:
            r0 = self.rates[0] if self.loss_type == "all" else 1.0       # a single-task loss carries no rate (`run.py:165-179`)
            self._col_scales[B] = torch.cat([torch.full((k,), r0 / (B * k)) for _, k in self.heads]).to(self.dev)
        return self._col_scales[B]

    def relu_outputs(self) -> Dict[str, torch.Tensor]:
        """post-ReLU activations of the last step by BatchNorm name, NCHW on the host (tests: ReLU-gate comparison)"""
        return {u.bn: a.permute(0, 3, 1, 2).cpu() for (u, _, _, _, _, a, relu, _) in self.last_saved if relu}

    def apply_update(self):
        if self._pending:                                   # buckets were reduced during the backward
            for h in self._pending:
                h.wait()
            self._pending = []
            scale = 1.0 / self._ddp_world()
        elif getattr(self, "exchange", True):
            scale = allreduce_sum_flat(self.G, self.pg)
        else:
            scale = 1.0                                     # exchange=False: rank-local step (bench: the step without its exchange)
        ops.sgd_step(self.P, self.G, self.lr, self.wd, scale)
        self._refresh_transposed()
