"""One training step of the temporal head on MI355X: forward (train mode), BCE-with-logits over the 4 FPN levels,
backward and SGD, as `Temporal_tenco/run.py:181-235` does with autograd -- here as explicit HIP launches.

* forward: the inference kernels (`mt4_conv_nhwc`), keeping each layer's input and ReLU output;
* data gradients: the SAME implicit-GEMM kernel with transposed / tap-reversed weights (`mt4_transpose_pack_conv1d_f32`),
  the ReLU gate and the residual fan-in fused into its epilogue (act "relu_gate", `residual`);
* weight gradients: `mt4_wgrad_conv1d_f32` (fp32 MFMA, contraction over time), the bias gradient (column sums of dY) in the same launch;
* loss: `mt4_bce_logits_f32` on the concatenated [T][131] heads; optimizer: `mt4_sgd_step_f32` on ONE flat parameter
  buffer (packed layouts), so DDP is ONE all-reduce of the flat gradient buffer over RCCL (videos shard over ranks).

Randomness (75 % input mask, Dropout2d, per-layer Dropout; `network.py:43-48,123-127,194-196`) is drawn on the host side
of this class (`draw_masks`) or passed in explicitly, so that parity tests feed the oracle the same draw.
Parameters that the FPN configuration never reaches (PG.conv_out, Rs.*.conv_1x1, Rs.*.conv_out, fpn.latlayer2/3) get no
gradient and, like torch.optim.SGD with `grad is None`, are left untouched.
"""
from __future__ import annotations

from typing import Dict, List, Optional

import torch

from . import ops
from .shapes import tenco_shapes

HEADS = (("", 100, 1.0), ("_i", 6, 0.1), ("_v", 10, 0.1), ("_t", 15, 0.1))   # loss = 0.1 (i + v + t) + ivt (`run.py:212`)
NH = 131
NHP = 132  # padded to a multiple of 4 (16-byte rows for the gradient GEMMs)


class _Conv:
    __slots__ = ("name", "cout", "cin", "taps", "w", "b", "gw", "gb", "wt", "kpad")


class TencoTrainer:
    def __init__(self, num_layers_PG=11, num_layers_R=10, num_R=3, num_f_maps=512, dim=512, num_classes=100, lr=0.1, weight_decay=1e-5,
                 device: str = "cuda", process_group=None, overlap: bool = True, hier: bool = False):
        # hier = `args.hier` (`network.py:147,154-155`): AvgPool1d(7, 3) behind every refinement stage, so level l has its own length T_l, the FPN
        # resamples (`:96`) and `fusion` scores every level against the labels resized to it (`run.py:159-179,196-212`)
        self.hier = bool(hier)
        self.overlap = overlap            # DDP: all-reduce a stage's gradients as soon as its backward has written them (eager steps)
        self._pending: list = []
        self._capturing = False
        assert num_classes == 100 and num_R == 3, "the FPN training recipe (Scripts/train_fold1.sh:28) has PG + 3 refinement stages"
        self.LP, self.LR, self.R, self.C, self.D = num_layers_PG, num_layers_R, num_R, num_f_maps, dim
        self.lr, self.wd = lr, weight_decay
        self.dev = torch.device(device)
        self.pg = process_group
        self._table = tenco_shapes(num_layers_PG, num_layers_R, num_R, num_f_maps, dim, 100, fpn=True)
        self._extra: Dict[str, torch.Tensor] = {}   # parameters outside the trained graph, kept verbatim
        self.convs: Dict[str, _Conv] = {}
        self._graphs: Dict[int, object] = {}
        self._scales: Dict[int, torch.Tensor] = {}
        self._label_idx: Dict[tuple, torch.Tensor] = {}

    # ------------------------------------------------------------------ parameters
    def _stages(self):
        return [("PG", self.LP)] + [(f"Rs.{r}", self.LR) for r in range(self.R)]

    def load_state_dict(self, sd: Dict[str, torch.Tensor]):
        names = [k for k, _ in self._table]
        assert all(k in sd for k in names), "state dict incomplete"
        C, D = self.C, self.D
        specs = [("PG.conv_1x1", C, D, 1)]
        for prefix, n in self._stages():
            for i in range(n):
                specs += [(f"{prefix}.layers.{i}.conv_dilated", C, C, 3), (f"{prefix}.layers.{i}.conv_1x1", C, C, 1)]
        specs += [("fpn.latlayer1", C, C, 1), ("heads", NHP, C, 1)]
        f32 = torch.float32
        sizes = []
        for name, cout, cin, taps in specs:
            kp = ops.packed_k(cin, 1, taps, f32)
            sizes.append(cout * kp + ((cout + 3) // 4) * 4)
        total = sum(sizes)
        self.P = torch.zeros(total, dtype=f32, device=self.dev)
        self.G = torch.zeros(total, dtype=f32, device=self.dev)
        self._tab = ops.RefreshTable(self.dev)   # the transposed, tap-reversed data-gradient operators: ONE launch after every update
        off = 0
        self._ranges: Dict[str, list] = {}     # flat-buffer range of every gradient bucket: "PG", "Rs.0".."Rs.2", "heads" (FPN lateral + heads)
        for name, cout, cin, taps in specs:
            bucket = name.split(".layers.")[0] if ".layers." in name else ("PG" if name == "PG.conv_1x1" else "heads")
            self._ranges.setdefault(bucket, [off, off])
            c = _Conv()
            c.name, c.cout, c.cin, c.taps = name, cout, cin, taps
            c.kpad = ops.packed_k(cin, 1, taps, f32)
            nw = cout * c.kpad
            c.w, c.gw = self.P[off:off + nw].view(cout, c.kpad), self.G[off:off + nw].view(cout, c.kpad)
            c.b, c.gb = self.P[off + nw:off + nw + cout], self.G[off + nw:off + nw + cout]
            off += nw + ((cout + 3) // 4) * 4
            self._ranges[bucket][1] = off
            if name == "heads":
                w = torch.cat([sd[f"conv_out{s}.weight"] for s, _, _ in HEADS], 0).float()
                b = torch.cat([sd[f"conv_out{s}.bias"] for s, _, _ in HEADS], 0).float()
                w = torch.cat([w, torch.zeros(NHP - NH, cin, 1)], 0)
                b = torch.cat([b, torch.zeros(NHP - NH)], 0)
            else:
                w, b = sd[name + ".weight"].float(), sd[name + ".bias"].float()
            c.w.copy_(ops.pack_conv_weight(w.to(self.dev).unsqueeze(2), None, f32))
            c.b.copy_(b.to(self.dev))
            # (the input projection needs no data gradient)
            c.wt = self._tab.add(c.w, cout, cin, f32, True, [taps - 1 - i for i in range(taps)]) if name != "PG.conv_1x1" else None
            self.convs[name] = c
        trained = {n + suffix for n, *_ in specs for suffix in (".weight", ".bias")} | \
                  {f"conv_out{s}.{p}" for s, _, _ in HEADS for p in ("weight", "bias")}
        self._extra = {k: sd[k].detach().clone() for k in names if k not in trained}
        self._refresh_transposed()
        return self

    def _refresh_transposed(self):
        """(a launch of `mt4_transpose_pack_conv1d_f32` per layer was 98 launches = 0.5 ms of an 8 ms whole-video step)"""
        self._tab.run()

    def state_dict(self) -> Dict[str, torch.Tensor]:
        """reference layout and key names (`Temporal_tenco/network.py`), on the CPU"""
        out = dict(self._extra)
        for name, c in self.convs.items():
            w = c.w[:, :c.taps * c.cin].reshape(c.cout, c.taps, c.cin).permute(0, 2, 1).contiguous().cpu()
            b = c.b.clone().cpu()
            if name == "heads":
                o = 0
                for s, k, _ in HEADS:
                    out[f"conv_out{s}.weight"], out[f"conv_out{s}.bias"] = w[o:o + k].clone(), b[o:o + k].clone()
                    o += k
            else:
                out[name + ".weight"], out[name + ".bias"] = w, b
        return {k: out[k] for k, _ in self._table}

    def grads(self) -> Dict[str, torch.Tensor]:
        """gradients of the last step in reference layout (trained parameters only), on the CPU"""
        out = {}
        for name, c in self.convs.items():
            g = c.gw[:, :c.taps * c.cin].reshape(c.cout, c.taps, c.cin).permute(0, 2, 1).contiguous().cpu()
            gb = c.gb.clone().cpu()
            if name == "heads":
                o = 0
                for s, k, _ in HEADS:
                    out[f"conv_out{s}.weight"], out[f"conv_out{s}.bias"] = g[o:o + k].clone(), gb[o:o + k].clone()
                    o += k
            else:
                out[name + ".weight"], out[name + ".bias"] = g, gb
        return out

    def level_lengths(self, t: int) -> List[int]:
        """frames of the four FPN levels for a video of t frames: all t, or t -> (t - 7) // 3 + 1 per refinement stage with --hier"""
        out = [t]
        for _ in range(self.R):
            if self.hier:
                if t < 7:
                    raise ValueError(f"--hier: a level of {t} < 7 frames cannot be pooled (AvgPool1d(7, 3))")
                t = (t - 7) // 3 + 1
            out.append(t)
        return out

    # ------------------------------------------------------------------ randomness
    def draw_masks(self, t: int, generator: Optional[torch.Generator] = None) -> dict:
        """the train-time random pieces as reference-shaped tensors ([1,D,T] / [1,D,1] / {prefix: [1,C,T_stage]}: a refinement stage runs at the
        length of the level it reads)"""
        g = generator
        n = self.D * t
        perm = torch.randperm(n, generator=g)
        flat = torch.cat((torch.zeros(n - int(n * 0.75)), torch.ones(int(n * 0.75))))[perm]   # `network.py:44-47`
        masks = {"input_mask": flat.view(1, self.D, t), "channel_mask": (torch.rand(1, self.D, 1, generator=g) >= 0.5).float() * 2.0,
                 "layer_masks": {}}
        lens = self.level_lengths(t)
        for si, (prefix, nl) in enumerate(self._stages()):
            ts = lens[max(si - 1, 0)]
            for i in range(nl):
                masks["layer_masks"][f"{prefix}.layers.{i}"] = (torch.rand(1, self.C, ts, generator=g) >= 0.5).float() * 2.0
        return masks

    # ------------------------------------------------------------------ one step
    def _conv(self, x, c: _Conv, dil=1, residual=None, act=None, transposed=False, cout=None):
        w = c.wt if transposed else c.w
        b = None if transposed else (c.b if cout is None else c.b[:cout])
        pad = dil if c.taps == 3 else 0
        return ops.conv_nhwc(x, w, b, kh=1, kw=c.taps, pad=(0, pad), dil=(1, dil), residual=residual, act=act)

    @ops.with_latency_tiles
    def train_step(self, x: torch.Tensor, labels: Dict[str, torch.Tensor], masks: Optional[dict] = None, apply_update: bool = True,
                   use_graph: bool = False):
        """x [1,T,D] fp32 on the GPU; labels {'': [T,100], '_i': [T,6], '_v': [T,10], '_t': [T,15]} multi-hot.
        Returns (loss, {head: loss term}).  use_graph: replay a hipGraph of the whole forward+backward captured for this T
        (no masks: the draw changes every step) -- ~900 launches become one."""
        assert x.dim() == 3 and x.shape[0] == 1 and x.shape[2] == self.D
        T, dev = x.shape[1], self.dev
        z = labels if torch.is_tensor(labels) else self.prepare_labels(labels)
        assert z.is_cuda and tuple(z.shape) == (T, NH) and z.dtype == torch.float32
        self.bucket_order = []            # gradient buckets in the order their all-reduce was issued this step (DDP)
        if use_graph and not masks:
            # data-parallel steps with bucket overlap: the backward is captured in segments cut where a stage's gradients are complete, the
            # stage's all-reduce is issued between two replays (`graph.SegmentedGraph`); otherwise one graph, one flat all-reduce behind it
            seg = self.overlap and getattr(self, "exchange", True) and self._ddp_world() > 1
            g = self._graphs.get((T, bool(seg)))
            if g is None:
                from .graph import GraphedForward, SegmentedGraph
                self._capturing = True
                try:
                    if seg:
                        def fn_cut(cut, xx, zz):
                            self._cut = cut
                            try:
                                return self._fwd_bwd(xx, zz, None)
                            finally:
                                self._cut = None
                        g = SegmentedGraph(fn_cut, [x, z])
                    else:
                        g = GraphedForward(lambda xx, zz: self._fwd_bwd(xx, zz, None), [x, z])
                    self._graphs[(T, bool(seg))] = g
                finally:
                    self._capturing = False
            if seg:
                col_loss = g(x, z, on_cut=self._issue_bucket)
            else:
                col_loss = g(x, z)
        else:
            col_loss = self._fwd_bwd(x, z, masks)
        cl = col_loss.cpu()                                  # [levels, 131]: BCE sums per level and column
        lens = self.level_lengths(T)
        terms, o = {}, 0
        for s, k, _ in HEADS:
            terms[s] = float(sum(cl[l, o:o + k].sum() / (lens[l] * k) for l in range(len(lens))))   # BCEWithLogitsLoss(mean) per level (`run.py:196-210`)
            o += k
        loss = sum(w * terms[s] for s, _, w in HEADS)
        if apply_update:
            self.apply_update()
        return loss, terms

    def prepare_labels(self, labels: Dict[str, torch.Tensor]) -> torch.Tensor:
        """{'': [T,100], '_i': [T,6], '_v': [T,10], '_t': [T,15]} -> one fp32 [T,131] device tensor.  Goes through pinned
        memory: a pageable copy of this size makes ROCm pin the user pages on the fly, which was measured to stall the
        step by 70-170 ms every other step (tools/probe_train2.py)."""
        z = torch.cat([labels[s].to(torch.float32) for s, _, _ in HEADS], 1).contiguous()
        return z.pin_memory().to(self.dev, non_blocking=True) if not z.is_cuda else z

    def _fwd_bwd(self, x: torch.Tensor, z: torch.Tensor, masks: Optional[dict]) -> torch.Tensor:
        """device part of a step (enqueue only): forward, loss, backward into self.G.  Returns the per-column loss sums."""
        T, C, dev = x.shape[1], self.C, self.dev
        cv = self.convs
        to_rows = lambda m: m[0].transpose(0, 1).contiguous().to(dev)        # [1,C,T] -> [T,C]
        h0 = x.contiguous().view(1, 1, T, self.D)
        if masks and masks.get("input_mask") is not None:
            h0 = ops.mul_add(h0, to_rows(masks["input_mask"]).view_as(h0))
        if masks and masks.get("channel_mask") is not None:
            h0 = ops.mul_add(h0, masks["channel_mask"][0].transpose(0, 1).expand(T, self.D).contiguous().to(dev).view_as(h0))
        lm = {k: to_rows(v).view(1, 1, v.shape[-1], C) for k, v in ((masks or {}).get("layer_masks") or {}).items()}

        # ---- forward, saving layer inputs z and ReLU outputs u
        lens = self.level_lengths(T)                       # frames per FPN level (all T without --hier)
        saved: List[tuple] = []
        f = self._conv(h0, cv["PG.conv_1x1"])
        stage_out = []
        for si, (prefix, n) in enumerate(self._stages()):
            ts = lens[max(si - 1, 0)]                      # a refinement stage runs at the length of the level it reads
            for i in range(n):
                p = f"{prefix}.layers.{i}"
                d = 2 ** i
                u = self._conv(f, cv[p + ".conv_dilated"], dil=d, act="relu")
                if p in lm:
                    o = self._conv(u, cv[p + ".conv_1x1"])
                    fn = ops.mul_add(o, lm[p], f)
                else:
                    fn = self._conv(u, cv[p + ".conv_1x1"], residual=f)
                saved.append((p, d, f, u, ts))
                f = fn
            if self.hier and si > 0:                       # `Refinement.forward` (:154-155): the stage's feature is the pooled map
                f = ops.avgpool1d_rows(f.view(1, ts, C), 7, 3).view(1, 1, lens[si], C)
            stage_out.append(f)
        c1, c2, c3, p4 = stage_out
        lat = cv["fpn.latlayer1"]
        # `FPN._upsample_add` (:93-96): linear interpolation to the lateral's length (the identity at equal lengths) + lateral
        up = (lambda xx, li: ops.interp_linear_rows(xx.view(1, lens[li + 1], C), lens[li]).view(1, 1, lens[li], C)) if self.hier else (lambda xx, li: xx)
        p3 = self._conv(c3, lat, residual=up(p4, 2))
        p2 = self._conv(c2, lat, residual=up(p3, 1))
        p1 = self._conv(c1, lat, residual=up(p2, 0))
        levels = [p1, p2, p3, p4]
        hd = cv["heads"]
        logits = [self._conv(l, hd) for l in levels]                          # [1,1,T_l,132], column 131 is padding (= 0)

        # ---- loss + dL/dlogits
        # every weight / bias gradient below ADDS into the flat buffer, zeroed here once: zeroing per tensor in front of its (atomic) sums was
        # ~150 launches = 0.8 ms of an 8 ms whole-video step
        self.G.zero_()
        col_loss = torch.zeros((len(levels), NH), device=dev)
        dys = []
        for li, lg in enumerate(logits):
            tl = lens[li]
            dy = torch.zeros((1, 1, tl, NHP), device=dev)
            ops.bce_logits(lg.view(tl, NHP)[:, :NH], self._level_labels(z, T, tl), self._col_scale(tl), dy.view(tl, NHP), col_loss[li])
            dys.append(dy)
        # ---- backward: heads and FPN (top-down adds fan the level gradients into each other)
        g = None
        dstage = [None, None, None, None]
        for li, (lv, dy) in enumerate(zip(levels, dys)):
            tl = lens[li]
            ops.wgrad_conv1d(dy.view(tl, NHP), lv.view(tl, C), hd.gw, batch=1, t=tl, taps=1, dil=1, pad=0, accumulate=True, bias_grad=hd.gb)
            if g is not None and self.hier:                                    # p_{li} = interp(p_{li+1}) + lateral: the adjoint of the resampling
                g = ops.interp_linear_rows_bwd(g.view(1, lens[li - 1], C), tl).view(1, 1, tl, C)
            g = self._conv(dy, hd, transposed=True, residual=g)               # gradient w.r.t. p_{li+1}
            if li < 3:
                cl_ = stage_out[li]                                            # lateral input c_{li+1}
                ops.wgrad_conv1d(g.view(tl, C), cl_.view(tl, C), lat.gw, batch=1, t=tl, taps=1, dil=1, pad=0, accumulate=True, bias_grad=lat.gb)
                dstage[li] = self._conv(g, lat, transposed=True)
            else:
                dstage[3] = g                                                  # p4 is the last stage's (pooled) output itself
        self._reduce_bucket("heads")
        # ---- backward through the stages, last to first
        df = dstage[3]
        idx = len(saved)
        for si in range(len(self._stages()) - 1, -1, -1):
            prefix, n = self._stages()[si]
            ts = lens[max(si - 1, 0)]
            if self.hier and si > 0:                                           # through the stage's AvgPool1d(7, 3)
                df = ops.avgpool1d_rows_bwd(df.view(1, lens[si], C), ts, 7, 3).view(1, 1, ts, C)
            for i in range(n - 1, -1, -1):
                idx -= 1
                p, d, zin, u, _ = saved[idx]
                w1, wd = cv[p + ".conv_1x1"], cv[p + ".conv_dilated"]
                do = ops.mul_add(df, lm[p]) if p in lm else df
                ops.wgrad_conv1d(do.view(ts, C), u.view(ts, C), w1.gw, batch=1, t=ts, taps=1, dil=1, pad=0, accumulate=True, bias_grad=w1.gb)
                du = self._conv(do, w1, transposed=True, residual=u, act="relu_gate")
                ops.wgrad_conv1d(du.view(ts, C), zin.view(ts, C), wd.gw, batch=1, t=ts, taps=3, dil=d, pad=d, accumulate=True, bias_grad=wd.gb)
                df = self._conv(du, wd, dil=d, transposed=True, residual=df)
            if si > 0:
                df = ops.mul_add(df, torch.ones_like(df), dstage[si - 1])      # + gradient of this stage's input as lateral c
                self._reduce_bucket(prefix)                                    # this stage's gradients are complete: exchange them behind
        pin = cv["PG.conv_1x1"]                                                #   the backward of the earlier stages
        ops.wgrad_conv1d(df.view(T, C), h0.view(T, self.D), pin.gw, batch=1, t=T, taps=1, dil=1, pad=0, accumulate=True, bias_grad=pin.gb)
        self._reduce_bucket("PG")
        return col_loss

    def _ddp_world(self) -> int:
        import torch.distributed as dist
        return dist.get_world_size(self.pg) if (dist.is_available() and dist.is_initialized()) else 1

    def _reduce_bucket(self, name: str):
        """DDP overlap: the bucket's all-reduce is enqueued behind the kernels that wrote it and runs while the backward of the
        earlier stages continues; `apply_update` waits for all of them.  Under graph replay the backward is replayed in segments cut at these
        points (`train_step`)."""
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

    def _level_labels(self, z: torch.Tensor, T: int, tl: int) -> torch.Tensor:
        """`fusion` (`run.py:169-175`): the labels of a level of tl != T frames are `F.interpolate(labels, size=tl, mode='nearest')` of the [T, K] rows:
        row j <- row min(floor(j * float32(T / tl)), T - 1) (torch's nearest index arithmetic, in float32)"""
        if tl == T:
            return z
        idx = self._label_idx.get((T, tl))
        if idx is None:
            import numpy as np
            scale = np.float32(T) / np.float32(tl)
            src = np.minimum(np.floor(np.arange(tl, dtype=np.float32) * scale).astype(np.int64), T - 1)
            idx = self._label_idx[(T, tl)] = torch.from_numpy(src).to(self.dev)
        return z.index_select(0, idx)

    def _col_scale(self, T: int) -> torch.Tensor:
        cs = self._scales.get(T)
        if cs is None:
            cs = self._scales[T] = torch.cat([torch.full((k,), w / (T * k)) for _, k, w in HEADS]).to(self.dev)
        return cs

    def apply_update(self):
        """DDP exchange (one all-reduce of the flat gradient buffer, mean over ranks) + SGD + refresh of the transposed copies"""
        if self._pending:                                   # buckets were reduced during the backward
            for h in self._pending:
                h.wait()
            self._pending = []
            scale = 1.0 / self._ddp_world()
        else:
            scale = allreduce_sum_flat(self.G, self.pg) if getattr(self, "exchange", True) else 1.0   # exchange=False: rank-local step (bench only)
        ops.sgd_step(self.P, self.G, self.lr, self.wd, scale)
        self._refresh_transposed()


def allreduce_sum_flat(flat_grad: torch.Tensor, group=None) -> float:
    """The ONE exchange of a data-parallel step: sum the flat gradient buffer over ranks in place (RCCL on the GPU,
    gloo in the CPU tests) and return the factor that turns the sum into the mean (1/world).  No-op for one rank."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return 1.0
    dist.all_reduce(flat_grad, op=dist.ReduceOp.SUM, group=group)
    return 1.0 / dist.get_world_size(group)


def lr_at_epoch(epoch: int, lr: float, power: float, warmup: int, decay_rate: float) -> float:
    """The schedule of `Temporal_tenco/run.py:341-348` (same in Spatial_cnn): SGD(lr/power) under
    SequentialLR([LinearLR(start_factor=power, total_iters=warmup), ExponentialLR(gamma)], milestones=[warmup+1]);
    the value torch's schedulers hold during epoch `epoch` (0-based)."""
    base = lr / power
    if epoch <= warmup:
        return base * (power + (1.0 - power) * min(epoch, warmup) / warmup)
    return base * decay_rate ** (epoch - warmup - 1)
