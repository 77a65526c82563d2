"""Spatial_cnn on MI355X: host-side mirror of `MT4MTLKD/Spatial_cnn/network.py` (VideoNas / BaseModel /
Classifier) with the torchvision ResNet-18/50 trunk (`Spatial_transformer/models/resnet.py` graph).

Same state-dict keys (`basemodel.basemodel.*`, `classifier_*`, `wi/wv/wt/mi/mv/mt`) and the same return
tuple as the reference.  Every Conv2d+BatchNorm2d(eval)[+ReLU][+residual] is ONE launch of the
implicit-GEMM kernel (BN folded at load time), activations are NHWC in `dtype` (float32 for the 1e-3
parity mode, bfloat16 for throughput), heads run in fp32 on the fp32 pooled feature.
"""
from __future__ import annotations

from typing import Dict, Optional

import torch

from . import ops
from .shapes import resnet_feat_dim, spatial_cnn_shapes
from .synth import IMAGENET_MEAN, IMAGENET_STD

_DEPTHS = {"resnet18": (2, 2, 2, 2), "resnet50": (3, 4, 6, 3)}
_HEADS = (("i", 6), ("v", 10), ("t", 15), ("ivt", 100))


class VideoNas:
    """Drop-in for `Spatial_cnn.network.VideoNas` (eval / extraction path).

    args needs: network ('resnet18'|'resnet50'), loss_type, student_dim, teacher_dim, train."""

    def __init__(self, args=None, num_tool=6, num_verb=10, num_target=15, num_triplet=100, dtype: torch.dtype = torch.float32,
                 device: str = "cuda"):
        self.args = args
        self.network = args.network
        self.loss_type = args.loss_type
        self.feat_dim = getattr(args, "student_dim", None) or resnet_feat_dim(self.network)
        assert self.feat_dim == resnet_feat_dim(self.network), "student_dim must equal the trunk's feature width"
        self.dtype = dtype
        self.device = torch.device(device)
        self.training = False
        self._table = spatial_cnn_shapes(self.network, self.feat_dim, getattr(args, "teacher_dim", 1536), self.loss_type)
        self._sd: Dict[str, torch.Tensor] = {}
        self._p: Dict[str, object] = {}
        self._streams = []
        # fused launches of the bf16 ResNet-50 trunk (every one bit-identical to the launches it replaces except fuse_downsample, which keeps one
        # fp32 accumulator chain where the two-launch form rounds the branch to bf16; attributes so that tests and A/B tools can switch them):
        self.fuse_stem_pool = True      # stem conv + max-pool in one launch (uint8-frame path)
        self.fuse_expand = True         # layer2's stride-1 blocks: conv2 + conv3 in one launch (large batches; used when a block is not chained)
        self.fuse_next_block = True     # layer2.0's conv1 behind the last layer1 block, in its launch
        self.fuse_downsample = True     # strided Bottlenecks: conv3 + downsample branch as one GEMM
        self.fuse_bottleneck = True     # layer1 Bottlenecks in one launch each
        # conv3 (+ bn3 + add + ReLU) of an identity Bottleneck and conv1 (+ bn1 + ReLU) of the block behind it in ONE launch (`ops.chain_gemm`,
        # K-chunk accumulation: the 4 x planes map is written once and not read back): the layers it is used in, gated on the row count
        self.chain_layers = (2, 3)

    def train(self, mode: bool = True):
        self.training = bool(mode)
        return self

    def _train_engine(self):
        """the forward of train mode lives in `SpatialCnnTrainer` (batch-statistics BatchNorm kernels, KD branch); built on first use from
        this module's parameters"""
        if getattr(self, "_trainer", None) is None:
            from .spatial_cnn_train import SpatialCnnTrainer
            self._trainer = SpatialCnnTrainer(self.network, teacher_dim=int(getattr(self.args, "teacher_dim", 1536)), loss_type=self.loss_type,
                                              device=str(self.device))
            self._trainer.exchange = False
            self._trainer.load_state_dict(self.state_dict())
        return self._trainer

    def eval(self):
        self.training = False
        return self

    def cuda(self):
        return self

    def state_dict(self):
        return dict(self._sd)

    def load_state_dict(self, sd: Dict[str, torch.Tensor], strict: bool = True):
        names = [k for k, _ in self._table]
        missing = [k for k in names if k not in sd]
        if strict and (missing or len(sd) != len(names)):
            raise KeyError(f"state dict mismatch: missing {missing[:4]}, unexpected {[k for k in sd if k not in names][:4]}")
        for k, shp in self._table:
            if k in sd:
                if tuple(sd[k].shape) != tuple(shp):
                    raise ValueError(f"{k}: shape {tuple(sd[k].shape)} != {shp}")
                self._sd[k] = sd[k].detach()
        self._trainer = None          # (the train-mode engine is rebuilt from the new parameters on first use)
        self._pack()
        return self

    # ------------------------------------------------------------------ load-time packing (BN folding)
    def _fold(self, conv: str, bn: str, stem: bool = False):
        sd, dev = self._sd, self.device
        g = sd[bn + ".weight"].double()
        beta = sd[bn + ".bias"].double()
        mu = sd[bn + ".running_mean"].double()
        var = sd[bn + ".running_var"].double()
        scale = g / torch.sqrt(var + 1e-5)
        bias = (beta - mu * scale).float().to(dev).contiguous()
        w = sd[conv + ".weight"].float().to(dev)
        sc = scale.float().to(dev)
        wp = ops.pack_stem_weight(w, sc, self.dtype) if stem else ops.pack_conv_weight(w, sc, self.dtype)
        return wp, bias

    def _pack(self):
        pre = "basemodel.basemodel."
        p: Dict[str, object] = {}
        p["stem"] = self._fold(pre + "conv1", pre + "bn1", stem=True)
        if self.dtype == torch.bfloat16:   # space-to-depth stem (even frame sizes): one 128-byte run per kernel row, LDS-DMA path
            sd, dev = self._sd, self.device
            scale = (sd[pre + "bn1.weight"].double() / torch.sqrt(sd[pre + "bn1.running_var"].double() + 1e-5)).float().to(dev)
            p["stem_s2d"] = ops.stem_s2d_weight(sd[pre + "conv1.weight"].float().to(dev), scale)
        bottleneck = self.network == "resnet50"
        for li, n in enumerate(_DEPTHS[self.network], start=1):
            for b in range(n):
                q = f"{pre}layer{li}.{b}."
                for ci in ((1, 2, 3) if bottleneck else (1, 2)):
                    p[f"{q}conv{ci}"] = self._fold(f"{q}conv{ci}", f"{q}bn{ci}")
                if (q + "downsample.0.weight") in self._sd:
                    p[q + "ds"] = self._fold(q + "downsample.0", q + "downsample.1")
                if bottleneck and li == 1 and self.dtype == torch.bfloat16:   # the block's weights in the fused kernel's fragment order
                    p[q + "fused"] = ops.bottleneck_pack(p[q + "conv1"], p[q + "conv2"], p[q + "conv3"], p.get(q + "ds"))
                if bottleneck and li > 1 and (q + "ds") in p and self.dtype == torch.bfloat16:   # conv3 and the downsample branch as ONE GEMM: K ranges back to back
                    (w3, b3), (wd, bd) = p[q + "conv3"], p[q + "ds"]
                    p[q + "conv3ds"] = (torch.cat([w3, wd], 1).contiguous(), (b3 + bd).contiguous())
                    if li == 2:      # its conv1 rides behind the last layer1 block (ops.bottleneck_fused_next)
                        p[q + "conv1next"] = ops.bottleneck_pack_next(p[q + "conv1"])
                if bottleneck and li in (2, 3) and (q + "ds") not in p and self.dtype == torch.bfloat16:   # conv3 behind conv2 in one launch (ops.conv3x3_expand) / in front of the next conv1 (ops.chain_gemm)
                    p[q + "conv3frag"] = ops.pack_fragments(p[q + "conv3"][0])
                if bottleneck and li in (2, 3, 4) and self.dtype == torch.bfloat16:
                    p[q + "conv1frag"] = ops.pack_fragments(p[q + "conv1"][0])
        ws, bs, self._head_slices, o = [], [], {}, 0
        for task, k in _HEADS:
            if self.loss_type in (task, "all"):
                ws.append(self._sd[f"classifier_{task}.fc.weight"].float())
                bs.append(self._sd[f"classifier_{task}.fc.bias"].float())
                self._head_slices[task] = (o, o + k)
                o += k
        # the heads run as one fp32 GEMM (exact fp32 MFMA) through the 1x1 mode of the conv kernel
        p["heads.w"] = ops.pack_conv_weight(torch.cat(ws, 0).to(self.device)[:, :, None, None], None, torch.float32)
        p["heads.b"] = torch.cat(bs, 0).to(self.device).contiguous()
        if self.loss_type == "all":   # KD adaptors (`network.py:27-33`), fp32 like the heads
            for n in ("wi", "wv", "wt", "mi", "mv", "mt"):
                p[n] = (ops.pack_linear_weight(self._sd[n + ".weight"].float().to(self.device), torch.float32), self._sd[n + ".bias"].float().to(self.device).contiguous())
        self._p = p

    # ------------------------------------------------------------------ trunk
    def _conv(self, x, key, k, stride=1, pad=0, residual=None, relu=True, out=None):
        wp, b = self._p[key]
        return ops.conv_nhwc(x, wp, b, kh=k, kw=k, stride=(stride, stride), pad=(pad, pad), residual=residual, relu=relu, out=out)

    def _stem(self, xp: torch.Tensor, h: int, w: int) -> torch.Tensor:
        b, hp, wp_, _ = xp.shape
        wst, bst = self._p["stem"]
        ho, wo = (h - 1) // 2 + 1, (w - 1) // 2 + 1
        xv = xp.view(b, hp, wp_ // 2, 8)  # pixel pairs: one 7-wide kernel row = 4 pairs (one zero slot)
        y = torch.empty((b, ho, wo, wst.shape[0]), dtype=self.dtype, device=xp.device)
        ops.conv_nhwc(xv, wst, bst, kh=7, kw=4, stride=(2, 1), relu=True, out=y)
        return ops.maxpool3x3s2(y)

    def _layers(self, x: torch.Tensor, first: int, last: int, out: Optional[torch.Tensor] = None) -> torch.Tensor:
        """residual layers first..last (1-based, inclusive); the last conv writes into `out` when given"""
        pre = "basemodel.basemodel."
        bottleneck = self.network == "resnet50"
        pending = None          # conv1 output of the block about to run, when the previous block's conv3 launch produced it
        x_is_even = False       # x holds the previous block's output at the even pixels only (ops.bottleneck_fused_next)
        for li in range(first, last + 1):
            n = _DEPTHS[self.network][li - 1]
            for bi in range(n):
                q = f"{pre}layer{li}.{bi}."
                s = 2 if (bi == 0 and li > 1) else 1
                o_buf = out if (li == last and bi == n - 1) else None
                if (bottleneck and li == 1 and self.fuse_bottleneck and self.dtype == torch.bfloat16 and pending is None and (q + "fused") in self._p):
                    # conv1 -> conv2 -> conv3 (+ downsample) of a 64-channel stride-1 block in one launch, intermediates in LDS (bit-identical)
                    nq = f"{pre}layer2.0.conv1next"
                    if bi == n - 1 and li < last and self.fuse_downsample and self.fuse_next_block and nq in self._p:
                        # ... and the next block's conv1 on the result while it is in LDS; the map itself is kept at the even pixels only
                        # (its other reader is the stride-2 downsample branch)
                        x, pending = ops.bottleneck_fused_next(x, self._p[q + "fused"], self._p[nq])
                        x_is_even = True
                    else:
                        x = ops.bottleneck_fused(x, self._p[q + "fused"], out=o_buf)
                    continue
                if bottleneck and (q + "conv3ds") in self._p and self.fuse_downsample:
                    # conv3 and the downsample branch in ONE accumulator chain (K = planes + Cin): no identity map written and read back
                    o = self._conv(pending if pending is not None else self._conv(x, q + "conv1", 1), q + "conv2", 3, stride=s, pad=1)
                    pending = None
                    wcat, bcat = self._p[q + "conv3ds"]
                    x = ops.conv_nhwc(o, wcat, bcat, kh=1, kw=1, relu=True, out=o_buf, second=(x, 1 if x_is_even else s))
                    x_is_even = False
                    continue
                idt = self._conv(x, q + "ds", 1, stride=s, relu=False) if (q + "ds") in self._p else x
                if bottleneck:  # resnet.py:101-121 (stride on the 3x3)
                    o = pending if pending is not None else self._conv(x, q + "conv1", 1)
                    pending = None
                    nq = f"{pre}layer{li}.{bi + 1}." if bi + 1 < n else (f"{pre}layer{li + 1}.0." if li < last else None)
                    if (li in self.chain_layers and nq is not None and o_buf is None and self.dtype == torch.bfloat16 and (q + "conv3frag") in self._p
                            and (nq + "conv1frag") in self._p and ops.chain_gemm_pays(o.shape[0] * ((o.shape[1] - 1) // s + 1) * ((o.shape[2] - 1) // s + 1))
                            and ops.chain_gemm_supported(self._p[q + "conv3frag"].shape[1], self._p[q + "conv3frag"].shape[0], self._p[nq + "conv1frag"].shape[0], True)):
                        # conv2, then conv3 + add + ReLU and the NEXT block's conv1 + ReLU in one launch: this block's output map is written once
                        # (the next block's residual) and never read back
                        o = self._conv(o, q + "conv2", 3, stride=s, pad=1)
                        bb, hh, ww, cc = o.shape
                        y1, t1 = ops.chain_gemm(o.view(-1, cc), self._p[q + "conv3frag"], self._p[q + "conv3"][1], self._p[nq + "conv1frag"],
                                                self._p[nq + "conv1"][1], r1=idt)
                        x, pending = y1.view(bb, hh, ww, -1), t1.view(bb, hh, ww, -1)
                        continue
                    if (q + "conv3frag") in self._p and li == 2 and self.fuse_expand:
                        # conv2 and conv3 (+ residual) of a layer2 identity block in one launch: the 128-channel map stays in LDS (bit-identical;
                        # None where the patch kernel does not run -- small batches)
                        (w2, b2), b3 = self._p[q + "conv2"], self._p[q + "conv3"][1]
                        y = ops.conv3x3_expand(o, w2, b2, self._p[q + "conv3frag"], b3, idt, out=o_buf)
                        if y is not None:
                            x = y
                            continue
                    o = self._conv(o, q + "conv2", 3, stride=s, pad=1)
                    x = self._conv(o, q + "conv3", 1, residual=idt, out=o_buf)
                else:           # resnet.py:35-72
                    o = self._conv(x, q + "conv1", 3, stride=s, pad=1)
                    x = self._conv(o, q + "conv2", 3, pad=1, residual=idt, out=o_buf)
        return x

    def trunk_from_padded(self, xp: torch.Tensor, h: int, w: int) -> torch.Tensor:
        """xp: stem input [B,H+6,Wp,4] (ops.preprocess_u8 / ops.pad_nchw).  Returns pooled fp32 [B,C].
        (Running the HBM-bound head -- stem, max-pool, layer1[, layer2] -- in sub-batches of 64-167 frames so that producer ->
        consumer -> residual stays inside the 256 MB Infinity Cache was measured 4-15 % SLOWER at 1336 frames, twice; the whole
        batch goes through every layer.)"""
        return ops.global_avgpool(self._layers(self._stem(xp, h, w), 1, 4))

    def conv_plan(self, h: int, w: int):
        """Geometry of every conv launch of one forward, in launch order (algorithmic dims: the stem is the
        true 7x7x3).  Used by bench.py for FLOP accounting and per-layer tables."""
        plan = []
        ho, wo = (h - 1) // 2 + 1, (w - 1) // 2 + 1
        plan.append(dict(name="stem", Cin=3, Cout=64, kh=7, kw=7, stride=2, Ho=ho, Wo=wo))
        hh, ww = (ho - 1) // 2 + 1, (wo - 1) // 2 + 1
        cin = 64
        bottleneck = self.network == "resnet50"
        for li, (planes, n) in enumerate(zip((64, 128, 256, 512), _DEPTHS[self.network]), start=1):
            for bi in range(n):
                s = 2 if (bi == 0 and li > 1) else 1
                h2, w2 = (hh - 1) // s + 1, (ww - 1) // s + 1
                cout = planes * (4 if bottleneck else 1)
                q = f"layer{li}.{bi}."
                if bi == 0 and (s != 1 or cin != cout):
                    plan.append(dict(name=q + "ds", Cin=cin, Cout=cout, kh=1, kw=1, stride=s, Ho=h2, Wo=w2))
                if bottleneck:
                    plan.append(dict(name=q + "conv1", Cin=cin, Cout=planes, kh=1, kw=1, stride=1, Ho=hh, Wo=ww))
                    plan.append(dict(name=q + "conv2", Cin=planes, Cout=planes, kh=3, kw=3, stride=s, Ho=h2, Wo=w2))
                    plan.append(dict(name=q + "conv3", Cin=planes, Cout=cout, kh=1, kw=1, stride=1, Ho=h2, Wo=w2))
                else:
                    plan.append(dict(name=q + "conv1", Cin=cin, Cout=planes, kh=3, kw=3, stride=s, Ho=h2, Wo=w2))
                    plan.append(dict(name=q + "conv2", Cin=planes, Cout=planes, kh=3, kw=3, stride=1, Ho=h2, Wo=w2))
                cin, hh, ww = cout, h2, w2
        return plan

    def launch_groups(self, h: int, w: int, batch: Optional[int] = None):
        """indices into `conv_plan(h, w)` per kernel launch of one forward at bench batch sizes (or at `batch` frames: the chained launches are
        gated on the row count like `_layers` gates them, `ops.chain_gemm_pays`), in launch order (the walk of `_layers`): one conv
        per launch, except (bf16 ResNet-50) the layer1 Bottlenecks (one `mt4_bottleneck_fused_bf16` launch each, the last one carrying layer2.0's
        conv1), conv3 + downsample of the strided blocks (one GEMM), conv2 + conv3 of layer2's identity blocks (`mt4_conv_desc.fuse_expand`) and
        conv3 + the next block's conv1 in the layers of `chain_layers` (`mt4_chain_gemm_bf16`)"""
        plan = self.conv_plan(h, w)
        idx = {rec["name"]: i for i, rec in enumerate(plan)}
        r50 = self.network == "resnet50" and self.dtype == torch.bfloat16
        if not r50:
            return [[i] for i in range(len(plan))]
        fused1 = self.fuse_bottleneck
        ds_fused = self.fuse_downsample
        groups = [[idx["stem"]]]
        pending = False                      # the conv1 of the block about to run was part of the previous launch
        depths = _DEPTHS[self.network]
        planes = (64, 128, 256, 512)
        for li, n in enumerate(depths, start=1):
            for bi in range(n):
                q = f"layer{li}.{bi}."
                has_ds = (q + "ds") in idx
                nq = f"layer{li}.{bi + 1}." if bi + 1 < n else (f"layer{li + 1}.0." if li < 4 else None)
                if li == 1 and fused1:
                    g = ([idx[q + "ds"]] if has_ds else []) + [idx[q + "conv1"], idx[q + "conv2"], idx[q + "conv3"]]
                    if bi == n - 1 and ds_fused and self.fuse_next_block:
                        g.append(idx["layer2.0.conv1"])
                        pending = True
                    groups.append(g)
                    continue
                if has_ds and ds_fused:
                    if not pending:
                        groups.append([idx[q + "conv1"]])
                    pending = False
                    groups += [[idx[q + "conv2"]], [idx[q + "conv3"], idx[q + "ds"]]]
                    continue
                if has_ds:
                    groups.append([idx[q + "ds"]])
                if not pending:
                    groups.append([idx[q + "conv1"]])
                pending = False
                nplanes = planes[li - 1] if bi + 1 < n else (planes[li] if li < 4 else 0)
                rows = None if batch is None else batch * plan[idx[q + "conv3"]]["Ho"] * plan[idx[q + "conv3"]]["Wo"]
                if (li in self.chain_layers and nq is not None and not has_ds and li in (2, 3) and (rows is None or ops.chain_gemm_pays(rows))
                        and ops.chain_gemm_supported(planes[li - 1], 4 * planes[li - 1], nplanes, True)):
                    groups += [[idx[q + "conv2"]], [idx[q + "conv3"], idx[nq + "conv1"]]]
                    pending = True
                elif li == 2 and not has_ds and self.fuse_expand:
                    groups.append([idx[q + "conv2"], idx[q + "conv3"]])
                else:
                    groups += [[idx[q + "conv2"]], [idx[q + "conv3"]]]
        return groups

    def _finish(self, feat: torch.Tensor):
        b = feat.shape[0]
        logits = ops.conv_nhwc(feat.view(b, 1, 1, -1), self._p["heads.w"], self._p["heads.b"], kh=1, kw=1).view(b, -1)
        outs = {}
        for task, k in _HEADS:
            if task in self._head_slices:
                lo, hi = self._head_slices[task]
                outs[task] = logits[:, lo:hi]
            else:  # network.py:79-82: zeros for absent heads
                outs[task] = torch.zeros((b, k), device=feat.device)
        return (0, outs["i"]), (0, outs["v"]), (0, outs["t"]), (feat, outs["ivt"])

    def forward(self, inputs: torch.Tensor, tool=None, verb=None, target=None):
        """inputs: normalised float32 NCHW [B,3,H,W] on the GPU (the reference's module boundary)."""
        if not self._p:
            raise RuntimeError("load_state_dict first")
        kd = self.loss_type == "all" and getattr(self.args, "train", False)     # `network.py:47`: gated by args.train, not by the module mode
        if kd and (tool is None or verb is None or target is None):
            raise TypeError("args.train with loss_type 'all' runs the KD branch (network.py:47-71): pass tool, verb, target features")
        if self.training:
            # `model.train(); model(img, feat_i, feat_v, feat_t)` (`run.py:152-157`): BatchNorm on batch statistics.  fp32, through the trainer's
            # forward kernels; running statistics advance and are written back to this module
            tr = self._train_engine()
            out = tr.forward_train(inputs, (tool, verb, target) if kd else None)
            self._sd.update(tr.running_stats())
            self._pack()
            return out
        _, _, h, w = inputs.shape
        out = self._finish(self.trunk_from_padded(ops.pad_nchw(inputs, self.dtype), h, w))
        if kd:   # the validation loop of `run.py:231-256` calls the module in eval mode with the teacher features: the branch still runs
            feat = out[3][0]
            teas = [ops.linear(t.to(feat.device, torch.float32).contiguous(), *self._p[m]) for m, t in zip(("mi", "mv", "mt"), (tool, verb, target))]
            cams = [ops.linear(mx, *self._p[w_]) for mx, w_ in zip(ops.kd_mix(feat, *teas), ("wi", "wv", "wt"))]
            out = tuple((c, o[1]) for c, o in zip(cams, out[:3])) + (out[3],)
        return out

    __call__ = forward

    def _trunk_u8(self, frames_u8: torch.Tensor) -> torch.Tensor:
        _, h, w, _ = frames_u8.shape
        if "stem_s2d" in self._p and h % 2 == 0 and w % 2 == 0:
            xs = ops.preprocess_u8_s2d(frames_u8, IMAGENET_MEAN, IMAGENET_STD)
            if self.fuse_stem_pool and ops.stem_maxpool_supported(h, w):   # conv1 / bn1 / relu / maxpool in one launch (bit-identical)
                return ops.global_avgpool(self._layers(ops.stem_maxpool(xs, self._p["stem_s2d"], self._p["stem"][1]), 1, 4))
            y = ops.conv_nhwc(xs, self._p["stem_s2d"], self._p["stem"][1], kh=4, kw=1, relu=True, run_pixels=4, out_hw=(h // 2, w // 2))
            return ops.global_avgpool(self._layers(ops.maxpool3x3s2(y), 1, 4))
        return self.trunk_from_padded(ops.preprocess_u8(frames_u8, IMAGENET_MEAN, IMAGENET_STD, self.dtype), h, w)

    def extract_u8(self, frames_u8: torch.Tensor, streams: int = 1):
        """Fused input path: uint8 frames [B,H,W,3] -> (ToTensor+Normalize on the GPU) -> forward.
        streams > 1: the batch is cut into that many contiguous parts which run on their own HIP streams -- frames are independent,
        and the HBM-bound layers of one part co-run with the MFMA-bound layers of another (+3 % frames/s with 2 parts of 1336 frames);
        the parts' results are byte-identical to the single-stream call."""
        _, h, w, _ = frames_u8.shape
        if streams <= 1 or frames_u8.shape[0] < 2 * streams:
            return self._finish(self._trunk_u8(frames_u8))
        main = torch.cuda.current_stream()
        if len(self._streams) < streams:
            self._streams += [torch.cuda.Stream() for _ in range(streams - len(self._streams))]
        b = frames_u8.shape[0]
        cuts = [b * i // streams for i in range(streams + 1)]
        feats = []
        for i in range(streams):
            st = self._streams[i]
            st.wait_stream(main)
            with torch.cuda.stream(st):
                part = frames_u8[cuts[i]:cuts[i + 1]]
                f = self._trunk_u8(part)
            feats.append(f)
        for i in range(streams):
            main.wait_stream(self._streams[i])
            feats[i].record_stream(main)
        return self._finish(torch.cat(feats, 0))
