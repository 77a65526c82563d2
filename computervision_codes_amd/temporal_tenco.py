"""Temporal_tenco on MI355X: host-side mirror of `MT4MTLKD/Temporal_tenco/network.py` (VideoNas).

Same constructor arguments, state-dict keys and return structure as the reference module; the
arithmetic runs in libmt4hip.so.  Activations stay frame-major ([T][C], the layout of the frame
feature files and of the module's own input [1,T,D]) -- the reference's permute to [1,D,T]
(`network.py:42`) is folded into addressing, and the [1,K,T] logits it returns are permuted VIEWS
of our [T][K] buffers (same shape and values, different strides).

One DilatedResidualLayer (`network.py:186-198`) = two dependent launches: dilated k3 conv + bias + ReLU, then 1x1 conv + bias +
residual.  Short videos (B*T <= TCN_PATH_MAX_ROWS[dtype]) run the latency path (`mt4_tcn_stage` / `mt4_tcn_conv`, csrc/tcn_kernels.hip: comb
tiles, one workgroup per 32 frames x 16 channels, no barrier in the K loop); long ones the implicit-GEMM kernel with its large tiles.
The four 1x1 heads (`network.py:21-24`) are concatenated into one [131][C] GEMM; on the latency path the three FPN laterals run as
one launch over the stacked stage outputs and the heads as one launch over the four levels.
"""
from __future__ import annotations

from typing import Dict, List

import torch

from . import ops
from .shapes import tenco_shapes

# rows (B*T) up to which a forward takes the latency path: above it the 128 x 128 tiles of the implicit-GEMM kernel move fewer operand
# bytes per FLOP than 32 x 16.  Measured crossover, 4-stage head, hipGraph replay, ms (latency path | implicit GEMM):
#   fp32  T=512 0.89 | 0.96   T=768 1.34 | 1.53   T=1024 1.60 | 1.54        bf16  T=512 0.57 | 0.64   T=768 0.85 | 0.72
TCN_PATH_MAX_ROWS = {torch.float32: 896, torch.bfloat16: 640}


class VideoNas:
    """Drop-in for `Temporal_tenco.network.VideoNas` (inference path; eval semantics).

    args needs: fpn, output, hier.  output=False is what every shipped script uses and the only form the reference itself can run at the
    shipped widths (True raises here, naming the lines); hier=True (AvgPool1d(7, 3) behind every refinement stage, linear re-interpolation
    in the FPN: `network.py:147,154-155,96`; set by no shipped script) runs on the implicit-GEMM path.
    """

    def __init__(self, args, num_layers_PG, num_layers_R, num_R, num_f_maps, dim, num_classes, num_i=6, num_v=10,
                 num_t=15, device: str = "cuda", dtype: torch.dtype = torch.float32, path: str = "auto"):
        """dtype float32: the parity mode (exact-fp32 MFMA chain); bfloat16: throughput mode (bf16 weights / activations, fp32
        accumulate and epilogue, fp32 logits).  path: "auto" (by B*T), "tcn" (latency path), "igemm" (implicit-GEMM kernel)"""
        self.dtype = dtype
        assert path in ("auto", "tcn", "igemm")
        self.path = path
        if getattr(args, "output", False):
            raise NotImplementedError("--output is never set by the shipped scripts (Scripts/*.sh) and cannot run in the reference at the shipped widths "
                                      "(Refinement.conv_1x1 takes num_classes channels, network.py:141, and is handed the num_f_maps-channel "
                                      "feature, :58,150-151)")
        self.hier = bool(getattr(args, "hier", False))
        self.args = args
        self.use_fpn = bool(getattr(args, "fpn", False))
        self.num_layers_PG, self.num_layers_R, self.num_R = num_layers_PG, num_layers_R, num_R
        self.C, self.D, self.K = num_f_maps, dim, num_classes
        self.head_sizes = (num_classes, num_i, num_v, num_t)
        self.device = torch.device(device)
        self.training = False
        self._table = tenco_shapes(num_layers_PG, num_layers_R, num_R, num_f_maps, dim, num_classes, fpn=self.use_fpn,
                                   num_i=num_i, num_v=num_v, num_t=num_t)
        self._sd: Dict[str, torch.Tensor] = {}
        self._p: Dict[str, torch.Tensor] = {}
        self.tile = 0            # implicit-GEMM path: 0 = the library's choice per launch; a tile id forces it (tools/tcn_long_sweep.py)
        # bf16, 512 channels: the window of 64-frame tile counts (B x ceil(T / 64)) in which a layer runs as ONE launch.  Below it the chip is not
        # filled (one tile per CU), above it the generic 256 x 256 tiles amortise the weights better than 2 MB per 64 rows through L2
        # (profiles/r04_tcn_fused_layer_ab.txt: 24 ... 64 videos of 256 frames +2 ... +20 %, 8 whole videos of 2000 frames 2.98 -> 1.98 ms; 128 videos: even)
        self.fused_layer_min_tiles, self.fused_layer_max_tiles = 96, 384

    # ------------------------------------------------------------------ nn.Module-like surface
    def eval(self):
        self.training = False
        return self

    def cuda(self):
        return self

    def state_dict(self) -> Dict[str, torch.Tensor]:
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
                self._sd[k] = sd[k].detach().to(torch.float32)
        self._pack()
        return self

    def _pack(self):
        dev, p = self.device, {}

        def pk(name):  # Conv1d weight [Cout,Cin,k] -> OIHW [Cout,Cin,1,k] -> packed
            w = self._sd[name + ".weight"].to(dev)
            p[name + ".w"] = ops.pack_conv_weight(w.unsqueeze(2), None, self.dtype)
            p[name + ".b"] = self._sd[name + ".bias"].to(dev).contiguous()

        stages = [("PG", self.num_layers_PG)] + [(f"Rs.{r}", self.num_layers_R) for r in range(self.num_R)]
        pk("PG.conv_1x1")
        for prefix, n in stages:
            for i in range(n):
                pk(f"{prefix}.layers.{i}.conv_dilated")
                pk(f"{prefix}.layers.{i}.conv_1x1")
                if self.dtype == torch.bfloat16 and self.C == 512:      # fragment-ordered copies for the one-launch layer of the throughput mode
                    for c in ("conv_dilated", "conv_1x1"):
                        p[f"{prefix}.layers.{i}.{c}.wf"] = ops.pack_fragments(p[f"{prefix}.layers.{i}.{c}.w"])
        if not self.use_fpn:
            pk("PG.conv_out")
        else:
            pk("fpn.latlayer1")
            w = torch.cat([self._sd[f"conv_out{s}.weight"] for s in ("", "_i", "_v", "_t")], 0).to(dev)
            b = torch.cat([self._sd[f"conv_out{s}.bias"] for s in ("", "_i", "_v", "_t")], 0).to(dev)
            p["heads.w"] = ops.pack_conv_weight(w.unsqueeze(2), None, self.dtype)
            p["heads.b"] = b.contiguous()
        self._p = p
        self._stages = {prefix: ops.TcnStage([p[f"{prefix}.layers.{i}.conv_dilated.w"] for i in range(n)],
                                             [p[f"{prefix}.layers.{i}.conv_dilated.b"] for i in range(n)],
                                             [p[f"{prefix}.layers.{i}.conv_1x1.w"] for i in range(n)],
                                             [p[f"{prefix}.layers.{i}.conv_1x1.b"] for i in range(n)]) for prefix, n in stages}

    # ------------------------------------------------------------------ forward
    def _c1(self, x, name, residual=None, relu=False):
        return ops.conv_nhwc(x, self._p[name + ".w"], self._p[name + ".b"], kh=1, kw=1, residual=residual, relu=relu, tile=self.tile)

    def _layer(self, x, prefix, d):
        p = self._p
        b, _, t, _ = x.shape
        if (prefix + ".conv_dilated.wf") in p and not self.hier and self.fused_layer_min_tiles <= b * ((t + 63) // 64) <= self.fused_layer_max_tiles:
            # throughput mode (several videos per forward): the whole DilatedResidualLayer in ONE launch, the 512-channel hidden map stays in LDS
            # (`ops.tcn_layer_fused`, bit-identical to the two launches below; gated on the tile count: one 64-frame tile per CU)
            return ops.tcn_layer_fused(x.view(b, t, self.C), p[prefix + ".conv_dilated.wf"], p[prefix + ".conv_dilated.b"], p[prefix + ".conv_1x1.wf"],
                                       p[prefix + ".conv_1x1.b"], d).view(b, 1, t, self.C)
        h = ops.conv_nhwc(x, p[prefix + ".conv_dilated.w"], p[prefix + ".conv_dilated.b"], kh=1, kw=3, pad=(0, d), dil=(1, d),
                          relu=True, tile=self.tile)
        return ops.conv_nhwc(h, p[prefix + ".conv_1x1.w"], p[prefix + ".conv_1x1.b"], kh=1, kw=1, residual=x, tile=self.tile)

    def _stage(self, x, prefix, n):
        for i in range(n):
            x = self._layer(x, f"{prefix}.layers.{i}", 2 ** i)
        return x

    def _use_tcn_path(self, rows: int) -> bool:
        ok = ops.tcn_supported(self.C, self.dtype) and ops.tcn_supported(self.D, self.dtype)
        if self.path == "tcn":
            if not ok:
                raise ValueError("latency path needs channel counts that are whole 128-byte K-steps")
            return True
        return self.path == "auto" and ok and rows <= TCN_PATH_MAX_ROWS[self.dtype]

    def _forward_tcn(self, x: torch.Tensor):
        """latency path: x [B,T,D] of self.dtype -> the reference's return structure"""
        b, t, _ = x.shape
        p = self._p
        as_ref = lambda y: y.permute(0, 2, 1)                     # [B,T,K] -> [B,K,T] view
        f0 = ops.tcn_conv(x, p["PG.conv_1x1.w"], p["PG.conv_1x1.b"], taps=1)
        out_list: List[torch.Tensor] = []
        out_i: List[torch.Tensor] = []
        out_v: List[torch.Tensor] = []
        out_t: List[torch.Tensor] = []
        if not self.use_fpn:
            f = ops.tcn_stage(self._stages["PG"], f0)
            out_list.append(as_ref(ops.tcn_conv(f, p["PG.conv_out.w"], p["PG.conv_out.b"], taps=1, out_dtype=torch.float32)))
            f_list = [f]
            for r in range(self.num_R):
                f = ops.tcn_stage(self._stages[f"Rs.{r}"], f)
                f_list.append(f)
        else:
            # the stage outputs c1..c3, p4 live in one [4,B,T,C] buffer: the three laterals (all `latlayer1`, network.py:103-105) are ONE
            # launch over the rows of c1..c3, and the heads one launch over the four levels
            nlev = self.num_R + 1
            cs = torch.empty((nlev, b, t, self.C), dtype=self.dtype, device=x.device)
            ops.tcn_stage(self._stages["PG"], f0, out=cs[0])
            for r in range(self.num_R):
                ops.tcn_stage(self._stages[f"Rs.{r}"], cs[r], out=cs[r + 1])
            if nlev > 1:
                lat = ops.tcn_conv(cs[:nlev - 1].view(1, (nlev - 1) * b * t, self.C), p["fpn.latlayer1.w"], p["fpn.latlayer1.b"], taps=1)
                ops.fpn_topdown(lat.view(nlev - 1, b * t * self.C), cs.view(nlev, b * t * self.C))   # p_l = lat(c_l) + p_{l+1}
            y = ops.tcn_conv(cs.view(1, nlev * b * t, self.C), p["heads.w"], p["heads.b"], taps=1,
                             out_dtype=torch.float32).view(nlev, b, t, -1)
            k0, k1, k2, k3 = self.head_sizes
            f_list = [cs[l] for l in range(nlev)]
            for l in range(nlev):
                yl = as_ref(y[l])
                out_list.append(yl[:, :k0])
                out_i.append(yl[:, k0:k0 + k1])
                out_v.append(yl[:, k0 + k1:k0 + k1 + k2])
                out_t.append(yl[:, k0 + k1 + k2:])
        f_ref = [as_ref(ff) for ff in f_list]
        return out_list, out_i, out_v, out_t, f_ref, f_ref

    def _forward_hier(self, x: torch.Tensor):
        """`--hier True`: every refinement stage ends in AvgPool1d(7, 3) (`network.py:147,154-155`), so level l has its own length T_l and the
        FPN's `F.interpolate(x, size=W, mode='linear')` (:96) resamples.  x [B,T,D] of self.dtype"""
        b, t, _ = x.shape
        p = self._p
        as_ref = lambda y, tl: y.view(b, tl, -1).permute(0, 2, 1)
        f = self._stage(self._c1(x.view(b, 1, t, self.D), "PG.conv_1x1"), "PG", self.num_layers_PG)
        levels = [(f, t)]
        out_list: List[torch.Tensor] = []
        out_i: List[torch.Tensor] = []
        out_v: List[torch.Tensor] = []
        out_t: List[torch.Tensor] = []
        if not self.use_fpn:
            out_list.append(as_ref(ops.conv_nhwc(f, p["PG.conv_out.w"], p["PG.conv_out.b"], kh=1, kw=1, out_dtype=torch.float32), t))
        tl = t
        for r in range(self.num_R):
            if tl < 7:
                raise ValueError(f"--hier: level {r + 1} would pool {tl} < 7 frames (AvgPool1d(7, 3))")
            g = self._stage(f, f"Rs.{r}", self.num_layers_R)
            f = ops.avgpool1d_rows(g.view(b, tl, self.C), 7, 3)
            tl = f.shape[1]
            f = f.view(b, 1, tl, self.C)
            levels.append((f, tl))
        if self.use_fpn:
            # p_l = interpolate(p_{l+1} -> T_l) + latlayer1(c_l)   (`FPN.forward`, :98-106: latlayer1 serves every lateral)
            top, ttop = levels[-1]
            ps = [(top, ttop)]
            for c, tc in reversed(levels[:-1]):
                up = ops.interp_linear_rows(top.view(b, ttop, self.C), tc).view(b, 1, tc, self.C)
                top, ttop = self._c1(c, "fpn.latlayer1", residual=up), tc
                ps.append((top, ttop))
            levels = ps[::-1]
            k0, k1, k2, k3 = self.head_sizes
            for lvl, tlv in levels:
                y = as_ref(ops.conv_nhwc(lvl, p["heads.w"], p["heads.b"], kh=1, kw=1, out_dtype=torch.float32), tlv)
                out_list.append(y[:, :k0])
                out_i.append(y[:, k0:k0 + k1])
                out_v.append(y[:, k0 + k1:k0 + k1 + k2])
                out_t.append(y[:, k0 + k1 + k2:])
        f_ref = [as_ref(ff, tlv) for ff, tlv in levels]
        return out_list, out_i, out_v, out_t, f_ref, f_ref

    @ops.with_latency_tiles
    def forward(self, x: torch.Tensor, ismask: bool = False):
        """x [B,T,D] float32 on the GPU.  Returns (out_list, out_list_i, out_list_v, out_list_t, f_list, f_list)
        with tensors shaped like the reference's ([B,K,T] logits, [B,C,T] features)."""
        if ismask and self.training:
            raise NotImplementedError("train-time masking (network.py:43-48) is not part of the inference path")
        if not self._p:
            raise RuntimeError("load_state_dict first")
        assert x.dim() == 3 and x.shape[2] == self.D and x.dtype == torch.float32
        b, t, _ = x.shape
        if self.hier:
            return self._forward_hier(x.contiguous().to(self.dtype))
        if self._use_tcn_path(b * t):
            return self._forward_tcn(x.contiguous().to(self.dtype))
        x4 = x.contiguous().to(self.dtype).view(b, 1, t, self.D)
        f = self._stage(self._c1(x4, "PG.conv_1x1"), "PG", self.num_layers_PG)
        f_list = [f]
        out_list: List[torch.Tensor] = []
        out_i: List[torch.Tensor] = []
        out_v: List[torch.Tensor] = []
        out_t: List[torch.Tensor] = []
        as_ref = lambda y: y.view(b, t, -1).permute(0, 2, 1)  # [B,1,T,K] -> [B,K,T] view
        if not self.use_fpn:
            out_list.append(as_ref(ops.conv_nhwc(f, self._p["PG.conv_out.w"], self._p["PG.conv_out.b"], kh=1, kw=1, out_dtype=torch.float32)))
        # per-stage conv_out logits are computed and discarded by the reference (network.py:133,160): skipped
        for r in range(self.num_R):
            f = self._stage(f, f"Rs.{r}", self.num_layers_R)
            f_list.append(f)
        if self.use_fpn:
            c1, c2, c3, p4 = f_list
            p3 = self._c1(c3, "fpn.latlayer1", residual=p4)   # interpolate to equal length == identity
            p2 = self._c1(c2, "fpn.latlayer1", residual=p3)
            p1 = self._c1(c1, "fpn.latlayer1", residual=p2)
            f_list = [p1, p2, p3, p4]
            k0, k1, k2, k3 = self.head_sizes
            for lvl in f_list:
                y = as_ref(ops.conv_nhwc(lvl, self._p["heads.w"], self._p["heads.b"], kh=1, kw=1, out_dtype=torch.float32))
                out_list.append(y[:, :k0])
                out_i.append(y[:, k0:k0 + k1])
                out_v.append(y[:, k0 + k1:k0 + k1 + k2])
                out_t.append(y[:, k0 + k1 + k2:])
        f_ref = [as_ref(ff) for ff in f_list]
        return out_list, out_i, out_v, out_t, f_ref, f_ref

    __call__ = forward
