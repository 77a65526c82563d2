"""Oracle for Spatial_cnn (`MT4MTLKD/Spatial_cnn/network.py`) and its ResNet trunk
(`Spatial_transformer/models/resnet.py`, identical graph to torchvision resnet18/50), eval mode.
Test infrastructure only."""
from __future__ import annotations

from typing import Dict

import torch
import torch.nn.functional as F

SD = Dict[str, torch.Tensor]
_DEPTHS = {"resnet18": (2, 2, 2, 2), "resnet50": (3, 4, 6, 3)}


def _bn(sd: SD, p: str, x: torch.Tensor) -> torch.Tensor:
    return F.batch_norm(x, sd[p + ".running_mean"], sd[p + ".running_var"], sd[p + ".weight"], sd[p + ".bias"],
                        training=False, eps=1e-5)


def _bottleneck(sd: SD, p: str, x: torch.Tensor, stride: int) -> torch.Tensor:
    """`resnet.py:101-121` (stride on the 3x3: v1.5)."""
    idt = x
    o = F.relu(_bn(sd, p + "bn1", F.conv2d(x, sd[p + "conv1.weight"])))
    o = F.relu(_bn(sd, p + "bn2", F.conv2d(o, sd[p + "conv2.weight"], stride=stride, padding=1)))
    o = _bn(sd, p + "bn3", F.conv2d(o, sd[p + "conv3.weight"]))
    if (p + "downsample.0.weight") in sd:
        idt = _bn(sd, p + "downsample.1", F.conv2d(x, sd[p + "downsample.0.weight"], stride=stride))
    return F.relu(o + idt)


def _basic(sd: SD, p: str, x: torch.Tensor, stride: int) -> torch.Tensor:
    """`resnet.py:35-72`."""
    idt = x
    o = F.relu(_bn(sd, p + "bn1", F.conv2d(x, sd[p + "conv1.weight"], stride=stride, padding=1)))
    o = _bn(sd, p + "bn2", F.conv2d(o, sd[p + "conv2.weight"], padding=1))
    if (p + "downsample.0.weight") in sd:
        idt = _bn(sd, p + "downsample.1", F.conv2d(x, sd[p + "downsample.0.weight"], stride=stride))
    return F.relu(o + idt)


def resnet_trunk(sd: SD, x: torch.Tensor, arch: str = "resnet50", prefix: str = "") -> torch.Tensor:
    """`ResNet._forward_impl` (`resnet.py:201-214`) up to and including avgpool: [B,3,H,W] -> [B,C,1,1].
    The trunk's 1000-way fc runs in the reference but its output is discarded (`Spatial_cnn/network.py:117`)."""
    x = F.relu(_bn(sd, prefix + "bn1", F.conv2d(x, sd[prefix + "conv1.weight"], stride=2, padding=3)))
    x = F.max_pool2d(x, kernel_size=3, stride=2, padding=1)
    block = _bottleneck if arch == "resnet50" else _basic
    for li, n in enumerate(_DEPTHS[arch], start=1):
        for b in range(n):
            stride = 2 if (b == 0 and li > 1) else 1
            x = block(sd, f"{prefix}layer{li}.{b}.", x, stride)
    return F.adaptive_avg_pool2d(x, (1, 1))


def _r16(t: torch.Tensor) -> torch.Tensor:
    """round to bfloat16 and back (round-to-nearest-even: what `v_cvt_pk_bf16_f32` does)"""
    return t.to(torch.bfloat16).to(torch.float32)


def resnet_trunk_bf16_emulation(sd: SD, x: torch.Tensor, arch: str = "resnet50", prefix: str = "", fuse_downsample: bool = True) -> torch.Tensor:
    """The SAME graph with the roundings of the bf16 throughput mode of the MI355X path put where its kernels put them (and nowhere else), all
    arithmetic in fp32: the normalised frame is rounded to bf16; every convolution's weights are the eval-mode BatchNorm scale folded in fp32 and
    THEN rounded to bf16, its bias (beta - mu * scale) stays fp32; a unit's output -- conv + bias [+ residual], ReLU -- is rounded to bf16 once,
    when it is stored; in the strided Bottlenecks conv3 and the downsample branch are ONE fp32 sum (`fuse_downsample`, the mode's default), so the
    branch is not rounded on its own; max-pool on bf16 values is exact; the pooled feature and the heads are fp32.  What is left between this and
    the kernels is the fp32 summation order inside a convolution (and the rare bf16 rounding tie it flips), so the bf16 mode can be held to ~1e-2
    of the logit range against it instead of the 5e-2 it needs against the fp32 reference -- tight enough to see a wrong tap or a missing
    rounding.  Test infrastructure only."""
    def fold(conv: str, bn: str):
        g, b = sd[bn + ".weight"].double(), sd[bn + ".bias"].double()
        mu, var = sd[bn + ".running_mean"].double(), sd[bn + ".running_var"].double()
        scale = g / torch.sqrt(var + 1e-5)
        w = _r16(sd[conv + ".weight"].float() * scale.float()[:, None, None, None])
        return w, (b - mu * scale).float()

    def unit(xx, conv, bn, stride=1, padding=0, relu=True, add=None, store=True):
        w, b = fold(conv, bn)
        y = F.conv2d(xx, w, stride=stride, padding=padding) + b[None, :, None, None]
        if add is not None:
            y = y + add
        if relu:
            y = F.relu(y)
        return _r16(y) if store else y
    x = unit(_r16(x), prefix + "conv1", prefix + "bn1", stride=2, padding=3)
    x = F.max_pool2d(x, kernel_size=3, stride=2, padding=1)
    for li, n in enumerate(_DEPTHS[arch], start=1):
        for bi in range(n):
            s = 2 if (bi == 0 and li > 1) else 1
            q = f"{prefix}layer{li}.{bi}."
            has_ds = (q + "downsample.0.weight") in sd
            if arch == "resnet50":
                o = unit(x, q + "conv1", q + "bn1")
                o = unit(o, q + "conv2", q + "bn2", stride=s, padding=1)
                if has_ds and fuse_downsample and li > 1:      # one accumulator chain over K = planes + Cin (layer1.0's branch is rounded: fused kernel)
                    idt = unit(x, q + "downsample.0", q + "downsample.1", stride=s, relu=False, store=False)
                else:
                    idt = unit(x, q + "downsample.0", q + "downsample.1", stride=s, relu=False) if has_ds else x
                x = unit(o, q + "conv3", q + "bn3", add=idt)
            else:
                idt = unit(x, q + "downsample.0", q + "downsample.1", stride=s, relu=False) if has_ds else x
                o = unit(x, q + "conv1", q + "bn1", stride=s, padding=1)
                x = unit(o, q + "conv2", q + "bn2", padding=1, add=idt)
    return F.adaptive_avg_pool2d(x, (1, 1))


def spatial_cnn_forward(sd: SD, img: torch.Tensor, network: str = "resnet50", loss_type: str = "all", emulate_bf16: bool = False):
    """`VideoNas.forward` (`Spatial_cnn/network.py:45-92`) in eval (`args.train` False): the KD branch is
    skipped and its three slots are the integer 0.  `emulate_bf16`: the trunk with the bf16 mode's roundings (`resnet_trunk_bf16_emulation`)."""
    high = (resnet_trunk_bf16_emulation if emulate_bf16 else resnet_trunk)(sd, img, network, prefix="basemodel.basemodel.")
    feat = high.squeeze(-1).squeeze(-1)
    b = feat.shape[0]
    flat = torch.flatten(high, 1)
    logits = {}
    for task, k in (("ivt", 100), ("i", 6), ("v", 10), ("t", 15)):
        if loss_type in (task, "all"):
            logits[task] = F.linear(flat, sd[f"classifier_{task}.fc.weight"], sd[f"classifier_{task}.fc.bias"])
        else:
            logits[task] = torch.zeros((b, k))
    return (0, logits["i"]), (0, logits["v"]), (0, logits["t"]), (feat, logits["ivt"])


def kd_branch(sd: SD, s: torch.Tensor, tool: torch.Tensor, verb: torch.Tensor, target: torch.Tensor):
    """Train-time KD mixing (`Spatial_cnn/network.py:47-71`), in its algebraically reduced form.

    The reference stacks C copies of s to a [B,C,C] tensor and contracts it with the three projected
    teacher features; row c of that product is s[b,c]/sqrt(C) * sum_d tea_n[b,d].  The softmax runs
    over the 3 teachers."""
    c = s.shape[1]
    teas = [F.conv1d(t.unsqueeze(-1), sd[f"{m}.weight"], sd[f"{m}.bias"]).squeeze(-1)
            for m, t in (("mi", tool), ("mv", verb), ("mt", target))]
    tsum = torch.stack([t.sum(dim=1) for t in teas], dim=-1)                # [B,3]
    attn = torch.softmax((s / (c ** 0.5)).unsqueeze(-1) * tsum.unsqueeze(1), dim=-1)   # [B,C,3]
    outs = []
    for n, w in enumerate(("wi", "wv", "wt")):
        outs.append(F.conv1d((s * attn[:, :, n]).unsqueeze(-1), sd[f"{w}.weight"], sd[f"{w}.bias"]).squeeze(-1))
    return tuple(outs)
