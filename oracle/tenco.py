"""Oracle for Temporal_tenco (`MT4MTLKD/Temporal_tenco/network.py`), eval mode.  Test infrastructure only."""
from __future__ import annotations

from typing import Dict, List, Tuple

import torch
import torch.nn.functional as F

SD = Dict[str, torch.Tensor]


def dilated_residual_layer(sd: SD, prefix: str, x: torch.Tensor, dilation: int) -> torch.Tensor:
    """`network.py:186-198`: x + conv_1x1(relu(conv_dilated(x))); dropout is identity in eval."""
    h = F.conv1d(x, sd[prefix + ".conv_dilated.weight"], sd[prefix + ".conv_dilated.bias"],
                 padding=dilation, dilation=dilation)
    h = F.relu(h)
    h = F.conv1d(h, sd[prefix + ".conv_1x1.weight"], sd[prefix + ".conv_1x1.bias"])
    return x + h


def _stage(sd: SD, prefix: str, x: torch.Tensor, n_layers: int, project: bool, pool: bool = False) -> Tuple[torch.Tensor, torch.Tensor]:
    """`BaseCausalTCN.forward` (:118-135, project=True) / `Refinement.forward` (:149-162 with
    use_output False -> no input projection; pool = args.hier: nn.AvgPool1d(7, 3) (:147,154-155) behind the layers, the stage's
    feature AND its logits come from the pooled map)."""
    out = F.conv1d(x, sd[prefix + ".conv_1x1.weight"], sd[prefix + ".conv_1x1.bias"]) if project else x
    for i in range(n_layers):
        out = dilated_residual_layer(sd, f"{prefix}.layers.{i}", out, 2 ** i)
    if pool:
        out = F.avg_pool1d(out, kernel_size=7, stride=3)
    logits = F.conv1d(out, sd[prefix + ".conv_out.weight"], sd[prefix + ".conv_out.bias"])
    return out, logits


def tenco_forward(sd: SD, x: torch.Tensor, num_layers_PG: int = 11, num_layers_R: int = 10, num_R: int = 3,
                  fpn: bool = True, hier: bool = False):
    """`VideoNas.forward` (:36-68) with ismask False, args.output False; hier = args.hier (every refinement stage ends in
    AvgPool1d(7, 3), so the levels have different lengths and the FPN's linear interpolation (:96) is a real resampling).

    x: [B,T,D].  Returns (out_list, out_list_i, out_list_v, out_list_t, f_list, f_list) like the reference.
    """
    x = x.permute(0, 2, 1)
    out_list: List[torch.Tensor] = []
    out_i: List[torch.Tensor] = []
    out_v: List[torch.Tensor] = []
    out_t: List[torch.Tensor] = []
    f, out1 = _stage(sd, "PG", x, num_layers_PG, project=True)
    f_list = [f]
    if not fpn:
        out_list.append(out1)
    for r in range(num_R):
        f, out1 = _stage(sd, f"Rs.{r}", f, num_layers_R, project=False, pool=hier)
        f_list.append(f)
    if fpn:
        # `FPN.forward` (:98-106): latlayer1 serves all three laterals; the linear interpolate to an
        # equal length (:96) is an exact identity, kept here as the actual call.
        w, b = sd["fpn.latlayer1.weight"], sd["fpn.latlayer1.bias"]
        c1, c2, c3, p4 = f_list

        def up_add(xx, yy):
            return F.interpolate(xx, size=yy.shape[-1], mode="linear") + yy

        p3 = up_add(p4, F.conv1d(c3, w, b))
        p2 = up_add(p3, F.conv1d(c2, w, b))
        p1 = up_add(p2, F.conv1d(c1, w, b))
        f_list = [p1, p2, p3, p4]
        for f in f_list:
            out_list.append(F.conv1d(f, sd["conv_out.weight"], sd["conv_out.bias"]))
            out_i.append(F.conv1d(f, sd["conv_out_i.weight"], sd["conv_out_i.bias"]))
            out_v.append(F.conv1d(f, sd["conv_out_v.weight"], sd["conv_out_v.bias"]))
            out_t.append(F.conv1d(f, sd["conv_out_t.weight"], sd["conv_out_t.bias"]))
    return out_list, out_i, out_v, out_t, f_list, f_list
