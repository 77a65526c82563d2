"""Ordered (key, shape) tables of the reference models' state dicts.

The key names and their order are the checkpoint contract (SURVEY 5 "Checkpoint / resume", Appendix
A): a ``.pth`` written by the reference loads here and vice versa.  `oracle/gen_golden.py` asserts,
in the container that has the reference, that every table equals ``module.state_dict()`` of the
corresponding reference module (names, order and shapes).
"""
from __future__ import annotations

from typing import List, Tuple

Shape = Tuple[int, ...]
Table = List[Tuple[str, Shape]]


# --------------------------------------------------------------------------- Temporal_tenco
def tenco_shapes(num_layers_PG: int = 11, num_layers_R: int = 10, num_R: int = 3, num_f_maps: int = 512,
                 dim: int = 512, num_classes: int = 100, fpn: bool = True,
                 num_i: int = 6, num_v: int = 10, num_t: int = 15) -> Table:
    """`Temporal_tenco/network.py:14-34` VideoNas (PG, conv_out*, Rs, fpn) in registration order."""
    t: Table = []

    def conv1d(prefix: str, cout: int, cin: int, k: int):
        t.append((prefix + ".weight", (cout, cin, k)))
        t.append((prefix + ".bias", (cout,)))

    def stage(prefix: str, n_layers: int, in_dim: int):
        conv1d(prefix + ".conv_1x1", num_f_maps, in_dim, 1)
        for i in range(n_layers):
            conv1d(f"{prefix}.layers.{i}.conv_dilated", num_f_maps, num_f_maps, 3)
            conv1d(f"{prefix}.layers.{i}.conv_1x1", num_f_maps, num_f_maps, 1)
        conv1d(prefix + ".conv_out", num_classes, num_f_maps, 1)

    stage("PG", num_layers_PG, dim)
    conv1d("conv_out", num_classes, num_f_maps, 1)
    conv1d("conv_out_i", num_i, num_f_maps, 1)
    conv1d("conv_out_v", num_v, num_f_maps, 1)
    conv1d("conv_out_t", num_t, num_f_maps, 1)
    for r in range(num_R):
        stage(f"Rs.{r}", num_layers_R, num_classes)
    if fpn:
        for i in (1, 2, 3):
            conv1d(f"fpn.latlayer{i}", num_f_maps, num_f_maps, 1)
    return t


# --------------------------------------------------------------------------- ResNet trunk
_RESNET_CFG = {
    "resnet18": ("basic", (2, 2, 2, 2)),
    "resnet50": ("bottleneck", (3, 4, 6, 3)),
}


def _bn(t: Table, prefix: str, c: int):
    t.append((prefix + ".weight", (c,)))
    t.append((prefix + ".bias", (c,)))
    t.append((prefix + ".running_mean", (c,)))
    t.append((prefix + ".running_var", (c,)))
    t.append((prefix + ".num_batches_tracked", ()))


def resnet_shapes(arch: str = "resnet50", prefix: str = "") -> Table:
    """torchvision ResNet v1.5 state dict (`Spatial_transformer/models/resnet.py:124-220`)."""
    kind, depths = _RESNET_CFG[arch]
    exp = 4 if kind == "bottleneck" else 1
    t: Table = []
    t.append((prefix + "conv1.weight", (64, 3, 7, 7)))
    _bn(t, prefix + "bn1", 64)
    inplanes = 64
    for li, (planes, nblocks) in enumerate(zip((64, 128, 256, 512), depths), start=1):
        for b in range(nblocks):
            stride = 2 if (b == 0 and li > 1) else 1
            p = f"{prefix}layer{li}.{b}."
            if kind == "bottleneck":
                t.append((p + "conv1.weight", (planes, inplanes, 1, 1)))
                _bn(t, p + "bn1", planes)
                t.append((p + "conv2.weight", (planes, planes, 3, 3)))
                _bn(t, p + "bn2", planes)
                t.append((p + "conv3.weight", (planes * 4, planes, 1, 1)))
                _bn(t, p + "bn3", planes * 4)
            else:
                t.append((p + "conv1.weight", (planes, inplanes, 3, 3)))
                _bn(t, p + "bn1", planes)
                t.append((p + "conv2.weight", (planes, planes, 3, 3)))
                _bn(t, p + "bn2", planes)
            if b == 0 and (stride != 1 or inplanes != planes * exp):
                t.append((p + "downsample.0.weight", (planes * exp, inplanes, 1, 1)))
                _bn(t, p + "downsample.1", planes * exp)
            inplanes = planes * exp
    t.append((prefix + "fc.weight", (1000, 512 * exp)))
    t.append((prefix + "fc.bias", (1000,)))
    return t


def resnet_feat_dim(arch: str) -> int:
    return 2048 if arch == "resnet50" else 512


def spatial_cnn_shapes(network: str = "resnet50", student_dim: int | None = None, teacher_dim: int = 1536,
                       loss_type: str = "all") -> Table:
    """`Spatial_cnn/network.py:13-44` VideoNas: trunk under ``basemodel.basemodel.``, KD adaptors, heads."""
    c = student_dim if student_dim is not None else resnet_feat_dim(network)
    t = resnet_shapes(network, prefix="basemodel.basemodel.")
    if loss_type == "all":
        for n in ("wi", "wv", "wt"):
            t.append((n + ".weight", (teacher_dim, c, 1)))
            t.append((n + ".bias", (teacher_dim,)))
        for n in ("mi", "mv", "mt"):
            t.append((n + ".weight", (c, teacher_dim, 1)))
            t.append((n + ".bias", (c,)))
    for task, k in (("i", 6), ("v", 10), ("t", 15), ("ivt", 100)):
        if loss_type in (task, "all"):
            t.append((f"classifier_{task}.fc.weight", (k, c)))
            t.append((f"classifier_{task}.fc.bias", (k,)))
    return t
