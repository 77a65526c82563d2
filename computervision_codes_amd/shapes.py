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


# --------------------------------------------------------------------------- Swin + Q2L (parameters only)
SWIN_CFG = {  # `Spatial_transformer/models/swin_transformer.py:596-631`
    "swin_T_224_1k": dict(embed_dim=96, depths=(2, 2, 6, 2), num_heads=(3, 6, 12, 24), window_size=7),
    "swin_B_224_22k": dict(embed_dim=128, depths=(2, 2, 18, 2), num_heads=(4, 8, 16, 32), window_size=7),
    "swin_B_384_22k": dict(embed_dim=128, depths=(2, 2, 18, 2), num_heads=(4, 8, 16, 32), window_size=12),
    "swin_L_224_22k": dict(embed_dim=192, depths=(2, 2, 18, 2), num_heads=(6, 12, 24, 48), window_size=7),
    "swin_L_384_22k": dict(embed_dim=192, depths=(2, 2, 18, 2), num_heads=(6, 12, 24, 48), window_size=12),
}
# non-parameter state-dict entries (recomputed, never filled): `attn.relative_position_index`, `attn_mask`, `backbone.1.pe`
SWIN_BUFFER_SUFFIXES = ("relative_position_index", "attn_mask", ".pe")


def swin_window(name: str, img_size: int, stage: int):
    """effective (window, resolution) of a stage (`swin_transformer.py:193-196`)"""
    cfg = SWIN_CFG[name]
    res = img_size // 4 // (2 ** stage)
    ws = cfg["window_size"]
    return (min(ws, res), res)


def swin_param_shapes(name: str, img_size: int, prefix: str = "") -> Table:
    """Parameters of `SwinTransformer` without avgpool/head (`backbone.py:198-201`), registration order."""
    cfg = SWIN_CFG[name]
    e = cfg["embed_dim"]
    t: Table = [(prefix + "patch_embed.proj.weight", (e, 3, 4, 4)), (prefix + "patch_embed.proj.bias", (e,)),
                (prefix + "patch_embed.norm.weight", (e,)), (prefix + "patch_embed.norm.bias", (e,))]
    for s, (depth, nh) in enumerate(zip(cfg["depths"], cfg["num_heads"])):
        c = e * 2 ** s
        ws, _ = swin_window(name, img_size, s)
        for b in range(depth):
            p = f"{prefix}layers.{s}.blocks.{b}."
            t += [(p + "norm1.weight", (c,)), (p + "norm1.bias", (c,)),
                  (p + "attn.relative_position_bias_table", ((2 * ws - 1) ** 2, nh)),
                  (p + "attn.qkv.weight", (3 * c, c)), (p + "attn.qkv.bias", (3 * c,)),
                  (p + "attn.proj.weight", (c, c)), (p + "attn.proj.bias", (c,)),
                  (p + "norm2.weight", (c,)), (p + "norm2.bias", (c,)),
                  (p + "mlp.fc1.weight", (4 * c, c)), (p + "mlp.fc1.bias", (4 * c,)),
                  (p + "mlp.fc2.weight", (c, 4 * c)), (p + "mlp.fc2.bias", (c,))]
        if s < 3:
            p = f"{prefix}layers.{s}.downsample."
            t += [(p + "reduction.weight", (2 * c, 4 * c)), (p + "norm.weight", (4 * c,)), (p + "norm.bias", (4 * c,))]
    t += [(prefix + "norm.weight", (8 * e,)), (prefix + "norm.bias", (8 * e,))]
    return t


def q2l_transformer_shapes(prefix: str, d: int, ffn: int = 8192) -> Table:
    """`build_transformer` (`transformer.py:347-359`): 1 encoder layer, 2 decoder layers without self-attn."""
    t: Table = []

    def mha(p):
        t.extend([(p + ".in_proj_weight", (3 * d, d)), (p + ".in_proj_bias", (3 * d,)), (p + ".out_proj.weight", (d, d)),
                  (p + ".out_proj.bias", (d,))])

    def ffn_(p):
        t.extend([(p + ".linear1.weight", (ffn, d)), (p + ".linear1.bias", (ffn,)), (p + ".linear2.weight", (d, ffn)),
                  (p + ".linear2.bias", (d,))])

    def ln(p):
        t.extend([(p + ".weight", (d,)), (p + ".bias", (d,))])

    e = prefix + "encoder.layers.0"
    mha(e + ".self_attn"); ffn_(e); ln(e + ".norm1"); ln(e + ".norm2")
    for i in range(2):
        q = f"{prefix}decoder.layers.{i}"
        mha(q + ".multihead_attn"); ffn_(q); ln(q + ".norm2"); ln(q + ".norm3")
    ln(prefix + "decoder.norm")
    return t


def q2l_param_shapes(backbone: str, img_size: int, hidden_dim: int, loss_type: str, teacher_dim: int = 512) -> Table:
    """`Qeruy2Label` (`Spatial_transformer/network.py:48-80`): backbone.0.* + one Decoder per task.  For `loss_type all` the four
    decoders hold the SAME Transformer object (`network.py:66-73`): its parameters appear once (under decoder_i, the first
    registration, as `named_parameters()` reports them); `q2l_state_dict_aliases` lists the three aliased copies a
    `state_dict()` additionally holds.  KD adaptors wi/wv/wt (student_dim=hidden -> teacher_dim) and mi/mv/mt follow."""
    assert loss_type in ("i", "v", "t", "all")
    kmap = {"i": 6, "v": 10, "t": 15, "ivt": 100}
    c = SWIN_CFG[backbone]["embed_dim"] * 8
    t = swin_param_shapes(backbone, img_size, prefix="backbone.0.")
    tasks = ("i", "v", "t", "ivt") if loss_type == "all" else (loss_type,)
    for n, task in enumerate(tasks):
        p = f"decoder_{task}."
        if n == 0:
            t += q2l_transformer_shapes(p + "transformer.", hidden_dim)
        t += [(p + "input_proj.weight", (hidden_dim, c, 1, 1)), (p + "input_proj.bias", (hidden_dim,)),
              (p + "query_embed.weight", (kmap[task], hidden_dim)), (p + "fc.W", (1, kmap[task], hidden_dim)), (p + "fc.b", (1, kmap[task]))]
    if loss_type == "all":
        for n in ("wi", "wv", "wt"):
            t += [(n + ".weight", (teacher_dim, hidden_dim, 1)), (n + ".bias", (teacher_dim,))]
        for n in ("mi", "mv", "mt"):
            t += [(n + ".weight", (hidden_dim, teacher_dim, 1)), (n + ".bias", (hidden_dim,))]
    return t


def q2l_state_dict_aliases(hidden_dim: int):
    """(alias key, source key) pairs of the shared transformer in a `loss_type all` state dict"""
    out = []
    for k, _ in q2l_transformer_shapes("decoder_i.transformer.", hidden_dim):
        for task in ("v", "t", "ivt"):
            out.append((k.replace("decoder_i.", f"decoder_{task}.", 1), k))
    return out


# --------------------------------------------------------------------------- MS-TCT
def mstct_shapes(in_feat_dim: int = 2048, inter_channels=(256, 384, 576, 864), num_block: int = 2, mlp_ratio: int = 8,
                 final_dim: int = 512, loss_type: str = "ivt") -> Table:
    """`Temporal_mstct/network.py:48-73` VideoNas (TemporalEncoder, Temporal_Mixer, one Classifier)."""
    t: Table = []

    def lin(p, o, i):
        t.extend([(p + ".weight", (o, i)), (p + ".bias", (o,))])

    def ln(p, c):
        t.extend([(p + ".weight", (c,)), (p + ".bias", (c,))])

    cin = in_feat_dim
    for s, c in enumerate(inter_channels, start=1):
        m = f"TemporalEncoder.Temporal_Merging_Block{s}"
        t.extend([(m + ".proj.weight", (c, cin, 3)), (m + ".proj.bias", (c,))])
        ln(m + ".norm", c)
        for b in range(num_block):
            q = f"TemporalEncoder.block{s}.{b}"
            ln(q + ".norm1", c)
            g = q + ".Global_Relational_Block"
            lin(g + ".q", c, c); lin(g + ".kv", 2 * c, c); lin(g + ".proj", c, c)
            ln(q + ".norm2", c)
            l = q + ".Local_Relational_Block"
            lin(l + ".linear1", mlp_ratio * c, c)
            t.extend([(l + ".TC.weight", (mlp_ratio * c, 1, 3)), (l + ".TC.bias", (mlp_ratio * c,))])
            lin(l + ".linear2", c, mlp_ratio * c)
        ln(f"TemporalEncoder.norm{s}", c)
        cin = c
    for i, c in zip((4, 3, 2, 1), reversed(inter_channels)):
        lin(f"Temporal_Mixer.linear_f{i}.proj", final_dim, c)
    for i in range(1, 10):
        t.extend([(f"Temporal_Mixer.linear{i}.weight", (final_dim, final_dim, 1)), (f"Temporal_Mixer.linear{i}.bias", (final_dim,))])
    k = {"i": 6, "v": 10, "t": 15, "ivt": 100}[loss_type]
    q = f"classifier_{loss_type}"
    t.extend([(q + ".linear_fuse.weight", (final_dim, 4 * final_dim, 1)), (q + ".linear_fuse.bias", (final_dim,)),
              (q + ".linear_pred.weight", (k, final_dim, 1)), (q + ".linear_pred.bias", (k,))])
    return t
