#!/usr/bin/env python3
"""What the bf16 throughput modes that bench.py quotes do to the discrete readouts: Swin-B/384 `loss_type all` (BASELINE configs[2]) and the
Temporal_tenco head in bf16 against the reference goldens and against the fp32 parity mode on more inputs.  Prints the measured rates; the
floors asserted in tests/test_gpu_models.py sit below them."""
import os, sys, types
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import torch
from computervision_codes_amd import shapes, synth
from conftest import load_golden

cuda = torch.device("cuda:0")


def agreement(a, ref):
    a, ref = a.float().cpu(), torch.as_tensor(np.asarray(ref)).float()
    k = min(5, a.shape[1])
    am = float((a.argmax(1) == ref.argmax(1)).float().mean())
    t5 = float((a.topk(k, 1).indices.sort(1).values == ref.topk(k, 1).indices.sort(1).values).all(1).float().mean())
    return am, t5


def q2l():
    from computervision_codes_amd.spatial_transformer import build_q2l
    z, cfg = load_golden("q2l_swinB_384_all")
    args = types.SimpleNamespace(backbone=cfg["backbone"], img_size=cfg["img"], hidden_dim=cfg["hidden"], loss_type="all")
    sd = synth.fill_from_shapes(shapes.q2l_param_shapes(cfg["backbone"], cfg["img"], cfg["hidden"], "all"), seed=cfg["seed"])
    m16 = build_q2l(args, dtype=torch.bfloat16).eval().load_state_dict(sd)
    m32 = build_q2l(args, dtype=torch.float32).eval().load_state_dict(sd)
    frames = synth.synthetic_frames(cfg["B"], cfg["img"], cfg["img"], seed=cfg["seed"]).to(cuda)
    tf = [synth.synthetic_features(cfg["B"], 512, seed=cfg["seed"] + k)[0].to(cuda) for k in (1, 2, 3)]
    out = m16(frames, *tf)
    for gi, key in enumerate(("logit_i", "logit_v", "logit_t", "logit_ivt")):
        ref = z[key]
        err = float((out[gi][1].float().cpu() - torch.from_numpy(ref)).abs().max())
        print(f"q2l_swinB_384_all bf16 vs golden {key}: max err {err:.4f} = {err / np.abs(ref).max():.4f} of range {np.abs(ref).max():.3f}; argmax/top5 {agreement(out[gi][1], ref)}")
    err = float((out[3][0].float().cpu() - torch.from_numpy(z["feat"])).abs().max())
    print(f"  feat: max err {err:.4f} of range {np.abs(z['feat']).max():.3f}")
    n = 24
    fr = synth.synthetic_frames(n, cfg["img"], cfg["img"], seed=77).to(cuda)
    tfn = [synth.synthetic_features(n, 512, seed=80 + k)[0].to(cuda) for k in (1, 2, 3)]
    o16 = m16(fr, *tfn)
    o32 = [torch.cat([m32(fr[s:s + 8], *[t[s:s + 8] for t in tfn])[gi][1] for s in range(0, n, 8)]) for gi in range(4)]
    for gi, key in enumerate(("i", "v", "t", "ivt")):
        l32 = o32[gi].float().cpu()
        top2 = l32.topk(2, 1).values
        print(f"  {n} frames bf16 vs fp32 mode head {key}: argmax/top5 {agreement(o16[gi][1], l32.numpy())}  max err {float((o16[gi][1].float().cpu() - l32).abs().max()):.4f}"
              f"  range {float(l32.abs().max()):.3f}  median top-1 margin {float((top2[:, 0] - top2[:, 1]).median()):.4f}")


def tenco():
    from computervision_codes_amd.temporal_tenco import VideoNas
    for name in ("tenco_config1", "tenco_4stage", "tenco_ragged"):
        try:
            z, cfg = load_golden(name)
        except Exception as e:
            print(name, "no golden", e)
            continue
        args = types.SimpleNamespace(fpn=cfg["fpn"], output=False, hier=False, mask=True)
        m = VideoNas(args, cfg["num_layers_PG"], cfg["num_layers_R"], cfg["num_R"], cfg["num_f_maps"], cfg["dim"], 100, dtype=torch.bfloat16).eval()
        m.load_state_dict(synth.fill_from_shapes(shapes.tenco_shapes(cfg["num_layers_PG"], cfg["num_layers_R"], cfg["num_R"], cfg["num_f_maps"], cfg["dim"],
                                                                     100, fpn=cfg["fpn"]), seed=cfg["seed"]))
        x = synth.synthetic_features(cfg["T"], cfg["dim"], seed=cfg["seed"]).to(cuda)
        out = m(x, False)
        for gi, g in enumerate(("ivt", "i", "v", "t")):
            for li, o in enumerate(out[gi]):
                ref = z[f"logit_{g}_{li}"]                     # [1,K,T]
                a, r = o.float().cpu()[0].t(), torch.from_numpy(ref)[0].t()          # [T,K]
                err = float((a - r).abs().max())
                top2 = r.topk(2, 1).values
                print(f"{name} bf16 vs golden logit_{g}_{li}: err {err:.4f} = {err / float(r.abs().max()):.4f} of range; per-frame argmax/top5 {agreement(a, r.numpy())}"
                      f"  median margin {float((top2[:, 0] - top2[:, 1]).median()):.4f}")


if __name__ == "__main__":
    tenco()
    q2l()
