#!/usr/bin/env python3
"""MS-TCT training step, fp32 against bf16 GEMM operands: fixture error and step time (GPU box)"""
import os, sys, ast
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from computervision_codes_amd import shapes, synth
from computervision_codes_amd.mstct_train import MstctTrainer
dev = torch.device("cuda:0")
z = np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "mstct_train_full.npz"))
cfg = ast.literal_eval(str(z["cfg"]))
table = shapes.mstct_shapes(cfg["D"], cfg["inter"], 2, 8, cfg["final"], cfg["loss_type"])
sd = synth.fill_from_shapes(table, seed=cfg["seed"])
x = torch.cat([synth.synthetic_features(cfg["T"], cfg["D"], seed=cfg["seed"] + b) for b in range(cfg["B"])], 0).permute(0, 2, 1).contiguous()
k = {"i": 6, "v": 10, "t": 15, "ivt": 100}[cfg["loss_type"]]
y = torch.from_numpy((synth.uniform01(cfg["seed"], 800, cfg["B"] * cfg["T"] * k) < 0.15).reshape(cfg["B"], cfg["T"], k).astype(np.int64))
for dt in (torch.float32, torch.bfloat16):
    tr = MstctTrainer(cfg["inter"], 2, 8, 8, cfg["D"], cfg["final"], cfg["loss_type"], lr=cfg["lr"], weight_decay=1e-5, operand_dtype=dt).load_state_dict(sd)
    loss = tr.train_step(x.to(dev), y, apply_update=False)
    g = tr.grads()
    ref = z["grad_norms"]
    rel = np.array([abs(float(g[kk].norm()) - r) / max(r, 1e-6 * ref.max()) for (kk, _), r in zip(table, ref)])
    print(dt, "loss", loss, "ref", float(z["loss"]), "grad-norm rel err median %.2e p90 %.2e max %.2e" % (np.median(rel), np.percentile(rel, 90), rel.max()), flush=True)
B, T, D = 31, 256, 1536
for dt in (torch.float32, torch.bfloat16):
    tr = MstctTrainer((256, 384, 576, 864), 2, 8, 8, D, 512, "i", lr=0.01, operand_dtype=dt).load_state_dict(
        synth.fill_from_shapes(shapes.mstct_shapes(D, (256, 384, 576, 864), 2, 8, 512, "i"), seed=47))
    xb = torch.randn(B, T, D, device=dev)
    zb = (torch.rand(B * T, 6, device=dev) < 0.15).float()
    masks = tr.draw_masks_device(B, T, 1, 0)
    ms = bench._time_call(lambda: tr.train_step_btd(xb, zb, masks=masks), iters=5)
    msg = bench._time_call(lambda: tr.train_step_btd(xb, zb, masks=masks, use_graph=True), iters=5)
    print(dt, f"b31 T256 D1536: eager {ms:.2f} ms, graph {msg:.2f} ms", flush=True)
