"""diagnostic: HIP spatial training step vs the CPU oracle in fp32 and fp64 -- whose gradients are closer to the fp64 truth?
python tools/diag_train2d.py "<cfg dict>" [key substring]"""
import sys, torch, numpy as np, warnings
warnings.filterwarnings("ignore")
sys.path[:0] = ["tests", "."]
from test_gpu_train2d import _inputs, _trainer
from oracle import spatial_cnn_train as o_ct
cfg = eval(sys.argv[1])
sub = sys.argv[2] if len(sys.argv) > 2 else ""
tr, sd, table = _trainer(cfg)
img, labels, tpred, tfeat = _inputs(cfg)
kw = dict(network=cfg["network"], lr=cfg["lr"], weight_decay=1e-5, rates=cfg["rates"], temp=4.0)
_, t32, g32 = o_ct.train_step(sd, img, labels, tpred, tfeat, **kw)
_, t64, g64 = o_ct.train_step_f64(sd, img, labels, tpred, tfeat, **kw)
terms = tr.train_step(img.cuda(), labels, tpred, tfeat, apply_update=False)
g = tr.grads()
print("loss hip/f32/f64", terms["loss"], t32["loss"], t64["loss"])
for k in g:
    if sub not in k:
        continue
    ref = g64[k]
    den = ref.abs().max()
    d = (g[k].double() - ref).abs()
    e_h, e_t = (d.max() / den).item(), ((g32[k].double() - ref).abs().max() / den).item()
    flag = "  <<<" if e_h > 6 * e_t + 2e-5 else ""
    print(f"{k:55s} hip {e_h:.2e} torch32 {e_t:.2e} |g|max {den.item():.2e} argmax-err {np.unravel_index(int(d.argmax()), tuple(d.shape))}{flag}")
