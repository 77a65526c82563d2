"""diagnostic: does the spatial training step read memory it never wrote?  Poison the caching allocator with NaN / large values, then step."""
import sys, torch, numpy as np, warnings
warnings.filterwarnings("ignore")
sys.path[:0] = ["tests", "."]
from test_gpu_train2d import _inputs, _trainer
cfg = dict(network="resnet18", B=4, H=64, W=64, seed=601, lr=0.05, rates=(1.0, 1.0, 1.0))
img, labels, tpred, tfeat = _inputs(cfg)
def run(poison):
    if poison is not None:
        junk = [torch.full((n,), poison, device="cuda") for n in (1 << 26, 1 << 24, 1 << 22, 1 << 20, 1 << 18, 1 << 16, 1 << 14, 1 << 12, 1 << 10) for _ in range(4)]
        del junk
    tr, sd, table = _trainer(cfg)
    terms = tr.train_step(img.cuda(), labels, tpred, tfeat, apply_update=False)
    return terms, tr.grads()
t0, g0 = run(None)
for poison in (float("nan"), 1e3):
    t1, g1 = run(poison)
    print("poison", poison, "loss", t0["loss"], t1["loss"])
    for k in g0:
        d = (g0[k] - g1[k]).abs().max().item() / max(g0[k].abs().max().item(), 1e-30)
        if not d < 1e-4:
            print(f"  {k:55s} rel diff {d:.2e}")
