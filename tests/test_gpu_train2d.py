"""GPU: the Spatial_cnn training step (train-mode ResNet + KD branch + hard/soft/KD losses + SGD, `Spatial_cnn/run.py:145-224`)
in HIP vs (a) fixtures captured from the REFERENCE VideoNas in train() mode + torch autograd + torch.optim.SGD and (b) the CPU
oracle; plus the individual training kernels against torch fp32 on the CPU."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from computervision_codes_amd import shapes, synth
from conftest import load_golden

pytestmark = pytest.mark.gpu


def _inputs(cfg):
    img = synth.normalize_frames(synth.synthetic_frames(cfg["B"], cfg["H"], cfg["W"], seed=cfg["seed"]))
    labels = [torch.from_numpy((synth.uniform01(cfg["seed"], 700 + i, cfg["B"] * k) < 0.15).reshape(cfg["B"], k).astype(np.int64))
              for i, k in enumerate((6, 10, 15, 100))]
    tpred = [synth.synthetic_features(cfg["B"], k, seed=cfg["seed"] + 10 + i)[0] * 2.0 for i, k in enumerate((6, 10, 15))]
    tfeat = [synth.synthetic_features(cfg["B"], 1536, seed=cfg["seed"] + 20 + i)[0] for i in range(3)]
    return img, labels, tpred, tfeat


def _rand(shape, seed, scale=1.0):
    n = int(np.prod(shape))
    return torch.from_numpy(((synth.uniform01(seed, 1, n) * 2 - 1) * scale).astype(np.float32).reshape(shape))


# ------------------------------------------------------------------------------------------------ kernels
@pytest.mark.parametrize("m,c,relu,res", [(200, 64, True, False), (1031, 96, True, True), (77, 256, False, False)])
def test_batchnorm_train_fwd_bwd(cuda, m, c, relu, res):
    from computervision_codes_amd import ops
    x, g, b = _rand((m, c), 1, 2.0) + 0.3, _rand((c,), 2) + 1.5, _rand((c,), 3)
    r = _rand((m, c), 4) if res else None
    dy = _rand((m, c), 5)
    rm, rv = _rand((c,), 6), _rand((c,), 7).abs() + 0.5
    xt, gt, bt = x.clone().requires_grad_(), g.clone().requires_grad_(), b.clone().requires_grad_()
    rt = r.clone().requires_grad_() if res else None
    rm_t, rv_t = rm.clone(), rv.clone()
    y_ref = F.batch_norm(xt, rm_t, rv_t, gt, bt, training=True, momentum=0.1, eps=1e-5)
    if res:
        y_ref = y_ref + rt
    if relu:
        y_ref = F.relu(y_ref)
    y_ref.backward(dy)
    xd, rmd, rvd = x.to(cuda), rm.to(cuda), rv.to(cuda)
    mean, invstd = ops.bn_stats(xd, rmd, rvd)
    y = ops.bn_apply(xd, mean, invstd, g.to(cuda), b.to(cuda), r.to(cuda) if res else None, relu)
    assert (y.cpu() - y_ref.detach()).abs().max() < 2e-5
    assert (rmd.cpu() - rm_t).abs().max() < 1e-5 and (rvd.cpu() - rv_t).abs().max() < 1e-5
    dg, db = torch.empty(c, device=cuda), torch.empty(c, device=cuda)
    dx, dres = ops.bn_backward(dy.to(cuda), y if relu else None, xd, mean, invstd, g.to(cuda), dg, db, relu=relu, want_dres=res)
    tol = lambda ref: 2e-5 * max(1.0, ref.abs().max().item())
    assert (dx.cpu() - xt.grad).abs().max() < tol(xt.grad)
    assert (dg.cpu() - gt.grad).abs().max() < tol(gt.grad) * 5 and (db.cpu() - bt.grad).abs().max() < tol(bt.grad) * 5
    if res:
        assert (dres.cpu() - rt.grad).abs().max() < 1e-6
    if relu and not res:   # the gate recomputed from x (the forward's own fp32 expression) == the gate read from the stored output, bit for bit
        dg2, db2 = torch.zeros(c, device=cuda), torch.zeros(c, device=cuda)
        dx2, _ = ops.bn_backward(dy.to(cuda), None, xd, mean, invstd, g.to(cuda), dg2, db2, relu=True, beta=b.to(cuda))
        assert torch.equal(dx2, dx) and torch.equal(dg2, dg) and torch.equal(db2, db)


@pytest.mark.parametrize("b,h,w,cin,cout,k,s,p", [(2, 14, 18, 64, 96, 3, 1, 1), (3, 16, 12, 64, 128, 3, 2, 1), (2, 16, 16, 128, 256, 1, 2, 0),
                                                  (2, 9, 11, 256, 64, 1, 1, 0), (2, 38, 38, 4, 64, 7, 2, 0),
                                                  (2, 120, 112, 64, 136, 3, 1, 1), (2, 184, 184, 160, 64, 1, 1, 0), (3, 150, 150, 64, 256, 1, 1, 0), (2, 132, 120, 128, 256, 3, 2, 1)])
# (the last four: enough pixels for the wide-tile kernels -- 128 x 128 with ragged channel tiles, 64 x 128, 128 x 64, 128 x 128 strided)
def test_conv2d_weight_and_data_gradients(cuda, b, h, w, cin, cout, k, s, p):
    from computervision_codes_amd import ops
    from computervision_codes_amd.spatial_cnn_train import SpatialCnnTrainer, _Unit
    x, wt = _rand((b, cin, h, w), 11), _rand((cout, cin, k, k), 12, 0.1)
    xt, wtt = x.clone().requires_grad_(), wt.clone().requires_grad_()
    y = F.conv2d(xt, wtt, None, s, p)
    dy = _rand(tuple(y.shape), 13)
    y.backward(dy)
    f32 = torch.float32
    u = _Unit()
    u.name, u.bn, u.cin, u.cout, u.k, u.stride, u.pad = "c", "b", cin, cout, k, s, p
    u.w = ops.pack_conv_weight(wt.to(cuda), None, f32)
    u.gw = torch.empty_like(u.w)
    u.wt = u.phase_w = None
    xd, dyd = x.permute(0, 2, 3, 1).contiguous().to(cuda), dy.permute(0, 2, 3, 1).contiguous().to(cuda)
    ops.wgrad_conv2d(dyd, xd, u.gw, k, k, (s, s), (p, p))
    got = u.gw[:, :k * k * cin].view(cout, k, k, cin).permute(0, 3, 1, 2).cpu()
    ref = wtt.grad
    assert (got - ref).abs().max() <= 3e-5 * ref.abs().max(), (got - ref).abs().max() / ref.abs().max()
    if cin == 4:
        return
    tr = SpatialCnnTrainer.__new__(SpatialCnnTrainer)
    tr.units, tr.lin, tr._row_maps, tr.dev = {"c": u}, {}, {}, cuda
    tr._refresh_transposed()
    for residual in (None, _rand((b, h, w, cin), 14).to(cuda)):
        dx = tr._dgrad(u, dyd, xd.shape, residual)
        want = xt.grad.permute(0, 2, 3, 1) + (residual.cpu() if residual is not None else 0)
        assert (dx.cpu() - want).abs().max() <= 3e-5 * want.abs().max()


def test_pool_backward_and_losses(cuda):
    from computervision_codes_amd import ops
    x = _rand((2, 8, 13, 10), 21)
    x = (x * 4).round() / 4          # ties inside windows: the gradient must go to the first maximum, like torch
    xt = x.clone().requires_grad_()
    y = F.max_pool2d(xt, 3, 2, 1)
    dy = _rand(tuple(y.shape), 22)
    y.backward(dy)
    dx = ops.maxpool3x3s2_bwd(x.permute(0, 2, 3, 1).contiguous().to(cuda), dy.permute(0, 2, 3, 1).contiguous().to(cuda))
    assert (dx.cpu().permute(0, 3, 1, 2) - xt.grad).abs().max() < 1e-6
    df = _rand((3, 16), 23)
    assert (ops.avgpool_bwd(df.to(cuda), 3, 5, 16).cpu() - (df / 5)[:, None, :].expand(3, 5, 16)).abs().max() < 1e-7
    # BCE with pos_weight
    yl, z, pw = _rand((5, 12), 24, 3.0), (_rand((5, 12), 25) > 0.5).float(), _rand((12,), 26).abs() + 0.5
    yt = yl.clone().requires_grad_()
    l = F.binary_cross_entropy_with_logits(yt, z, pos_weight=pw)
    l.backward()
    dyd, cl = torch.zeros(5, 12, device=cuda), torch.zeros(12, device=cuda)
    ops.bce_logits_pw(yl.to(cuda), z.to(cuda), pw.to(cuda), torch.full((12,), 1 / 60.0, device=cuda), dyd, cl)
    assert abs(cl.sum().item() / 60 - l.item()) < 1e-5 and (dyd.cpu() - yt.grad).abs().max() < 1e-6
    # DistillKL on a column slice, accumulating
    ys, tp = _rand((5, 20), 27, 3.0), _rand((5, 10), 28, 2.0)
    yt = ys.clone().requires_grad_()
    l = F.kl_div(F.log_softmax(yt[:, 4:14] / 4, 1), F.softmax(torch.sigmoid(tp) / 4, 1), reduction="sum") * 16 / 5
    (0.7 * l).backward()
    base = _rand((5, 20), 29)
    dyd, ls = base.clone().to(cuda), torch.zeros(1, device=cuda)
    ysd = ys.to(cuda)
    ops.distill_kl(ysd[:, 4:14], tp.to(cuda), dyd[:, 4:14], ls, 4.0, 0.7, accumulate=True)
    assert abs(ls.item() - l.item()) < 1e-5 and (dyd.cpu() - base - yt.grad).abs().max() < 1e-6
    # MSE
    a, bb = _rand((4, 33), 30), _rand((4, 33), 31)
    at = a.clone().requires_grad_()
    l = F.mse_loss(at, bb)
    (0.3 * l).backward()
    ls = torch.zeros(1, device=cuda)
    da = ops.mse(a.to(cuda), bb.to(cuda), ls, 0.3)
    assert abs(ls.item() - l.item()) < 1e-6 and (da.cpu() - at.grad).abs().max() < 1e-7


def test_kd_mix_backward(cuda):
    from computervision_codes_amd import ops
    b, c = 3, 64
    s = _rand((b, c), 41).abs()
    teas = [_rand((b, c), 42 + i, 0.2) for i in range(3)]
    gs = [_rand((b, c), 46 + i) for i in range(3)]
    st = s.clone().requires_grad_()
    tt = [t.clone().requires_grad_() for t in teas]
    tsum = torch.stack([t.sum(1) for t in tt], -1)
    attn = torch.softmax((st / c ** 0.5).unsqueeze(-1) * tsum.unsqueeze(1), -1)
    sum((st * attn[:, :, n] * gs[n]).sum() for n in range(3)).backward()
    ds, dtau = ops.kd_mix_bwd(s.to(cuda), [t.to(cuda) for t in teas], [g.to(cuda) for g in gs])
    assert (ds.cpu() - st.grad).abs().max() < 2e-5 * st.grad.abs().max()
    for n in range(3):          # d tea_n[b][d] is the same for every d
        assert (dtau[:, n].cpu()[:, None] - tt[n].grad).abs().max() < 2e-5 * max(tt[n].grad.abs().max().item(), 1e-3)


# ------------------------------------------------------------------------------------------------ whole step
def _trainer(cfg):
    from computervision_codes_amd.spatial_cnn_train import SpatialCnnTrainer
    from oracle.spatial_cnn_train import damp_residual_gamma, tie_free_bn
    table = shapes.spatial_cnn_shapes(cfg["network"])
    sd = damp_residual_gamma(synth.fill_from_shapes(table, seed=cfg["seed"]), cfg["network"], cfg.get("damp", 1.0))
    if cfg.get("tie_free"):
        sd = tie_free_bn(sd, cfg["network"])
    tr = SpatialCnnTrainer(cfg["network"], lr=cfg["lr"], weight_decay=1e-5, rates=cfg["rates"], temp=4.0)
    return tr.load_state_dict(sd), sd, table


def _gate_flips(tr, acts, tie=5e-4):
    """ReLU gates on which the HIP step and the oracle disagree.  Every disagreement must be a genuine near-tie (both values within
    `tie` x the layer's activation scale of zero); returns their count.  A flipped gate moves the gradient of its channel by ~1/M
    and everything upstream by 1e-3..1e-2 -- between ANY two fp32 implementations -- so callers relax tolerances when it is > 0."""
    n = 0
    for name, a in tr.relu_outputs().items():
        ref = acts[name].float()
        bad = (a > 0) != (ref > 0)
        if bad.any():
            scale = ref.abs().max().item()
            assert a[bad].abs().max().item() <= tie * scale and ref[bad].abs().max().item() <= tie * scale, (name, int(bad.sum()))
            n += int(bad.sum())
    return n


@pytest.mark.parametrize("name", ["cnn_train_resnet18", "cnn_train_resnet50", "cnn_train_resnet50_hard", "cnn_train_resnet50_tiefree"])
def test_train_step_vs_reference_autograd(cuda, name):
    """Fixtures captured from the reference model + torch autograd + torch.optim.SGD.  A random-weight ResNet in train mode is an
    ill-conditioned fp32 computation (fixture `grad_cond` = distance of the reference's OWN fp32 gradient from the fp64 gradient of
    the same step, up to 2e-1 for cnn_train_resnet50_hard), so the tolerance of each tensor is base + 8 x its grad_cond with
    base = 2e-4, or 5e-2 when a ReLU gate sits on a rounding-error tie and flipped (see _gate_flips)."""
    from oracle import spatial_cnn_train as o_ct
    z, cfg = load_golden(name)
    tr, sd, table = _trainer(cfg)
    img, labels, tpred, tfeat = _inputs(cfg)
    terms = tr.train_step(img.to(cuda), labels, tpred, tfeat, apply_update=False)
    for key in ("loss", "hard", "soft", "kd"):
        assert abs(terms[key] - float(z[key])) < 1e-4 * max(1.0, abs(float(z[key]))), (key, terms[key], float(z[key]))
    acts = {}
    o_ct.train_step(sd, img, labels, tpred, tfeat, cfg["network"], cfg["lr"], 1e-5, cfg["rates"], 4.0, acts=acts)
    flips = _gate_flips(tr, acts)
    if cfg.get("tie_free"):
        # the fixture whose ReLU inputs all lie > 0.2 from zero (margin checked on the reference's own ReLU inputs at generation,
        # oracle/gen_golden.py): no gate can flip, so this ResNet-50 step is held at the tight tolerance with no relaxation
        assert flips == 0
    base = 5e-2 if flips else 2e-4
    grads = tr.grads()
    names = [k for k, _ in table]
    cond = dict(zip(names, z["grad_cond"]))
    tol = lambda k: base + 8.0 * min(max(cond.get(k, 0.0), 0.0), 1.0)
    floor = max(1e-5, 1e-6 * float(z["grad_norms"].max()))    # (a gradient that is mathematically zero is rounding noise in any run)
    for k, ref in zip(names, z["grad_norms"]):
        if ref < 0:
            assert k not in grads
            continue
        gn = float(grads[k].norm())
        assert abs(gn - ref) <= tol(k) * max(ref, floor), (k, gn, ref, cond[k])
    tr.apply_update()
    new = tr.state_dict()
    for key in z.files:
        if not key.startswith("delta::"):
            continue
        k = key[len("delta::"):]
        flat = (new[k].float() - sd[k].float()).flatten()
        got = flat[:: max(1, flat.numel() // 2048)]
        ref = torch.from_numpy(z[key])
        ulp = 2.0 ** -22 * sd[k].float().abs().max().item()          # new - old is quantised by the parameter's own ulp
        assert (got - ref).abs().max().item() <= tol(k) * ref.abs().max().item() + ulp, (k, (got - ref).abs().max().item(), ref.abs().max().item())
    for k in names:
        if "num_batches_tracked" in k:
            assert int(new[k]) == int(sd[k]) + 1
        if k.startswith("basemodel.basemodel.fc."):
            assert torch.equal(new[k], sd[k])          # the trunk's own 1000-way fc never gets a gradient


def test_train_step_vs_oracle_every_tensor(cuda):
    """every gradient and every updated tensor (not only samples) against the CPU oracle, odd batch, rectangular frame; ResNet-18 at
    this size is well conditioned (torch fp32 vs fp64: 2e-5), so the tolerance is tight"""
    from oracle import spatial_cnn_train as o_ct
    cfg = dict(network="resnet18", B=3, H=64, W=96, seed=91, lr=0.05, rates=(1.0, 0.5, 2.0))
    tr, sd, table = _trainer(cfg)
    img, labels, tpred, tfeat = _inputs(cfg)
    acts = {}
    new_o, terms_o, g_o = o_ct.train_step(sd, img, labels, tpred, tfeat, cfg["network"], cfg["lr"], 1e-5, cfg["rates"], 4.0, acts=acts)
    terms = tr.train_step(img.to(cuda), labels, tpred, tfeat)
    for key in ("loss", "hard", "soft", "kd"):
        assert abs(terms[key] - terms_o[key]) < 1e-4 * max(1.0, abs(terms_o[key])), key
    flips = _gate_flips(tr, acts)
    gtol, ptol = (5e-2, 5e-3) if flips else (2e-4, 2e-5)
    grads = tr.grads()
    for k, g in grads.items():
        ref = g_o[k]
        assert (g - ref).abs().max().item() <= gtol * max(ref.abs().max().item(), 1e-6), (k, (g - ref).abs().max().item(), ref.abs().max().item())
    new = tr.state_dict()
    for k, _ in table:
        assert (new[k].float() - new_o[k].float()).abs().max().item() <= ptol * max(1.0, new_o[k].abs().max().item()), k


@pytest.mark.parametrize("cfg", [dict(network="resnet50", B=8, H=64, W=96, seed=602, lr=0.05, rates=(1.0, 1.0, 1.0)),
                                 dict(network="resnet50", B=5, H=96, W=64, seed=611, lr=0.05, rates=(1.0, 1.0, 1.0), damp=0.1),
                                 dict(network="resnet18", B=8, H=128, W=128, seed=612, lr=0.05, rates=(1.0, 1.0, 1.0))],
                         ids=["resnet50_plain_fill", "resnet50_damped", "resnet18_128"])
def test_gradients_as_close_to_fp64_as_torch_fp32(cuda, cfg):
    """The measure that does not depend on conditioning: the float64 oracle is the truth, torch's float32 autograd (what the
    reference runs) has some error against it, and the HIP step must not be further away -- in the median over all tensors within 2x
    of torch's error and per tensor within 6x + 2e-5, whenever all three runs take the same ReLU gates; when a gate sits on a
    rounding-error tie and flipped in one of the fp32 runs (see _gate_flips) only the coarse bounds apply."""
    from oracle import spatial_cnn_train as o_ct
    tr, sd, table = _trainer(cfg)
    img, labels, tpred, tfeat = _inputs(cfg)
    kw = dict(network=cfg["network"], lr=cfg["lr"], weight_decay=1e-5, rates=cfg["rates"], temp=4.0)
    a32, a64 = {}, {}
    _, t32, g32 = o_ct.train_step(sd, img, labels, tpred, tfeat, acts=a32, **kw)
    _, t64, g64 = o_ct.train_step_f64(sd, img, labels, tpred, tfeat, acts=a64, **kw)
    terms = tr.train_step(img.to(cuda), labels, tpred, tfeat, apply_update=False)
    flips_hip = _gate_flips(tr, a64)
    flips_t32 = sum(int(((a32[k] > 0) != (a64[k] > 0)).sum()) for k in a32)
    assert abs(terms["loss"] - t64["loss"]) <= 4 * abs(t32["loss"] - t64["loss"]) + 2e-5 * abs(t64["loss"])
    grads = tr.grads()
    e_hip, e_t32 = [], []
    for k, g in grads.items():
        den = max(g64[k].abs().max().item(), 1e-30)
        eh, et = (g.double() - g64[k]).abs().max().item() / den, (g32[k].double() - g64[k]).abs().max().item() / den
        e_hip.append(eh)
        e_t32.append(et)
    if flips_hip == 0 and flips_t32 == 0:      # same ReLU gates everywhere: the comparison is pure rounding
        outliers = [(k, eh, et) for k, eh, et in zip(grads, e_hip, e_t32) if eh > 6 * et + 2e-5]
        assert not outliers, outliers
        assert np.median(e_hip) <= 2 * np.median(e_t32) + 1e-6, (np.median(e_hip), np.median(e_t32))
    else:                                      # some gate on a rounding-error tie flipped in one of the fp32 runs
        assert max(e_hip) <= 0.3 and np.median(e_hip) <= max(2 * np.median(e_t32), 2e-2), (flips_hip, flips_t32, max(e_hip), np.median(e_hip))


@pytest.mark.parametrize("cfg", [dict(network="resnet18", B=8, H=256, W=448, seed=621, lr=0.05, rates=(1.0, 1.0, 1.0)),
                                 dict(network="resnet50", B=8, H=256, W=448, seed=622, lr=0.05, rates=(1.0, 1.0, 1.0), tie_free=True)],
                         ids=["resnet18_b8_256x448", "resnet50_b8_256x448"])
def test_full_size_student_step_as_close_to_fp64_as_torch_fp32(cuda, cfg):
    """The student's step at the size the shipped recipe runs and `bench.py spatial_train` quotes (`Scripts/train_fold1.sh:24`: --batch 8,
    Resize((256, 448)), `Spatial_cnn/dataloader.py:155`; ResNet-18 = the shipped student, ResNet-50 = BASELINE configs[3]), against the
    conditioning-free measure of test_gradients_as_close_to_fp64_as_torch_fp32: float64 oracle = truth, torch fp32 autograd = what the
    reference runs; the HIP fp32 step must be as close (median within 2x, every tensor within 6x + 2e-5 when all runs take the same ReLU gates;
    the ResNet-50 case uses the tie-free BatchNorm fill so that NO gate can flip and the tight bound always applies).  Then the bf16-operand
    mode at the same size (plain fill), at its declared tolerance (loss 3e-3; gradient norms 2 % median / 8 % p90 / 20 % worst)."""
    from computervision_codes_amd.spatial_cnn_train import SpatialCnnTrainer
    from oracle import spatial_cnn_train as o_ct
    tr, sd, table = _trainer(cfg)
    img, labels, tpred, tfeat = _inputs(cfg)
    kw = dict(network=cfg["network"], lr=cfg["lr"], weight_decay=1e-5, rates=cfg["rates"], temp=4.0)
    a32, a64 = {}, {}
    _, t32, g32 = o_ct.train_step(sd, img, labels, tpred, tfeat, acts=a32, **kw)
    _, t64, g64 = o_ct.train_step_f64(sd, img, labels, tpred, tfeat, acts=a64, **kw)
    terms = tr.train_step(img.to(cuda), labels, tpred, tfeat, apply_update=False)
    flips_hip = _gate_flips(tr, a64)
    flips_t32 = sum(int(((a32[k] > 0) != (a64[k] > 0)).sum()) for k in a32)
    del a32, a64
    if cfg.get("tie_free"):
        assert flips_hip == 0 and flips_t32 == 0
    assert abs(terms["loss"] - t64["loss"]) <= 4 * abs(t32["loss"] - t64["loss"]) + 2e-5 * abs(t64["loss"])
    for key in ("hard", "soft", "kd"):
        assert abs(terms[key] - t64[key]) <= 1e-4 * max(1.0, abs(t64[key])), key
    grads = tr.grads()
    e_hip, e_t32 = [], []
    for k, g in grads.items():
        den = max(g64[k].abs().max().item(), 1e-30)
        e_hip.append((g.double() - g64[k]).abs().max().item() / den)
        e_t32.append((g32[k].double() - g64[k]).abs().max().item() / den)
    if flips_hip == 0 and flips_t32 == 0:
        outliers = [(k, eh, et) for k, eh, et in zip(grads, e_hip, e_t32) if eh > 6 * et + 2e-5]
        assert not outliers, outliers
        assert np.median(e_hip) <= 2 * np.median(e_t32) + 1e-6, (np.median(e_hip), np.median(e_t32))
    else:
        assert max(e_hip) <= 0.3 and np.median(e_hip) <= max(2 * np.median(e_t32), 2e-2), (flips_hip, flips_t32, max(e_hip), np.median(e_hip))
    del tr, grads
    torch.cuda.empty_cache()
    # the bf16-operand mode at this size.  Its declared tolerance is stated for the plain synthetic fill (the reference fixtures' fill): under the
    # tie-free fill every BatchNorm weight is x 0.05, the gradients that survive the cancellations behind a train-mode BatchNorm (dL/dW is
    # orthogonal to W) shrink below the bf16 operand noise and their NORMS are off by 7 % in the median (measured) although fp32 holds 1e-5 --
    # a statement about that fill, not about the step.  So: plain fill, against the fp32 oracle of the same step.
    sd_b, ref_t, ref_g = sd, t64, g64
    if cfg.get("tie_free"):
        sd_b = synth.fill_from_shapes(table, seed=cfg["seed"])
        _, ref_t, ref_g = o_ct.train_step(sd_b, img, labels, tpred, tfeat, **kw)
    trb = SpatialCnnTrainer(cfg["network"], lr=cfg["lr"], weight_decay=1e-5, rates=cfg["rates"], temp=4.0, operand_dtype=torch.bfloat16).load_state_dict(sd_b)
    tb = trb.train_step(img.to(cuda), labels, tpred, tfeat, apply_update=False)
    for key in ("loss", "hard", "soft", "kd"):
        assert abs(tb[key] - ref_t[key]) <= 3e-3 * max(1.0, abs(ref_t[key])), (key, tb[key], ref_t[key])
    gb = trb.grads()
    norms = {k: float(ref_g[k].norm()) for k in gb}
    nmax = max(norms.values())
    rel = np.array([abs(float(gb[k].norm()) - norms[k]) / max(norms[k], 1e-6 * nmax) for k in gb if norms[k] > 0])
    assert np.median(rel) < 2e-2 and np.percentile(rel, 90) < 8e-2 and rel.max() < 0.2, (np.median(rel), np.percentile(rel, 90), rel.max())


def test_graph_replay_equals_eager_step(cuda):
    """hipGraph replay of the step: same losses, gradients, running statistics and update as the eager launches (atomics order aside)"""
    cfg = dict(network="resnet18", B=4, H=64, W=64, seed=77, lr=0.05, rates=(1.0, 1.0, 1.0))
    img, labels, tpred, tfeat = _inputs(cfg)
    fr = synth.synthetic_frames(cfg["B"], cfg["H"], cfg["W"], seed=cfg["seed"]).to(cuda)          # uint8 entry
    outs = []
    for use_graph in (False, True):
        tr, sd, table = _trainer(cfg)
        for _ in range(2):                                   # two steps: the replay must also see the updated weights
            terms = tr.train_step(fr, labels, tpred, tfeat, use_graph=use_graph)
        outs.append((terms, tr.state_dict()))
    (t0, s0), (t1, s1) = outs
    assert abs(t0["loss"] - t1["loss"]) < 1e-4 * abs(t0["loss"])
    for k in s0:
        if "num_batches_tracked" in k:
            assert int(s0[k]) == int(s1[k]) == int(sd[k]) + 2
        else:
            assert (s0[k].float() - s1[k].float()).abs().max().item() <= 1e-4 * max(1.0, s0[k].float().abs().max().item()), k


@pytest.mark.parametrize("task", ["i", "t"])
def test_single_task_train_step_vs_oracle(cuda, task):
    """`--loss_type i|v|t` (`Spatial_cnn/run.py:165-179`): only that classifier exists (`network.py:34-41`), the loss is its BCE(pos_weight)
    alone -- every gradient and every updated tensor against the CPU oracle (ResNet-18: well conditioned, tight tolerance)"""
    from computervision_codes_amd.spatial_cnn_train import SpatialCnnTrainer
    from oracle import spatial_cnn_train as o_ct
    cfg = dict(network="resnet18", B=3, H=64, W=96, seed=95, lr=0.05, rates=(1.0, 1.0, 1.0))
    table = shapes.spatial_cnn_shapes("resnet18", None, 1536, task)
    assert not any(k.startswith(("wi.", "mi.", "classifier_ivt")) for k, _ in table)
    sd = o_ct.tie_free_bn(synth.fill_from_shapes(table, seed=cfg["seed"]), "resnet18")     # no ReLU input near zero: no gate can flip
    tr = SpatialCnnTrainer("resnet18", lr=cfg["lr"], weight_decay=1e-5, loss_type=task).load_state_dict(sd)
    img, labels, _, _ = _inputs(cfg)
    acts = {}
    new_o, terms_o, g_o = o_ct.train_step(sd, img, labels, [], [], "resnet18", cfg["lr"], 1e-5, acts=acts, loss_type=task)
    terms = tr.train_step(img.to(cuda), labels, [], [])
    assert abs(terms["loss"] - terms_o["loss"]) < 1e-4 * max(1.0, abs(terms_o["loss"]))
    assert _gate_flips(tr, acts) == 0
    gtol, ptol = 1e-3, 2e-5
    gmax = max(float(v.abs().max()) for v in g_o.values() if v is not None)
    for k, g in tr.grads().items():
        ref = g_o[k]
        assert (g - ref).abs().max().item() <= gtol * max(ref.abs().max().item(), 1e-4 * gmax), (k, (g - ref).abs().max().item())
    new = tr.state_dict()
    assert list(new) == [k for k, _ in table]
    for k, _ in table:
        assert (new[k].float() - new_o[k].float()).abs().max().item() <= ptol * max(1.0, new_o[k].abs().max().item()), k


@pytest.mark.parametrize("name", ["cnn_train_resnet18", "cnn_train_resnet50_tiefree"])
def test_module_call_in_train_mode_vs_reference(cuda, name):
    """the module seam under `model.train()` (`Spatial_cnn/run.py:152-157`, `network.py:43-92`): `VideoNas(args.train = True).train()(img, feat_i,
    feat_v, feat_t)` returns the reference's tuple -- BatchNorm on batch statistics, KD branch outputs -- within 1e-3 of the values captured from
    the reference module; running statistics advance like torch's; eval() afterwards uses them; args.train keeps the KD branch on in eval mode"""
    import types
    from computervision_codes_amd.spatial_cnn import VideoNas
    from oracle.spatial_cnn_train import damp_residual_gamma, tie_free_bn
    z, cfg = load_golden(name)
    table = shapes.spatial_cnn_shapes(cfg["network"])
    sd = damp_residual_gamma(synth.fill_from_shapes(table, seed=cfg["seed"]), cfg["network"], cfg.get("damp", 1.0))
    if cfg.get("tie_free"):
        sd = tie_free_bn(sd, cfg["network"])
    img, _, _, tfeat = _inputs(cfg)
    args = types.SimpleNamespace(network=cfg["network"], loss_type="all", student_dim=shapes.resnet_feat_dim(cfg["network"]), teacher_dim=1536, train=True)
    m = VideoNas(args=args, dtype=torch.float32).load_state_dict(sd)
    with pytest.raises(TypeError):
        m.train()(img.to(cuda))                           # the KD branch needs the teacher features (`network.py:47`)
    out = m.train()(img.to(cuda), *[t.to(cuda) for t in tfeat])
    got = {"fwd_kd_i": out[0][0], "fwd_logit_i": out[0][1], "fwd_kd_v": out[1][0], "fwd_logit_v": out[1][1], "fwd_kd_t": out[2][0],
           "fwd_logit_t": out[2][1], "fwd_feat": out[3][0], "fwd_logit_ivt": out[3][1]}
    for k, v in got.items():
        ref = torch.from_numpy(z[k])
        assert tuple(v.shape) == tuple(ref.shape), k
        assert (v.float().cpu() - ref).abs().max().item() <= 1e-3 * max(1.0, ref.abs().max().item()), (k, (v.float().cpu() - ref).abs().max().item())
    after = m.state_dict()
    for key in z.files:
        if key.startswith("after::"):
            k = key[len("after::"):]
            assert (after[k].float() - torch.from_numpy(z[key])).abs().max().item() <= 1e-4 * max(1.0, float(np.abs(z[key]).max())), k
    assert int(after["basemodel.basemodel.bn1.num_batches_tracked"]) == int(sd["basemodel.basemodel.bn1.num_batches_tracked"]) + 1
    # eval() now folds the advanced running statistics; args.train keeps the KD branch on (validation loop, `run.py:231-256`)
    ev = m.eval()(img.to(cuda), *[t.to(cuda) for t in tfeat])
    assert torch.is_tensor(ev[0][0]) and tuple(ev[0][0].shape) == tuple(z["fwd_kd_i"].shape)
    m2 = VideoNas(args=args, dtype=torch.float32).load_state_dict(after)
    ev2 = m2.eval()(img.to(cuda), *[t.to(cuda) for t in tfeat])
    assert torch.equal(ev[3][0], ev2[3][0]) and torch.equal(ev[0][0], ev2[0][0])
