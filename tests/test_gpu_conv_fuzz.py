"""GPU: randomised shapes through `mt4_conv_nhwc` (fixed seed): every tile instantiation, FAST and generic staging, strides, dilations,
paddings, ragged M / N, residual, activations, fp32 (exact MFMA chain, tight tolerance) and bf16, against torch's CPU conv2d."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _case(rng):
    k = int(rng.choice([1, 1, 3, 3, 5]))
    kw = int(rng.choice([k, 1, 3])) if k > 1 else 1
    cin = int(rng.choice([8, 16, 24, 64, 72, 128, 256]))
    cout = int(rng.choice([8, 32, 48, 64, 96, 131, 256, 320]))
    b = int(rng.integers(1, 4))
    h, w = int(rng.integers(max(k, 3), 20)), int(rng.integers(max(kw, 3), 24))
    sh, sw = int(rng.choice([1, 1, 2])), int(rng.choice([1, 1, 2]))
    dh, dw = int(rng.choice([1, 1, 2])), int(rng.choice([1, 1, 3]))
    ph, pw = int(rng.integers(0, k // 2 * dh + 2)), int(rng.integers(0, kw // 2 * dw + 2))
    if (h + 2 * ph - dh * (k - 1) - 1) < 0 or (w + 2 * pw - dw * (kw - 1) - 1) < 0:
        ph, pw = dh * (k - 1), dw * (kw - 1)
    return dict(b=b, h=h, w=w, cin=cin, cout=cout, kh=k, kw=kw, s=(sh, sw), p=(ph, pw), d=(dh, dw),
                tile=int(rng.integers(0, 21)), res=bool(rng.integers(0, 2)), act=str(rng.choice(["none", "relu", "gelu"])))


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_conv_random_shapes(cuda, dtype):
    from computervision_codes_amd import ops
    rng = np.random.default_rng(20260101 if dtype == torch.float32 else 20260102)
    worst = 0.0
    for it in range(70):
        c = _case(rng)
        x = torch.from_numpy(rng.standard_normal((c["b"], c["cin"], c["h"], c["w"])).astype(np.float32))
        wt = torch.from_numpy((rng.standard_normal((c["cout"], c["cin"], c["kh"], c["kw"])) / np.sqrt(c["cin"] * c["kh"] * c["kw"])).astype(np.float32))
        bias = torch.from_numpy(rng.standard_normal(c["cout"]).astype(np.float32))
        if dtype == torch.bfloat16:
            x, wt = x.to(dtype).float(), wt.to(dtype).float()
        ref = F.conv2d(x, wt, bias, c["s"], c["p"], c["d"])
        res = torch.from_numpy(rng.standard_normal(tuple(ref.shape)).astype(np.float32)) if c["res"] else None
        if res is not None:
            if dtype == torch.bfloat16:
                res = res.to(dtype).float()
            ref = ref + res
        ref = {"none": lambda t: t, "relu": F.relu, "gelu": F.gelu}[c["act"]](ref)
        xd = x.permute(0, 2, 3, 1).contiguous().to(dtype).to(cuda)
        wp = ops.pack_conv_weight(wt.to(cuda), None, dtype)
        rd = res.permute(0, 2, 3, 1).contiguous().to(dtype).to(cuda) if res is not None else None
        y = ops.conv_nhwc(xd, wp, bias.to(cuda), kh=c["kh"], kw=c["kw"], stride=c["s"], pad=c["p"], dil=c["d"], residual=rd, act=c["act"],
                          tile=c["tile"])
        got = y.float().cpu().permute(0, 3, 1, 2)
        scale = max(1.0, ref.abs().max().item())
        err = (got - ref).abs().max().item() / scale
        worst = max(worst, err)
        assert err < (2e-5 if dtype == torch.float32 else 1.2e-2), (it, c, err)
