#!/usr/bin/env python3
"""Headline benchmark: spatial-extract frames/s (Spatial_cnn ResNet-50, 224x224, bf16, synthetic frames
resident in HBM) + per-video TCN latency (Temporal_tenco), beside the CPU oracle timed on the host.

  python bench.py --gpus N --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A "step" = one pass of the extraction hot path over one batch of B frames per GPU: uint8 frames ->
normalise -> ResNet-50 trunk -> avgpool feature [B,2048] + 4 logit heads (`Spatial_cnn/test.py:143-177`).
Frames are independent units, so ranks run disjoint batches with no data-path collective ("weak").
Prints ONE JSON line on rank 0.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import threading
import time
import types

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

PEAK_BF16_TFLOPS = 2500.0   # MI355X dense bf16 MFMA (MI355X_MICROARCH.md, chip-level parameters)
PEAK_F32_MFMA_TFLOPS = 157.3
PEAK_HBM_GBS = 8000.0


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=2672, help="frames per stream per step (multiples of 1336: layer3/4 tile counts land on whole rounds of 256 CUs)")
    ap.add_argument("--network", default="resnet50")
    ap.add_argument("--height", type=int, default=224)
    ap.add_argument("--width", type=int, default=224)
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "f32"])
    ap.add_argument("--streams", type=int, default=1, help="HIP streams per GPU; every stream gets --batch frames of a step (round 3: one stream over 2672 "
                    "frames = two streams of 1336 within 1 %, profiles/r03_stream_skew_ab.txt -- the launches of two parts do not overlap usefully)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-temporal", action="store_true")
    ap.add_argument("--no-ddp-train", action="store_true", help="N > 1: skip the data-parallel training sub-record")
    ap.add_argument("--ddp-timeout", type=float, default=300.0, help="N > 1: seconds after which the data-parallel training sub-record is given up and the headline line is printed without it")
    ap.add_argument("--per-layer", default="", help="write a per-conv-layer timing table (json) to this path")
    return ap.parse_args()


def conv_flops_per_frame(model, h, w):
    """algorithmic conv FLOPs (2*MAC, true K = Cin*kh*kw) of the trunk for one frame"""
    total = 0
    for rec in model.conv_plan(h, w):
        total += 2 * rec["Ho"] * rec["Wo"] * rec["Cout"] * rec["Cin"] * rec["kh"] * rec["kw"]
    return total


def time_conv_kernels(model, frames, iters=3):
    """Average duration of the conv launches of one step (implicit-GEMM kernels, the fused layer1 Bottlenecks, the stem + max-pool launch), measured with HIP events
    recorded on the stream the kernels are launched on (torch's current stream), one event pair per launch."""
    from computervision_codes_amd import ops
    pairs = []
    names = ("conv_nhwc", "conv3x3_expand", "bottleneck_fused", "bottleneck_fused_next", "stem_maxpool", "chain_gemm")
    origs = {n: getattr(ops, n) for n in names}

    def wrap(fn):
        def timed(*a, **k):
            e0 = torch.cuda.Event(enable_timing=True)
            e1 = torch.cuda.Event(enable_timing=True)
            e0.record()
            y = fn(*a, **k)
            e1.record()
            pairs.append((e0, e1))
            return y
        return timed

    for n in names:
        setattr(ops, n, wrap(origs[n]))
    try:
        per_iter = []
        for _ in range(iters):
            pairs.clear()
            model.extract_u8(frames)
            torch.cuda.synchronize()
            per_iter.append([a.elapsed_time(b) for a, b in pairs])
    finally:
        for n in names:
            setattr(ops, n, origs[n])
    n = len(per_iter[0])
    per_launch_ms = [min(it[i] for it in per_iter) for i in range(n)]
    return per_launch_ms


def host_cores():
    """cores this process may actually use: cgroup quota, else affinity, else cpu_count"""
    try:
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            return max(1, int(float(q) / float(p) + 0.5))
    except Exception:
        pass
    try:
        return len(os.sched_getaffinity(0))
    except Exception:
        return os.cpu_count() or 1


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.lower().startswith("model name"):
                return line.split(":", 1)[1].strip()
    except Exception:
        pass
    import platform
    return platform.processor() or "unknown"


def cpu_baseline_spatial(network, h, w, seed):
    """CPU oracle (port of the reference arithmetic) on the host cores, bounded sample."""
    from computervision_codes_amd import shapes, synth
    from oracle import spatial_cnn as o_cnn
    torch.set_num_threads(min(host_cores(), 64))
    sd = synth.fill_from_shapes(shapes.spatial_cnn_shapes(network), seed=seed)
    bs = 8
    img = synth.normalize_frames(synth.synthetic_frames(bs, h, w, seed=seed))
    reps = []
    with torch.no_grad():
        o_cnn.spatial_cnn_forward(sd, img, network)  # warm-up
        o_cnn.spatial_cnn_forward(sd, img, network)
        t_all = time.perf_counter()
        # >= 5 independent repeats of ~2.5 s each (the host cores are shared with other tenants of the node: one long loop drifted 133-211
        # frames/s between driver runs); value = the MEDIAN repeat, the spread is published beside it
        while len(reps) < 7 and (len(reps) < 5 or time.perf_counter() - t_all < 20.0):
            t0 = time.perf_counter()
            n = 0
            while True:
                o_cnn.spatial_cnn_forward(sd, img, network)
                n += 1
                dt = time.perf_counter() - t0
                if dt > 2.5 or n >= 30:
                    break
            reps.append((bs * n / dt, n))
    rates = sorted(r for r, _ in reps)
    med = rates[len(rates) // 2]
    return dict(value=round(med, 2), unit="frames/s", cores=torch.get_num_threads(), kind="port", cpu=cpu_model(), os_cpu_count=os.cpu_count(),
                repeats=len(rates), min=round(rates[0], 2), max=round(rates[-1], 2), spread=round((rates[-1] - rates[0]) / med, 3),
                sample=f"median of {len(rates)} repeats of {reps[0][1]}-{max(n for _, n in reps)} batches of {bs} frames {h}x{w} each "
                       f"(~{time.perf_counter() - t_all:.0f} s of CPU work), torch-CPU fp32 oracle, eval, no_grad")


def _time_call(fn, iters=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(iters):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        fn()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    ts.sort()
    return ts[len(ts) // 2]


def temporal_bench(dev, do_cpu):
    """per-video latency of the Temporal_tenco head (4-stage D=512 at T=256 / T=2000, BASELINE config 1) and of the
    MS-TCT teacher on one 256-frame window; eager launches and hipGraph replay."""
    from computervision_codes_amd import shapes, synth
    from computervision_codes_amd.graph import GraphedForward
    from computervision_codes_amd.temporal_mstct import VideoNas as MstctNas
    from computervision_codes_amd.temporal_tenco import VideoNas
    out = {}
    cases = [("tenco4_T256", dict(num_R=3, dim=512, fpn=True, T=256)), ("tenco4_T2000", dict(num_R=3, dim=512, fpn=True, T=2000)),
             ("config1_T256", dict(num_R=0, dim=2048, fpn=False, T=256))]
    for name, c in cases:
        args = types.SimpleNamespace(fpn=c["fpn"], output=False, hier=False, mask=True)
        table = shapes.tenco_shapes(11, 10, c["num_R"], 512, c["dim"], 100, fpn=c["fpn"])
        sd = synth.fill_from_shapes(table, seed=47)
        m = VideoNas(args, 11, 10, c["num_R"], 512, c["dim"], 100).eval().load_state_dict(sd)
        x = synth.synthetic_features(c["T"], c["dim"], seed=47).to(dev)
        eager = _time_call(lambda: m(x, False))
        g = GraphedForward(lambda xx: m(xx, False), [x])
        lat = _time_call(lambda: g(x))
        nparam = sum(v.numel() for v in sd.values())
        L = 11 + 10 * c["num_R"]
        nlev = 4 if c["fpn"] else 1
        alg_bytes = 4 * (nparam + c["T"] * (c["dim"] + 2 * 512 * L + nlev * (131 if c["fpn"] else 100)))
        flops = 2 * c["T"] * (c["dim"] * 512 + L * (512 * 1536 + 512 * 512) + (3 * 512 * 512 + nlev * 131 * 512 if c["fpn"] else 100 * 512))
        rec = dict(ms_per_video=round(lat, 4), ms_per_video_eager=round(eager, 4), T=c["T"], algorithmic_MB=round(alg_bytes / 1e6, 1),
                   hbm_frac=round(alg_bytes / (lat * 1e-3) / 1e9 / PEAK_HBM_GBS, 4),
                   f32_mfma_frac=round(flops / (lat * 1e-3) / 1e12 / PEAK_F32_MFMA_TFLOPS, 4))
        mb = VideoNas(args, 11, 10, c["num_R"], 512, c["dim"], 100, dtype=torch.bfloat16).eval().load_state_dict(sd)
        gb = GraphedForward(lambda xx: mb(xx, False), [x])
        rec["ms_per_video_bf16"] = round(_time_call(lambda: gb(x)), 4)   # throughput mode; the fp32 line above is the parity mode
        if do_cpu:
            from oracle import tenco as o_tenco
            xc = x.cpu()
            with torch.no_grad():
                o_tenco.tenco_forward(sd, xc, 11, 10, c["num_R"], c["fpn"])
                t0 = time.perf_counter()
                n = 0
                while n < 10 and time.perf_counter() - t0 < 4.0:
                    o_tenco.tenco_forward(sd, xc, 11, 10, c["num_R"], c["fpn"])
                    n += 1
                rec["cpu_ms_per_video"] = round((time.perf_counter() - t0) / n * 1e3, 2)
        out[name] = rec
    # throughput of the same head: B videos of 256 frames per forward (frames of different videos never meet: a video's rows are bit-identical
    # whatever rides along), hipGraph replay
    thr = {}
    args = types.SimpleNamespace(fpn=True, output=False, hier=False, mask=True)
    sd = synth.fill_from_shapes(shapes.tenco_shapes(11, 10, 3, 512, 512, 100, fpn=True), seed=47)
    for dt, tag in ((torch.float32, "f32"), (torch.bfloat16, "bf16")):
        m = VideoNas(args, 11, 10, 3, 512, 512, 100, dtype=dt).eval().load_state_dict(sd)
        for B in (8, 32, 64):          # (bf16, 24 ... 96 videos: every DilatedResidualLayer is ONE launch, `mt4_tcn_layer_fused_bf16`)
            xb = torch.stack([synth.synthetic_features(256, 512, seed=47 + i)[0] for i in range(8)]).repeat(B // 8, 1, 1).to(dev).contiguous()
            g = GraphedForward(lambda xx: m(xx, False), [xb])
            ms = _time_call(lambda: g(xb), iters=10)
            thr[f"B{B}_T256_{tag}"] = dict(videos_per_s=round(B / ms * 1e3, 1), ms_per_forward=round(ms, 4))
    out["tenco4_throughput"] = thr
    # MS-TCT teacher, one 256-frame window, D=2048 (Temporal_mstct/run.py:306-313)
    a = types.SimpleNamespace(loss_type="ivt")
    sd = synth.fill_from_shapes(shapes.mstct_shapes(2048, (256, 384, 576, 864), 2, 8, 512, "ivt"), seed=47)
    m = MstctNas(a, [256, 384, 576, 864], 2, 8, 8, 2048, 512).eval().load_state_dict(sd)
    x = synth.synthetic_features(256, 2048, seed=47).to(dev)
    eager = _time_call(lambda: m.forward_btd(x))
    g = GraphedForward(lambda xx: m.forward_btd(xx), [x])
    rec = dict(ms_per_window=round(_time_call(lambda: g(x)), 4), ms_per_window_eager=round(eager, 4), T=256, gflop=31.4)
    if do_cpu:
        from oracle import mstct as o_mstct
        xc = x.cpu().permute(0, 2, 1).contiguous()
        with torch.no_grad():
            o_mstct.mstct_forward(sd, xc, "ivt")
            t0 = time.perf_counter()
            n = 0
            while n < 5 and time.perf_counter() - t0 < 4.0:
                o_mstct.mstct_forward(sd, xc, "ivt")
                n += 1
            rec["cpu_ms_per_window"] = round((time.perf_counter() - t0) / n * 1e3, 2)
    out["mstct_T256"] = rec
    # one DDP-style training step of the 4-stage head (forward + BCE + backward + SGD; Temporal_tenco/run.py:181-235)
    from computervision_codes_amd.tenco_train import TencoTrainer
    import numpy as np
    tr = TencoTrainer(lr=0.01, device=str(dev)).load_state_dict(synth.fill_from_shapes(shapes.tenco_shapes(), seed=47))
    for T in (256, 2000):
        xt = synth.synthetic_features(T, 512, seed=47).to(dev)
        zl = tr.prepare_labels({s: torch.from_numpy((synth.uniform01(3, i, T * k) < 0.1).reshape(T, k).astype(np.int64))
                                for i, (s, k) in enumerate((("", 100), ("_i", 6), ("_v", 10), ("_t", 15)))})
        ms = _time_call(lambda: tr.train_step(xt, zl, use_graph=True), iters=10)
        out[f"tenco4_train_T{T}"] = dict(ms_per_step=round(ms, 3), T=T, note="fwd+BCE+bwd+SGD, hipGraph replay, 1 GPU")
    out["mstct_train_b31_T256"] = mstct_train_bench(dev)
    torch.cuda.empty_cache()
    out["q2l_train_swinL_384_b16"] = q2l_train_bench(dev)
    torch.cuda.empty_cache()
    return out


def mstct_train_bench(dev):
    """BASELINE configs[3]'s temporal half: one training step of the MS-TCT teacher on the reference's batch (31 windows of 256 frames,
    Swin-L features D = 1536, `Scripts/train_fold1.sh:16`; `Temporal_mstct/run.py:147-235`): forward + BCE + backward + SGD; fp32, and with
    the nn.Linear GEMMs on bf16 operand copies (fp32 activations, accumulation and master weights)"""
    from computervision_codes_amd import shapes, synth
    from computervision_codes_amd.mstct_train import MstctTrainer
    B, T, D = 31, 256, 1536
    out = {}
    for odt in (torch.float32, torch.bfloat16):
        tr = MstctTrainer((256, 384, 576, 864), 2, 8, 8, D, 512, "i", lr=0.01, device=str(dev), operand_dtype=odt).load_state_dict(
            synth.fill_from_shapes(shapes.mstct_shapes(D, (256, 384, 576, 864), 2, 8, 512, "i"), seed=47))
        x = torch.randn(B, T, D, device=dev)
        z = (torch.rand(B * T, 6, device=dev) < 0.15).float()
        masks = tr.draw_masks_device(B, T, 1, 0)
        ms = _time_call(lambda: tr.train_step_btd(x, z, masks=masks), iters=5)
        ms_g = _time_call(lambda: tr.train_step_btd(x, z, masks=masks, use_graph=True), iters=5)
        gflop = 3 * 31.2 * B     # ~3x the forward: 31.4 GFLOP per window at D = 2048, 31.2 at D = 1536 (D enters the first conv only)
        rec = dict(ms_per_step=round(ms, 2), ms_per_step_graph=round(ms_g, 2), windows_per_s=round(B / min(ms, ms_g) * 1e3, 1), batch=B, T=T, D=D,
                   dtype="f32" if odt == torch.float32 else "bf16 GEMM operands, fp32 activations / master weights",
                   approx_tflops=round(gflop / min(ms, ms_g), 1), note="fwd + BCE(pos_weight) + bwd + SGD, dropout masks drawn outside the timed region, 1 GPU")
        if odt == torch.float32:
            out.update(rec)
        else:
            out["bf16_operands"] = rec
        del tr
    return out


def q2l_train_bench(dev, backbone="swin_L_384_22k", img=384, hidden=1536, B=16):
    """BASELINE configs[3]'s spatial half: one training step of the Swin-L + Query2Label teacher on the reference's batch (16 frames at 384 x 384,
    `Scripts/train_fold1.sh:12`; `Spatial_transformer/run.py:150-229`): forward + BCE(pos_weight) + backward + SGD, DropPath 0.1 and the
    transformer's dropout 0.1 on; fp32, and with the nn.Linear GEMMs on bf16 operand copies"""
    from computervision_codes_amd import shapes, synth
    from computervision_codes_amd.q2l_train import Q2LTrainer
    gflop_fwd = {"swin_L_384_22k": 207.8 + 13.6, "swin_B_384_22k": 94.2 + 7.0, "swin_T_224_1k": 9.0 + 1.3}[backbone]   # 2 x MACs: backbone + 1 decoder
    out = {}
    for odt in (torch.float32, torch.bfloat16):
        torch.cuda.empty_cache()
        torch.cuda.reset_peak_memory_stats()
        tr = Q2LTrainer(backbone, img, hidden, "i", lr=0.01, device=str(dev), operand_dtype=odt).load_state_dict(
            synth.fill_from_shapes(shapes.q2l_param_shapes(backbone, img, hidden, "i"), seed=47))
        frames = device_frames(B, img, img, 7, dev)
        z = (torch.rand(B, 6, device=dev) < 0.3).float()
        masks = tr.draw_masks_device(B, 1, 0)
        ms = _time_call(lambda: tr.train_step(frames, z, masks=masks), iters=5)
        rec = dict(ms_per_step=round(ms, 2), frames_per_s=round(B / ms * 1e3, 1), batch=B, img=img, backbone=backbone,
                   dtype="f32" if odt == torch.float32 else "bf16 GEMM operands, fp32 activations / master weights",
                   approx_tflops=round(3 * gflop_fwd * B / ms, 1), peak_mem_gb=round(torch.cuda.max_memory_allocated() / 2 ** 30, 1),
                   note="fwd + BCE(pos_weight) + bwd + SGD, masks drawn outside the timed region, 1 GPU")
        if odt == torch.float32:
            out.update(rec)
        else:
            out["bf16_operands"] = rec
        del tr
    return out


def device_frames(n, h, w, seed, dev, nbase=16):
    """n uint8 frames on the device: `nbase` of them from the counter generator on the host (it makes ~2 M values/s), the others
    derived on the device as base[i % nbase] XOR a per-group byte -- still uniform random bytes, no two frames equal"""
    from computervision_codes_amd import synth
    nbase = min(n, nbase)
    base = synth.synthetic_frames(nbase, h, w, seed=seed).to(dev)
    idx = torch.arange(n, device=dev)
    return (base[idx % nbase] ^ ((idx // nbase * 37) % 256).to(torch.uint8)[:, None, None, None]).contiguous()


SWIN_BATCH = {384: 227, 224: 668}     # frames per forward: stage 2 then has 510.75 / 511.4 row tiles of 256 tokens and stage 3 + the decoders 127.7 / 127.9 --
                                      # whole rounds of the 256 CUs for the GEMMs with 2 or 4 column tiles (proj, fc2, linear2), which at 128 frames run
                                      # 2.25 / 1.125 rounds (profiles/r04_swin_batch_sweep.txt: 384^2 4398 -> 4702 frames/s, 224^2 10542 -> 12849)


def swin_bench(dev, batch=None):
    """BASELINE configs[2]: Swin-B + the CholecT50 triplet head = `loss_type all` (four Q2L decoders over the shared transformer +
    the KD mixing), bf16, frames/s at 384x384 (reference-legal swin_B_384_22k) and 224x224; the single-decoder teacher
    configuration (`loss_type i`, what Scripts/train_fold1.sh trains) beside it.  Frames per forward: SWIN_BATCH (and 128, the batch of the
    rounds before, for the four-decoder configuration)."""
    from computervision_codes_amd import shapes, synth
    from computervision_codes_amd.spatial_transformer import build_q2l
    out = {}
    # (the last row: the teacher Scripts/train_fold1.sh ships -- Swin-L/384, hidden 1536, one decoder; 2 x 103.9 GMAC backbone + the decoder at d = 1536)
    for name, img, hidden, lts in (("swin_B_384_22k", 384, 1024, ("all", "i")), ("swin_B_224_22k", 224, 1024, ("all", "i")), ("swin_L_384_22k", 384, 1536, ("i",))):
        for lt in lts:
            args = types.SimpleNamespace(backbone=name, img_size=img, hidden_dim=hidden, loss_type=lt)
            m = build_q2l(args, dtype=torch.bfloat16, device=str(dev)).eval()
            m.load_state_dict(synth.fill_from_shapes(shapes.q2l_param_shapes(name, img, hidden, lt), seed=7))
            ndec = 4 if lt == "all" else 1
            gf = (94.2 + 15.3 * ndec) if img == 384 else (30.9 + 10.2 * ndec)
            if name.startswith("swin_L"):
                gf = 207.8 + 15.3 * ndec * (1536 / 1024) ** 2
            for b in ([batch] if batch else ([SWIN_BATCH[img], 128] if lt == "all" else [SWIN_BATCH[img]])):
                frames = device_frames(b, img, img, 7, dev)
                tf = [synth.synthetic_features(b, 512, seed=7 + k)[0].to(dev) for k in (1, 2, 3)] if lt == "all" else []
                ms = _time_call(lambda: m(frames, *tf), iters=5)
                rec = dict(frames_per_s=round(b / ms * 1e3, 1), ms_per_batch=round(ms, 3), batch=b, decoders=ndec,
                           mfma_frac=round(gf * 1e9 * b / (ms * 1e-3) / 1e12 / PEAK_BF16_TFLOPS, 4))
                out[f"{name}_{lt}" + ("" if b == (batch or SWIN_BATCH[img]) else f"_b{b}")] = rec
                del frames, tf
            del m
    return out


def parity_mode_bench(dev, network, h, w, batch=256):
    """the same extractor in the fp32 parity mode (fp32 I/O, exact-fp32 MFMA chain: what the 1e-3 logit parity is claimed for)"""
    from computervision_codes_amd import shapes, synth
    from computervision_codes_amd.spatial_cnn import VideoNas
    args = types.SimpleNamespace(network=network, loss_type="all", student_dim=shapes.resnet_feat_dim(network), teacher_dim=1536, train=False)
    m = VideoNas(args=args, dtype=torch.float32, device=str(dev)).eval()
    m.load_state_dict(synth.fill_from_shapes(shapes.spatial_cnn_shapes(network), seed=1234))
    frames = device_frames(batch, h, w, 99, dev)
    ms = _time_call(lambda: m.extract_u8(frames), iters=5)
    per_launch = time_conv_kernels(m, frames)
    tf = conv_flops_per_frame(m, h, w) * batch / (sum(per_launch) * 1e-3) / 1e12
    return dict(frames_per_s=round(batch / ms * 1e3, 1), ms_per_batch=round(ms, 3), batch=batch, dtype="f32",
                conv_tflops=round(tf, 1), f32_mfma_frac=round(tf / PEAK_F32_MFMA_TFLOPS, 4))


def native_resolution_bench(dev, model, streams, batch=584):
    """the bench's extractor on the reference's own frame size, Resize((256, 448)) (`Spatial_cnn/dataloader.py:155`): 584 frames per stream
    are the pixels of 1336 frames of 224x224.  18.69 GFLOP per frame."""
    frames = device_frames(batch * max(1, streams), 256, 448, 77, dev)
    ms = _time_call(lambda: model.extract_u8(frames, streams=streams), iters=5)
    fl = conv_flops_per_frame(model, 256, 448)
    n = batch * max(1, streams)
    return dict(frames_per_s=round(n / ms * 1e3, 1), ms_per_step=round(ms, 3), frames_per_step=n, gflop_per_frame=round(fl / 1e9, 2),
                mfma_frac=round(fl * n / (ms * 1e-3) / 1e12 / PEAK_BF16_TFLOPS, 4))


def student_resnet18_bench(dev, streams):
    """the extractor with the reference's own student network (`--network resnet18 --student_dim 512`, Scripts/test_fold1.sh) at both
    frame sizes; 3.63 GFLOP per 224x224 frame"""
    from computervision_codes_amd import shapes, synth
    from computervision_codes_amd.spatial_cnn import VideoNas
    args = types.SimpleNamespace(network="resnet18", loss_type="all", student_dim=512, teacher_dim=1536, train=False)
    m = VideoNas(args=args, dtype=torch.bfloat16, device=str(dev)).eval()
    m.load_state_dict(synth.fill_from_shapes(shapes.spatial_cnn_shapes("resnet18"), seed=1234))
    out = {}
    for (h, w, batch) in ((224, 224, 1336), (256, 448, 584)):
        n = batch * max(1, streams)
        frames = device_frames(n, h, w, 55, dev)
        ms = _time_call(lambda: m.extract_u8(frames, streams=streams), iters=5)
        fl = conv_flops_per_frame(m, h, w)
        out[f"{h}x{w}"] = dict(frames_per_s=round(n / ms * 1e3, 1), ms_per_step=round(ms, 3), frames_per_step=n,
                                gflop_per_frame=round(fl / 1e9, 2), mfma_frac=round(fl * n / (ms * 1e-3) / 1e12 / PEAK_BF16_TFLOPS, 4))
    return out


def e2e_script_bench(dev):
    """Script-level extraction rate: exactly the per-video loop `drivers.spatial_cnn_test` runs (`extract.extract_video_device`: the
    video's frames in passes of --device_batch = 512 through the extractor, features + logits kept on the device, ONE pinned D2H per
    video) on a synthetic 2000-frame video of 256x448 uint8 frames that is already on the device -- PNG decode and the host-side resize
    input are excluded and stated; the D2H of features / logits and the per-video synchronisation are included.  Student network
    of the shipped scripts (ResNet-18, `Scripts/test_fold1.sh`) and the bench's ResNet-50."""
    from computervision_codes_amd import extract, shapes, synth
    from computervision_codes_amd.spatial_cnn import VideoNas
    out = {}
    n = 2000
    frames = device_frames(n, 256, 448, 321, dev)
    for net in ("resnet18", "resnet50"):
        args = types.SimpleNamespace(network=net, loss_type="all", student_dim=shapes.resnet_feat_dim(net), teacher_dim=1536, train=False)
        for dt in (torch.bfloat16, torch.float32):
            m = VideoNas(args=args, dtype=dt, device=str(dev)).eval().load_state_dict(synth.fill_from_shapes(shapes.spatial_cnn_shapes(net), seed=3))
            run = lambda: extract.extract_video_device(m, n, lambda s, e: frames[s:e], 512)
            run()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            reps = 3
            for _ in range(reps):
                feat, lgs = run()
            dt_s = (time.perf_counter() - t0) / reps
            assert feat.shape == (n, shapes.resnet_feat_dim(net)) and lgs[3].shape == (n, 100)
            out[f"{net}_{'bf16' if dt == torch.bfloat16 else 'f32'}_256x448"] = dict(
                frames_per_s=round(n / dt_s, 1), ms_per_video=round(dt_s * 1e3, 2), frames_per_video=n, device_batch=512,
                includes="preprocess + trunk + heads + D2H of features and logits (pinned, once per video) + sync",
                excludes="PNG decode and H2D of the frames (host side)")
            del m
    out["png_decode_480x854"] = png_decode_bench(dev)
    try:
        out["from_png_files_480x854"] = png_files_bench(dev)
    except Exception as e:                       # (no writable temp directory / no Pillow: the record says so, the bench line stands)
        out["from_png_files_480x854"] = {"error": repr(e)}
    return out


def png_files_bench(dev, n=2048):
    """The same per-video loop from PNG FILES on disk, decode included (what `Spatial_cnn/test.py` does per video): n synthetic frames of the
    dataset's native 480 x 854 written to a temp directory, `cholect.load_frames_device` (decode, Resize to 256 x 448) -> ResNet-50 bf16 ->
    one D2H; Pillow on 16 host threads against the device decoder as the extraction driver runs it (loads of 1024 frames, two ahead)."""
    import io
    import shutil
    import tempfile
    from PIL import Image
    from computervision_codes_amd import cholect, extract, shapes, synth
    from computervision_codes_amd.spatial_cnn import VideoNas
    root = tempfile.mkdtemp(prefix="mt4_png_")
    try:
        os.makedirs(os.path.join(root, "data", "VID01"))
        h, w = 480, 854
        rng = np.random.default_rng(0)
        y, x = np.mgrid[0:h, 0:w]
        blobs = []
        for i in range(16):
            base = (np.stack([(x + 7 * i) % 256, (y * 2 + 3 * i) % 256, ((x + y) // 2) % 256], -1)).astype(np.int32)
            wave = 40 * np.sin(x[..., None] / (17.0 + i) + np.arange(3)) * np.cos(y[..., None] / (23.0 + i))
            fr = np.clip(base * 0.5 + 60 + wave + rng.normal(0, 3.0, (h, w, 3)), 0, 255).astype(np.uint8)
            b = io.BytesIO()
            Image.fromarray(fr, "RGB").save(b, format="PNG")
            blobs.append(b.getvalue())
        for i in range(n):
            with open(os.path.join(root, "data", "VID01", f"{i:06d}.png"), "wb") as f:
                f.write(blobs[i % 16])
        args = types.SimpleNamespace(network="resnet50", loss_type="all", student_dim=2048, teacher_dim=1536, train=False)
        m = VideoNas(args=args, dtype=torch.bfloat16, device=str(dev)).eval().load_state_dict(synth.fill_from_shapes(shapes.spatial_cnn_shapes("resnet50"), seed=3))
        ids = np.arange(n)
        rec, ref = {"frames_per_video": n, "kb_per_png": len(blobs[0]) // 1000, "network": "resnet50 bf16, Resize to 256x448"}, None
        for key, decode, load_batch, depth in (("pillow_16_threads", "host", None, 1), ("device_decode", "device", 1024, 2)):
            load = lambda s, e: cholect.load_frames_device(root, "VID01", ids[s:e], 256, 448, device=dev, workers=16, decode=decode)
            run = lambda: extract.extract_video_device(m, n, load, 512, prefetch=depth, load_batch=load_batch)
            run()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            feat, _ = run()
            dt_s = time.perf_counter() - t0
            ref = feat.copy() if ref is None else ref
            rec[key] = dict(frames_per_s=round(n / dt_s, 1), ms_per_video=round(dt_s * 1e3, 1), same_features=bool(np.array_equal(ref, feat)))
        return rec
    finally:
        shutil.rmtree(root, ignore_errors=True)


def png_decode_bench(dev, n=2560):
    """the stage in front of the extractor on real data: PNG files -> uint8 frames on the device (`pngdec.decode_batch`: the host walks the chunk
    lists and uploads the compressed bytes, inflate + unfiltering run in HIP).  16 distinct synthetic frames of the dataset's native 480 x 854
    (smooth structure + sensor-like noise: ~650 KB per PNG), repeated to n per call (2560: what the extraction driver hands the decoder at once, one full round of ten frames per CU); kernel time from HIP events, wall time incl. the host part."""
    import io
    try:
        from PIL import Image
    except Exception:
        return None
    from computervision_codes_amd import pngdec
    h, w = 480, 854
    rng = np.random.default_rng(0)
    y, x = np.mgrid[0:h, 0:w]
    files = []
    for i in range(16):
        base = (np.stack([(x + 7 * i) % 256, (y * 2 + 3 * i) % 256, ((x + y) // 2) % 256], -1)).astype(np.int32)
        blobs = 40 * np.sin(x[..., None] / (17.0 + i) + np.arange(3)) * np.cos(y[..., None] / (23.0 + i))
        fr = np.clip(base * 0.5 + 60 + blobs + rng.normal(0, 3.0, (h, w, 3)), 0, 255).astype(np.uint8)
        b = io.BytesIO()
        Image.fromarray(fr, "RGB").save(b, format="PNG")
        files.append(b.getvalue())
    files = [files[i % 16] for i in range(n)]
    pngdec.decode_batch(files, dev)                 # (first call: page-locks the staging buffer)
    torch.cuda.synchronize()
    tm = {}
    t0 = time.perf_counter()
    out = pngdec.decode_batch(files, dev, timings=tm)
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    assert tuple(out.shape) == (n, h, w, 3)
    gpu_ms = tm["inflate_ms"] + tm["unfilter_ms"]
    return dict(frames_per_s_gpu_time=round(n / gpu_ms * 1e3, 1), frames_per_s_wall=round(n / wall, 1), frames_per_call=n, kb_per_png=round(sum(map(len, files)) / n / 1e3),
                inflate_ms=round(tm["inflate_ms"], 2), unfilter_ms=round(tm["unfilter_ms"], 2),
                note="wall = host chunk walk + gather + upload of the compressed bytes (one thread) + the two launches; the extraction records above start from decoded frames")


def spatial_train_bench(dev):
    """BASELINE configs[4] on one GPU: the student distillation step (`Spatial_cnn/run.py -t`, batch 8 of 256x448 frames as in
    Scripts/train_fold1.sh, and batch 64) -- forward, hard+soft+KD losses, backward, SGD; hipGraph replay of forward+backward; fp32 and the
    bf16-operand mode."""
    from computervision_codes_amd import shapes, synth
    from computervision_codes_amd.spatial_cnn_train import SpatialCnnTrainer
    out = {}
    # (batch 256: layer3 / layer4 of a 64-frame batch give their GEMMs 0.2 ... 0.9 rounds of 256 x 256 tiles; 288 GB of HBM hold far more --
    #  the step at 256 frames peaks at 31 GB -- and the rate rises 3.6 k -> 4.5 k frames/s)
    for net, B, odt in (("resnet18", 8, torch.float32), ("resnet50", 8, torch.float32), ("resnet50", 64, torch.float32), ("resnet50", 8, torch.bfloat16),
                        ("resnet50", 64, torch.bfloat16), ("resnet50", 256, torch.bfloat16), ("resnet18", 256, torch.bfloat16)):
        H, W = 256, 448
        tr = SpatialCnnTrainer(net, lr=0.01, device=str(dev), operand_dtype=odt).load_state_dict(synth.fill_from_shapes(shapes.spatial_cnn_shapes(net), seed=5))
        frames = synth.synthetic_frames(B, H, W, seed=1).to(dev) if B <= 64 else device_frames(B, H, W, 1, dev)
        z = torch.cat([torch.from_numpy((synth.uniform01(5, i, B * k) < 0.15).reshape(B, k).astype(np.float32)) for i, k in
                       enumerate((6, 10, 15, 100))], 1).to(dev)
        tp = [synth.synthetic_features(B, k, seed=11 + i)[0].to(dev) for i, k in enumerate((6, 10, 15))]
        tf = [synth.synthetic_features(B, 1536, seed=21 + i)[0].to(dev) for i in range(3)]
        ms = _time_call(lambda: tr.train_step(frames, z, tp, tf, use_graph=True), iters=10 if B <= 64 else 4)
        bf = odt == torch.bfloat16
        out[f"{net}_b{B}_{H}x{W}" + ("_bf16" if bf else "")] = dict(
            ms_per_step=round(ms, 3), frames_per_s=round(B / ms * 1e3, 1), dtype="bf16 GEMM operands, fp32 master weights / sums" if bf else "f32",
            note="fwd + BCE/DistillKL/MSE + bwd + SGD, train-mode BatchNorm, 1 GPU")
        del tr
    return out


def self_launch(a):
    """`python bench.py --gpus N` with N > 1 and no rendezvous in the environment: start the N ranks ourselves, as a CHILD
    `torch.distributed.run` (one process per GPU, RCCL), BEFORE anything here has touched the GPU, and exit with its code."""
    import socket
    import subprocess
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={a.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    raise SystemExit(subprocess.call(cmd, env=env))


def plumbing_only(a, world, rank):
    """MT4_BENCH_PLUMBING=1 (CPU tests of the N-rank protocol, no kernels): rendezvous over gloo, the barrier-bracketed timed region with
    a sleep for a step, MAX over ranks, one JSON line from rank 0 -- everything of the N > 1 path except the GPU work."""
    import torch.distributed as dist
    if world > 1:
        dist.init_process_group("gloo")
    for _ in range(a.warmup):
        time.sleep(0.001)
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        time.sleep(0.002 * (1 + rank))
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps({"metric": METRIC, "value": None, "unit": "frames/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
                          "ms_per_step": round(dt / a.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak", "plumbing_only": True}))


def ddp_train_bench(dev, dist, world, rank):
    """Data-parallel training rate of the two student trainers (BASELINE configs[3]/[4]; reference loops `Spatial_cnn/run.py:145-224`,
    `Temporal_tenco/run.py:181-235`): every rank steps on its own batch / video, gradients meet in the flat buffer's all-reduce (RCCL).
    Per case: whole-job rate, ms per step (MAX over ranks), the all-reduce of the full gradient payload timed alone, the step with the
    exchange switched off, and from those the share of the exchange that hides behind the backward."""
    from computervision_codes_amd import shapes, synth
    from computervision_codes_amd.spatial_cnn_train import SpatialCnnTrainer
    from computervision_codes_amd.tenco_train import TencoTrainer

    def sync_max(v):
        if dist is None:
            return v
        t = torch.tensor([v], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    def timed(fn, iters):
        for _ in range(2):
            fn()
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        t0 = time.perf_counter()
        for _ in range(iters):
            fn()
        torch.cuda.synchronize()
        return sync_max((time.perf_counter() - t0) / iters * 1e3)

    def allreduce_alone(flat):
        if dist is None:
            return 0.0
        buf = torch.zeros_like(flat)
        return timed(lambda: dist.all_reduce(buf), 5)

    out = {}
    for net, B, odt in (("resnet50", 8, torch.float32), ("resnet50", 64, torch.float32), ("resnet50", 8, torch.bfloat16), ("resnet50", 64, torch.bfloat16)):
        H, W = 256, 448
        tr = SpatialCnnTrainer(net, lr=0.01, device=str(dev), operand_dtype=odt).load_state_dict(synth.fill_from_shapes(shapes.spatial_cnn_shapes(net), seed=5))
        frames = synth.synthetic_frames(B, H, W, seed=1 + rank).to(dev)
        z = torch.cat([torch.from_numpy((synth.uniform01(5 + rank, i, B * k) < 0.15).reshape(B, k).astype(np.float32)) for i, k in
                       enumerate((6, 10, 15, 100))], 1).to(dev)
        tp = [synth.synthetic_features(B, k, seed=11 + i)[0].to(dev) for i, k in enumerate((6, 10, 15))]
        tf = [synth.synthetic_features(B, 1536, seed=21 + i)[0].to(dev) for i in range(3)]
        iters = 10 if B == 8 else 4
        ms = timed(lambda: tr.train_step(frames, z, tp, tf), iters)                       # eager: bucketed all-reduce behind the backward
        tr.exchange = False
        ms_local = timed(lambda: tr.train_step(frames, z, tp, tf), iters)
        tr.exchange = True
        ar = allreduce_alone(tr.G)
        exposed = max(0.0, ms - ms_local)
        out[f"spatial_{net}_b{B}" + ("_bf16" if odt == torch.bfloat16 else "")] = dict(frames_per_s=round(world * B / ms * 1e3, 1), ms_per_step=round(ms, 3), ms_per_step_no_exchange=round(ms_local, 3),
                                          allreduce_alone_ms=round(ar, 3), grad_MB=round(tr.G.numel() * 4 / 1e6, 1),
                                          overlap_fraction=round(1.0 - min(1.0, exposed / ar), 3) if ar > 0 else None,
                                          frames_per_rank=B, dtype="f32" if odt == torch.float32 else "bf16 GEMM operands, fp32 master weights / sums",
                                          mode="eager, bucketed all-reduce issued behind the backward")
        # the same step under hipGraph replay: the backward replayed in segments cut where a bucket is complete, its all-reduce issued between
        # two replays (graph.SegmentedGraph); exchange off = the single-graph replay
        ms_g = timed(lambda: tr.train_step(frames, z, tp, tf, use_graph=True), iters)
        tr.exchange = False
        ms_g_local = timed(lambda: tr.train_step(frames, z, tp, tf, use_graph=True), iters)
        tr.exchange = True
        exposed = max(0.0, ms_g - ms_g_local)
        out[f"spatial_{net}_b{B}" + ("_bf16" if odt == torch.bfloat16 else "") + "_graph"] = dict(
            frames_per_s=round(world * B / ms_g * 1e3, 1), ms_per_step=round(ms_g, 3), ms_per_step_no_exchange=round(ms_g_local, 3),
            overlap_fraction=round(1.0 - min(1.0, exposed / ar), 3) if ar > 0 else None,
            mode="hipGraph replay in segments (one per gradient bucket), the bucket's all-reduce issued between replays")
        del tr
    T = 1000
    tr = TencoTrainer(lr=0.01, device=str(dev)).load_state_dict(synth.fill_from_shapes(shapes.tenco_shapes(), seed=47))
    xt = synth.synthetic_features(T, 512, seed=47 + rank).to(dev)
    zl = tr.prepare_labels({s: torch.from_numpy((synth.uniform01(3 + rank, i, T * k) < 0.1).reshape(T, k).astype(np.int64))
                            for i, (s, k) in enumerate((("", 100), ("_i", 6), ("_v", 10), ("_t", 15)))})
    ms = timed(lambda: tr.train_step(xt, zl, use_graph=True), 10)
    tr.exchange = False
    ms_local = timed(lambda: tr.train_step(xt, zl, use_graph=True), 10)
    tr.exchange = True
    ar = allreduce_alone(tr.G)
    exposed = max(0.0, ms - ms_local)
    out[f"tenco4_T{T}"] = dict(videos_per_s=round(world / ms * 1e3, 2), ms_per_step=round(ms, 3), ms_per_step_no_exchange=round(ms_local, 3),
                               allreduce_alone_ms=round(ar, 3), grad_MB=round(tr.G.numel() * 4 / 1e6, 1),
                               overlap_fraction=round(1.0 - min(1.0, exposed / ar), 3) if ar > 0 else None,
                               mode="hipGraph replay in segments (heads, Rs.2, Rs.1, Rs.0, PG), a stage's all-reduce issued between replays")
    ms_e = timed(lambda: tr.train_step(xt, zl, use_graph=False), 10)                        # eager: per-stage buckets behind the backward
    tr.exchange = False
    ms_e_local = timed(lambda: tr.train_step(xt, zl, use_graph=False), 10)
    tr.exchange = True
    exposed = max(0.0, ms_e - ms_e_local)
    out[f"tenco4_T{T}_eager_buckets"] = dict(videos_per_s=round(world / ms_e * 1e3, 2), ms_per_step=round(ms_e, 3), ms_per_step_no_exchange=round(ms_e_local, 3),
                                             overlap_fraction=round(1.0 - min(1.0, exposed / ar), 3) if ar > 0 else None,
                                             mode="eager launches, per-stage all-reduce (heads, Rs.2, Rs.1, Rs.0, PG) issued behind the backward")
    return out


METRIC = "frames/sec spatial-extract + per-video TCN latency, Cholec80, 1/2/4/8 GPU"


def main():
    a = parse()
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        self_launch(a)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if os.environ.get("MT4_BENCH_PLUMBING") == "1":
        return plumbing_only(a, world, rank)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback for the product path)")
    backend = os.environ.get("MT4_BENCH_BACKEND", "nccl")    # "gloo": rehearsal of the N>1 path with several ranks on ONE GPU
    local = local % torch.cuda.device_count()
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    dist = None
    if world > 1:
        import torch.distributed as dist_
        dist = dist_
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)   # RCCL
        else:
            dist.init_process_group(backend)

    from computervision_codes_amd import shapes, synth
    from computervision_codes_amd.spatial_cnn import VideoNas

    dtype = torch.bfloat16 if a.dtype == "bf16" else torch.float32
    args = types.SimpleNamespace(network=a.network, loss_type="all", student_dim=shapes.resnet_feat_dim(a.network), teacher_dim=1536,
                                 train=False)
    model = VideoNas(args=args, dtype=dtype, device=str(dev)).eval()
    model.load_state_dict(synth.fill_from_shapes(shapes.spatial_cnn_shapes(a.network), seed=1234))
    # every rank gets its own disjoint frames (seeded by rank), resident in HBM before timing starts.  64 frames come from the
    # counter generator on the host, the rest of the batch is derived on the device (device_frames)
    nstep = a.batch * max(1, a.streams)      # frames of one step on this GPU: --batch per stream
    frames = device_frames(nstep, a.height, a.width, 1234 + rank, dev, nbase=64)

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(a.warmup):
        model.extract_u8(frames, streams=a.streams)
    barrier()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        out = model.extract_u8(frames, streams=a.streams)
    barrier()
    dt = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([dt], device=dev if backend == "nccl" else "cpu", dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    assert torch.isfinite(out[3][1]).all()

    ms_per_step = dt / a.steps * 1e3
    fps = world * nstep * a.steps / dt

    def rank0_record():
        """everything beyond the timed region that only rank 0 computes (per-launch roofline, single-GPU sub-records)"""
        flops_frame = conv_flops_per_frame(model, a.height, a.width)
        per_launch = time_conv_kernels(model, frames[:a.batch].contiguous())   # the launches of ONE stream's part of a step
        conv_ms = sum(per_launch)
        peak = PEAK_BF16_TFLOPS if dtype == torch.bfloat16 else PEAK_F32_MFMA_TFLOPS
        # `achieved` / `frac` follow from the TIMED number: algorithmic conv FLOPs of the step's frames / ms_per_step (everything the step does --
        # preprocess, avgpool, heads -- counts against it).  The conv-launches-only figure (HIP events around each launch) is kept beside it.
        achieved = flops_frame * nstep / (ms_per_step * 1e-3) / 1e12
        achieved_conv = flops_frame * a.batch / (conv_ms * 1e-3) / 1e12
        # HBM bytes and MFMA-busy share of the same launches from rocprofv3 PMC passes (tools/collect_traffic.py, tools/collect_mfma_util.py;
        # commands in profiles/README.md), collected OFFLINE and committed: a figure is published only while the sources of the measured
        # kernels are the ones it was collected on (`kernels_sha`), and the line says which collection it is; null otherwise
        from computervision_codes_amd.srcdigest import CONV_SOURCES, TCN_SOURCES, kernels_digest

        def committed(fname, field, sources, keys):
            """-> (value per frame or video unit, source string)"""
            path = os.path.join(ROOT, "profiles", fname)
            try:
                data = json.load(open(path))
            except Exception:
                return None, "not collected"
            for key, scale in keys:
                rec = data.get(key)
                if rec is None or field not in rec:
                    continue
                now = kernels_digest(sources)
                if rec.get("kernels_sha") != now:
                    return None, f"stale: profiles/{fname}[{key}] was collected on kernel sources {rec.get('kernels_sha')}, this library is built from {now}"
                return rec[field] * scale, f"committed PMC pass profiles/{fname}[{key}] ({rec.get('collected', 'this round')}) @kernels {now}; offline, not measured in this run"
            return None, "not collected for this configuration"

        cfg = f"{a.network}_{a.dtype}_b{{}}_{a.height}x{a.width}"
        # (per-frame figures: the conv launches' traffic scales with the frame count; the 1336-frame collection serves any multiple of it)
        keys = [(cfg.format(a.batch), max(1, a.streams))] + ([(cfg.format(1336), nstep / 1336.0)] if a.batch % 1336 == 0 else [])
        traffic, traffic_source = committed("traffic.json", "hbm_bytes_per_step", CONV_SOURCES, keys)
        mfma_util, mfma_util_source = committed("mfma_util.json", "mfma_util", CONV_SOURCES, [(k, 1.0) for k, _ in keys])
        roofline = dict(bound="mfma", kernel="conv launches of one step (igemm_conv_kernel, conv3x3_patch_kernel, stem_pool_kernel, bottleneck64_fused_kernel, chain_gemm_kernel)", achieved=round(achieved, 2),
                        peak=peak, unit="TFLOP/s", frac=round(achieved / peak, 4), traffic=traffic, traffic_source=traffic_source,
                        mfma_util_pmc=mfma_util, mfma_util_source=mfma_util_source,
                        launches_per_step=len(per_launch), conv_ms_per_step=round(conv_ms, 4), ms_per_step=round(ms_per_step, 4),
                        achieved_conv_launches=round(achieved_conv, 2), frac_conv_launches=round(achieved_conv / peak, 4),
                        frames_in_conv_ms=a.batch,
                        gflop_per_frame=round(flops_frame / 1e9, 3), traffic_unit="HBM bytes per step (all conv launches)",
                        note="achieved / frac = algorithmic conv FLOPs of the step / ms_per_step (the timed region); *_conv_launches = the same FLOPs / the "
                             "summed durations of the step's conv launches (HIP events on the launch stream, one pair per launch): "
                             "conv_ms_per_step <= ms_per_step, the rest is preprocess + avgpool + heads")
        if a.per_layer:
            plan = model.conv_plan(a.height, a.width)
            groups = model.launch_groups(a.height, a.width)
            assert len(groups) + 1 == len(per_launch), (len(groups), len(per_launch))   # (the last launch is the heads GEMM)
            rows = []
            for g, ms in zip(groups, per_launch):
                fl = sum(2 * a.batch * plan[i]["Ho"] * plan[i]["Wo"] * plan[i]["Cout"] * plan[i]["Cin"] * plan[i]["kh"] * plan[i]["kw"] for i in g)
                blk = lambda nm: nm.rsplit(".", 1)[0]
                if len(g) == 2 and blk(plan[g[0]]["name"]) == blk(plan[g[1]]["name"]):
                    gname = plan[g[0]]["name"] + " + " + ("downsample" if plan[g[1]]["name"].endswith(".ds") else plan[g[1]]["name"].rsplit(".", 1)[1])
                elif len(g) == 2:
                    gname = plan[g[0]]["name"] + " + " + plan[g[1]]["name"]
                else:
                    gname = blk(plan[g[0]]["name"]) + (" + " + plan[g[-1]]["name"] if blk(plan[g[-1]]["name"]) != blk(plan[g[0]]["name"]) else "")
                rec = dict(plan[g[0]]) if len(g) == 1 else dict(name=gname + " (one launch)", fused=[plan[i]["name"] for i in g])
                rows.append(dict(rec, ms=round(ms, 4), tflops=round(fl / (ms * 1e-3) / 1e12, 1)))
            os.makedirs(os.path.dirname(os.path.abspath(a.per_layer)), exist_ok=True)
            json.dump(rows, open(a.per_layer, "w"), indent=1)
        res = {
            "metric": METRIC,
            "value": round(fps, 1), "unit": "frames/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": round(ms_per_step, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": a.dtype, "data": "synthetic",
            "config": {"workload": f"Spatial_cnn {a.network} extractor, {a.height}x{a.width} frames, {a.dtype}, eval "
                                   f"(BASELINE configs[1])", "frames_per_gpu_per_step": nstep, "streams_per_gpu": a.streams,
                       "frames_per_stream": a.batch, "global_frames_per_step": nstep * world,
                       "parallelism": f"frame-sharded x{world}, no collective"},
            "roofline": roofline,
        }
        if world == 1 and not a.no_temporal:
            res["temporal"] = temporal_bench(dev, do_cpu=not a.no_cpu_baseline)
            # the per-video latency half of the metric, where the driver's parsed record keeps it
            t256, t2000 = res["temporal"]["tenco4_T256"], res["temporal"]["tenco4_T2000"]
            tcn_traffic, tcn_src = committed("traffic.json", "hbm_bytes_per_video", TCN_SOURCES, [("tenco4_f32_T256", 1.0)])
            # SCALAR keys (the driver's parsed record keeps scalars of `roofline` only)
            roofline.update(
                tcn_kernel="tcn_conv_kernel / igemm_conv_kernel launches of one Temporal_tenco forward (4 stages, D = 512), hipGraph replay",
                tcn_bound="hbm", tcn_T256_ms=t256["ms_per_video"], tcn_T256_bf16_ms=t256["ms_per_video_bf16"], tcn_T2000_ms=t2000["ms_per_video"],
                tcn_T2000_bf16_ms=t2000["ms_per_video_bf16"], tcn_algorithmic_MB_T256=t256["algorithmic_MB"],
                tcn_hbm_frac_T256=t256["hbm_frac"], tcn_f32_mfma_frac_T256=t256["f32_mfma_frac"], tcn_hbm_frac_T2000=t2000["hbm_frac"],
                tcn_f32_mfma_frac_T2000=t2000["f32_mfma_frac"],
                tcn_traffic_MB_T256=round(tcn_traffic / 1e6, 1) if tcn_traffic else None,
                tcn_traffic_ratio_T256=round(tcn_traffic / 1e6 / t256["algorithmic_MB"], 2) if tcn_traffic else None, tcn_traffic_source=tcn_src,
                tcn_cpu_T256_ms=t256.get("cpu_ms_per_video"), tcn_cpu_T2000_ms=t2000.get("cpu_ms_per_video"),
                config1_T256_ms=res["temporal"]["config1_T256"]["ms_per_video"],
                mstct_T256_ms=res["temporal"]["mstct_T256"]["ms_per_window"])
            for k_, v_ in (res["temporal"].get("tenco4_throughput") or {}).items():
                roofline[f"tcn_videos_per_s_{k_}"] = v_["videos_per_s"]
            if dtype == torch.bfloat16 and a.network == "resnet50":
                res["native_256x448"] = native_resolution_bench(dev, model, a.streams)
                res["student_resnet18"] = student_resnet18_bench(dev, a.streams)
            res["swin_q2l"] = swin_bench(dev)
            sw = res["swin_q2l"].get("swin_B_384_22k_all") or {}
            swl = res["swin_q2l"].get("swin_L_384_22k_i") or {}
            roofline.update(swin_b384_all_fps=sw.get("frames_per_s"), swin_b384_all_frac=sw.get("mfma_frac"), swin_b384_all_frames_per_forward=sw.get("batch"),
                            swin_l384_teacher_fps=swl.get("frames_per_s"), swin_l384_teacher_frac=swl.get("mfma_frac"))
            res["spatial_train"] = spatial_train_bench(dev)
            st = res["spatial_train"]
            roofline.update(train_resnet50_b64_bf16_fps=(st.get("resnet50_b64_256x448_bf16") or {}).get("frames_per_s"),
                            train_resnet50_b256_bf16_fps=(st.get("resnet50_b256_256x448_bf16") or {}).get("frames_per_s"),
                            train_resnet18_b256_bf16_fps=(st.get("resnet18_b256_256x448_bf16") or {}).get("frames_per_s"))
            res["e2e_script"] = e2e_script_bench(dev)
            res["parity_mode_f32"] = parity_mode_bench(dev, a.network, a.height, a.width)
        if world == 1 and not a.no_cpu_baseline:
            res["cpu_baseline"] = cpu_baseline_spatial(a.network, a.height, a.width, 1234)
            res["gpu_over_cpu"] = round(fps / res["cpu_baseline"]["value"], 1)
        return res

    res = None
    if rank == 0:
        try:
            res = rank0_record()
        except Exception as e:   # the other ranks are about to enter a collective: an exception here must not leave them waiting for rank 0
            import traceback
            traceback.print_exc()
            res = {"metric": METRIC, "value": round(fps, 1), "unit": "frames/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
                   "ms_per_step": round(ms_per_step, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": a.dtype,
                   "data": "synthetic", "config": {"workload": f"Spatial_cnn {a.network} extractor, {a.height}x{a.width} frames, {a.dtype}, eval (BASELINE configs[1])"},
                   "roofline": None, "error": f"rank-0 sub-records failed: {type(e).__name__}: {e}"}
    if world > 1 and not a.no_ddp_train:
        # the headline line must survive the sub-record: an exception is caught below; a collective that never returns (the RCCL exchange has
        # not run on more than one GPU before the driver's scaling run) is cut by a timer -- rank 0 prints the line it has, every rank leaves
        def give_up():
            if rank == 0:
                res["ddp_train"] = {"error": f"no result after {a.ddp_timeout} s (a collective did not return); the headline above was complete before this leg"}
                print(json.dumps(res), flush=True)
            os._exit(0)
        timer = threading.Timer(a.ddp_timeout, give_up)
        timer.daemon = True
        timer.start()
        try:
            rec = ddp_train_bench(dev, dist, world, rank)  # every rank takes part (the exchange is collective)
        except Exception as e:
            rec = {"error": f"{type(e).__name__}: {e}"}
        if rank == 0:
            res["ddp_train"] = rec
        if dist is not None:
            dist.barrier()
        timer.cancel()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(res))


if __name__ == "__main__":
    main()
