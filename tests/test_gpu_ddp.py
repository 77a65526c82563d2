"""GPU: data-parallel training semantics end to end with two ranks (both on cuda:0, gloo transport): after one step every rank holds
sgd(mean of the ranks' gradients), with the bucketed all-reduce overlapped with the backward and with the single flat all-reduce."""
import os
import subprocess
import sys

import pytest
import torch

from computervision_codes_amd import shapes, synth

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("overlap", [1, 0, "graph"])
def test_spatial_student_two_rank_step_equals_mean_gradient_step(cuda, tmp_path, overlap):
    """overlap 1: eager step, bucketed all-reduce behind the backward; 0: one flat all-reduce; "graph": hipGraph replay in SEGMENTS cut where a
    bucket is complete, the bucket's all-reduce issued between two replays (`graph.SegmentedGraph`) -- all three end at sgd(mean gradient)"""
    sys.path.insert(0, os.path.join(ROOT, "tests", "helpers"))
    from ddp_spatial_worker import batch
    from computervision_codes_amd.spatial_cnn_train import SpatialCnnTrainer
    env = dict(os.environ, PYTHONPATH=ROOT, MASTER_ADDR="127.0.0.1")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", str(29530 + (2 if overlap == "graph" else overlap)), os.path.join(ROOT, "tests", "helpers", "ddp_spatial_worker.py"), str(tmp_path), str(overlap)],
                       env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-3000:]
    got = torch.load(tmp_path / f"ddp_overlap{overlap}.pth", map_location="cpu")
    order = got.pop("__bucket_order__")
    if overlap in (1, "graph"):          # buckets leave in the order the backward completes them
        assert order == ["heads", "layer4", "layer3", "layer2", "layer1", "stem"], order
    # expected: gradients of the two batches from the same start, averaged, one SGD step; rank 0's BatchNorm statistics
    sd = synth.fill_from_shapes(shapes.spatial_cnn_shapes("resnet18"), seed=9)
    trs = []
    for rank in (0, 1):
        tr = SpatialCnnTrainer("resnet18", lr=0.05).load_state_dict(sd)
        fr, labels, tp, tf = batch(rank)
        tr.train_step(fr.to(cuda), labels, tp, tf, apply_update=False)
        trs.append(tr)
    trs[0].G.add_(trs[1].G).mul_(0.5)
    trs[0].apply_update()
    want = trs[0].state_dict()
    for k in want:
        a, b = got[k].float(), want[k].float()
        assert (a - b).abs().max().item() <= 2e-6 * max(1.0, b.abs().max().item()), k


@pytest.mark.parametrize("mode", ["eager", "graph"])
def test_tenco_two_rank_step_equals_mean_gradient_step(cuda, tmp_path, mode):
    """eager: per-stage all-reduce behind the backward; graph: the same buckets under hipGraph replay (one segment per bucket)"""
    sys.path.insert(0, os.path.join(ROOT, "tests", "helpers"))
    from ddp_tenco_worker import trainer, video
    env = dict(os.environ, PYTHONPATH=ROOT, MASTER_ADDR="127.0.0.1")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", "29533" if mode == "eager" else "29536", os.path.join(ROOT, "tests", "helpers", "ddp_tenco_worker.py"), str(tmp_path), mode],
                       env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-3000:]
    got = torch.load(tmp_path / "ddp_tenco.pth", map_location="cpu")
    assert got.pop("__bucket_order__") == ["heads", "Rs.2", "Rs.1", "Rs.0", "PG"]
    assert got.pop("__segments__") == (6 if mode == "graph" else 0)
    trs = []
    for rank in (0, 1):
        tr = trainer()
        x, labels = video(rank)
        tr.train_step(x.to(cuda), labels, apply_update=False)
        trs.append(tr)
    trs[0].G.add_(trs[1].G).mul_(0.5)
    trs[0].apply_update()
    want = trs[0].state_dict()
    for k in want:
        assert (got[k].float() - want[k].float()).abs().max().item() <= 2e-6 * max(1.0, want[k].float().abs().max().item()), k


def test_mstct_two_rank_step_equals_mean_gradient_step(cuda, tmp_path):
    sys.path.insert(0, os.path.join(ROOT, "tests", "helpers"))
    from ddp_mstct_worker import trainer, windows
    env = dict(os.environ, PYTHONPATH=ROOT, MASTER_ADDR="127.0.0.1")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", "29534", os.path.join(ROOT, "tests", "helpers", "ddp_mstct_worker.py"), str(tmp_path)],
                       env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-3000:]
    got = torch.load(tmp_path / "ddp_mstct.pth", map_location="cpu")
    trs = []
    for rank in (0, 1):
        tr = trainer()
        x, y = windows(rank)
        tr.train_step(x.to(cuda), y, apply_update=False)
        trs.append(tr)
    trs[0].G.add_(trs[1].G).mul_(0.5)
    trs[0].apply_update()
    want = trs[0].state_dict()
    for k in want:
        assert (got[k].float() - want[k].float()).abs().max().item() <= 2e-6 * max(1.0, want[k].float().abs().max().item()), k


def test_q2l_two_rank_step_equals_mean_gradient_step(cuda, tmp_path):
    """uint8 frames, every rank its own DropPath / dropout draw (device counter generator, seed per rank): the two-rank step equals one SGD step
    on the mean of the two single-rank gradient buffers"""
    sys.path.insert(0, os.path.join(ROOT, "tests", "helpers"))
    from ddp_q2l_worker import CFG, frames, trainer
    env = dict(os.environ, PYTHONPATH=ROOT, MASTER_ADDR="127.0.0.1")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", "29535", os.path.join(ROOT, "tests", "helpers", "ddp_q2l_worker.py"), str(tmp_path)],
                       env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-3000:]
    got = torch.load(tmp_path / "ddp_q2l.pth", map_location="cpu")
    trs = []
    for rank in (0, 1):
        tr = trainer()
        img, y = frames(rank)
        tr.train_step(img.to(cuda), y, masks=tr.draw_masks_device(CFG["B"], 77 + rank, 0), apply_update=False)
        trs.append(tr)
    trs[0].G.add_(trs[1].G).mul_(0.5)
    trs[0].apply_update()
    want = trs[0].state_dict()
    for k in want:
        assert (got[k].float() - want[k].float()).abs().max().item() <= 2e-6 * max(1.0, want[k].float().abs().max().item()), k
