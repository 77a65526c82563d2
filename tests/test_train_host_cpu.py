"""CPU: host logic of the training path -- LR schedule vs torch's own schedulers, and the DDP exchange over gloo."""
import os

import pytest
import torch
import torch.multiprocessing as mp

from computervision_codes_amd.tenco_train import allreduce_sum_flat, lr_at_epoch


@pytest.mark.parametrize("lr,power,warmup,gamma", [(1e-2, 0.1, 200, 0.99), (1e-3, 0.1, 58, 0.999), (5e-3, 0.1, 9, 0.9)])
def test_lr_schedule_matches_torch_sequential_lr(lr, power, warmup, gamma):
    p = torch.nn.Parameter(torch.zeros(1))
    opt = torch.optim.SGD([p], lr=lr / power, weight_decay=1e-5)
    a = torch.optim.lr_scheduler.LinearLR(opt, start_factor=power, total_iters=warmup)
    b = torch.optim.lr_scheduler.ExponentialLR(opt, gamma=gamma)
    sch = torch.optim.lr_scheduler.SequentialLR(opt, schedulers=[a, b], milestones=[warmup + 1])
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        for epoch in range(warmup + 30):
            got = opt.param_groups[0]["lr"]
            want = lr_at_epoch(epoch, lr, power, warmup, gamma)
            assert abs(got - want) <= 1e-9 + 1e-6 * got, (epoch, got, want)
            opt.step()
            sch.step()


def _worker(rank, world, port, out):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        g = torch.arange(10, dtype=torch.float32) * (rank + 1)       # rank-dependent "gradients"
        scale = allreduce_sum_flat(g)
        torch.save((g * scale), os.path.join(out, f"g{rank}.pt"))
    finally:
        dist.destroy_process_group()


def test_ddp_exchange_world2_gloo(tmp_path):
    assert allreduce_sum_flat(torch.ones(4)) == 1.0                   # not initialised: single rank
    port = 29600 + (os.getpid() % 2000)
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    g0, g1 = torch.load(tmp_path / "g0.pt"), torch.load(tmp_path / "g1.pt")
    want = torch.arange(10, dtype=torch.float32) * 1.5                # mean of x1 and x2
    assert torch.equal(g0, g1) and torch.allclose(g0, want)
