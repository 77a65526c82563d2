"""CPU: `bench.py --gpus 2` starts its own two ranks (child `torch.distributed.run`, rendezvous on 127.0.0.1) and rank 0 prints ONE JSON
line with `n_gpus: 2` -- the N > 1 protocol of bench.py (self-launch, barrier-bracketed timed region, MAX over ranks) without the GPU
work (MT4_BENCH_PLUMBING=1, gloo).  The same line through the driver's own launch form (torch.distributed.run around bench.py)."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(cmd):
    env = dict(os.environ, MT4_BENCH_PLUMBING="1", MT4_BENCH_BACKEND="gloo")
    env.pop("WORLD_SIZE", None)
    env.pop("RANK", None)
    p = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=240)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout
    return json.loads(lines[0])


def test_bench_self_launches_two_ranks():
    rec = _run([sys.executable, "bench.py", "--gpus", "2", "--steps", "3", "--warmup", "1"])
    assert rec["n_gpus"] == 2 and rec["steps"] == 3 and rec["plumbing_only"] is True
    # rank 1 sleeps twice as long as rank 0: the reported time is the MAX over ranks
    assert rec["ms_per_step"] >= 3.9


def test_bench_under_torchrun_and_single_rank():
    rec = _run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                "--master-port", "29731", "bench.py", "--gpus", "2", "--steps", "2", "--warmup", "0"])
    assert rec["n_gpus"] == 2
    rec1 = _run([sys.executable, "bench.py", "--steps", "2", "--warmup", "0"])
    assert rec1["n_gpus"] == 1
