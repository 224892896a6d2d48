"""The multi-GPU path of bench.py on ONE rank: under torchrun the process group is RCCL
(`init_process_group("nccl")`), the outputs go through `sharding.gather_outputs` and the closed loop through the sharded C5
path -- so an 8-GPU driver run does not meet any of them cold.  (World size 2 is covered on CPU with gloo:
tests/test_sharding_gloo.py.)"""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_bench_under_torchrun_with_one_rank():
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
           "--master-port", "29571", os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "1", "--warmup", "0", "--batch", "1024",
           "--no-cpu-baseline", "--closed-loop-rollouts", "64"]
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert line["n_gpus"] == 1 and line["value"] > 0 and line["success_rate"] > 0.99
    c5 = line["closed_loop"]                                   # the sharded closed loop (all_reduce of the counts, max of the times)
    assert "error" not in c5 and c5["converged_steps"] > 0.8 * 64 * 30


@pytest.mark.gpu
def test_bench_under_torchrun_with_two_ranks():
    """The first N > 1 RCCL run must not be the driver's 8-GPU bench: on a box with at least two GPUs the same command runs
    with two ranks (one per GPU, contiguous shards, the packed all_gather of sharding.gather_outputs over xGMI).  The 1-GPU
    boxes of this pool skip it."""
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs (this box has %d)" % torch.cuda.device_count())
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29573", os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0", "--batch", "1024",
           "--no-cpu-baseline", "--closed-loop-rollouts", "64"]
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=1200)
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["scaling"] == "weak" and line["success_rate"] > 0.99
    assert line["config"]["batch_per_gpu"] == 1024 and line["config"]["parallelism"] == "shard2"
    c5 = line["closed_loop"]
    assert "error" not in c5 and c5["converged_steps"] > 0.8 * 2 * 64 * 30
