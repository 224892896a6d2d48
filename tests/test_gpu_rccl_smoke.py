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
    """The first N > 1 RCCL run must not be the driver's 8-GPU bench: on a box with at least two GPUs a plain
    ``python bench.py --gpus 2`` (the driver's form; bench.launch_ranks re-runs it under torch.distributed.run) runs
    with two ranks (one per GPU, contiguous shards, the packed all_gather of sharding.gather_outputs over xGMI).  The 1-GPU
    boxes of this pool skip it."""
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs (this box has %d)" % torch.cuda.device_count())
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):                 # a bare call: bench.py starts its two ranks itself
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0", "--batch", "1024",
           "--no-cpu-baseline", "--closed-loop-rollouts", "64"]
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=1200)
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["scaling"] == "weak" and line["success_rate"] > 0.99
    assert line["config"]["batch_per_gpu"] == 1024 and line["config"]["parallelism"] == "shard2"
    c5 = line["closed_loop"]
    assert "error" not in c5 and c5["converged_steps"] > 0.8 * 2 * 64 * 30


def test_bare_gpus_n_never_prints_a_one_gpu_line():
    """``python bench.py --gpus 2`` on a node with fewer than two GPUs (this container has none, the pool's boxes one) exits
    non-zero with a message and prints NO JSON line -- it used to run one process on GPU 0 and report n_gpus 1.  Under a
    launcher whose world size contradicts --gpus it refuses as well."""
    import torch
    if torch.cuda.is_available() and torch.cuda.device_count() >= 2:
        pytest.skip("this node can run two ranks; covered by test_bench_under_torchrun_with_two_ranks")
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"], cwd=ROOT, env=env,
                       capture_output=True, text=True, timeout=600)
    assert r.returncode != 0 and "--gpus 2" in r.stderr and not [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"], cwd=ROOT,
                       env=dict(env, RANK="0", WORLD_SIZE="1", LOCAL_RANK="0"), capture_output=True, text=True, timeout=600)
    assert r.returncode != 0 and "WORLD_SIZE=1" in r.stderr and not [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
