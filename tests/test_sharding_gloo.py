"""world_size-2 run of the sharded batch path on CPU (gloo): shard -> solve locally -> one all_gather equals the
single-process result.  The local solve is the C oracle (test infrastructure) standing in for the GPU kernel."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from vehicle_motion_planning_with_obstacles_avoidance_using_mpc_amd import scenarios as sc
from vehicle_motion_planning_with_obstacles_avoidance_using_mpc_amd.sharding import gather_outputs, shard_bounds

TOTAL, N = 7, 5


def test_shard_bounds_cover_everything():
    for total in (1, 7, 8, 65536):
        for world in (1, 2, 3, 8):
            spans = [shard_bounds(total, world, r) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == total
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            assert max(h - l for l, h in spans) - min(h - l for l, h in spans) <= 1


def _solve(lo, hi):
    from oracle import c_oracle
    b = sc.make_batch(hi - lo, N, first=lo)
    o = c_oracle.solve_batch(4, N, b["m"], b["x0"], b["u0"], b["xref"], b["A"], b["b"], b["Ts"], threads=2)
    return {"xopt": torch.from_numpy(o["xopt"]), "ts_opt": torch.from_numpy(o["ts_opt"]),
            "status": torch.from_numpy(o["status"])}


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    lo, hi = shard_bounds(TOTAL, world, rank)
    full = gather_outputs(dist, _solve(lo, hi), TOTAL, world)
    if rank == 0:
        q.put({k: v.numpy() for k, v in full.items()})
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_gather_equals_single_process():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = q.get(timeout=300)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    ref = _solve(0, TOTAL)
    for k in ref:
        assert np.array_equal(got[k], ref[k].numpy()), k


# ---------------------------------------------------------------------------------------------------------------
# closed loop (config C5): every rank runs the whole receding-horizon loop for its slice of the rollouts, then ONE
# all_gather of the histories.  The local closed loop is the CPU build of the device harness + structured core
# (tests/native), standing in for obca_rollouts_run.
C5_TOTAL, C5_STEPS = 5, 4


def _c5_worlds():
    from vehicle_motion_planning_with_obstacles_avoidance_using_mpc_amd.rollouts import pack_worlds
    return pack_worlds([sc.make_world_c5(i, n_dyn=1) for i in range(C5_TOTAL)])


def _c5_run(w):
    from oracle import c_oracle
    from tests import native_build
    o = native_build.rollout_run(w, 5, c_oracle.default_params(), C5_STEPS, max_steps=C5_STEPS)
    return {k: torch.from_numpy(np.ascontiguousarray(o[k])) for k in ("x_closed", "u_closed", "T_closed", "variant", "status", "steps", "flags")}


def _c5_worker(rank, world, port, q):
    from vehicle_motion_planning_with_obstacles_avoidance_using_mpc_amd.sharding import gather_rollouts, shard_worlds
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    full = gather_rollouts(dist, _c5_run(shard_worlds(_c5_worlds(), world, rank)), C5_TOTAL, world)
    if rank == 0:
        q.put({k: v.numpy() for k, v in full.items()})
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_closed_loop_gather_equals_single_process():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_c5_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = q.get(timeout=600)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    ref = _c5_run(_c5_worlds())
    assert int(ref["steps"].sum()) > 0
    for k in ref:
        assert np.array_equal(got[k], ref[k].numpy()), k
