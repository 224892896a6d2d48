"""dev tool (advisor, round 5: 'the dodge rung also runs for single-start calls -- measure C5 with and without'): the C5 closed loop
(4096 rollouts, two moving boxes) with the ladder's dodge rung on (default) and off, and with shorter iteration limits of the later
passes; per configuration the launch time, the stopped rollouts and where the iterations of the longest steps go.
    python tools/gpu_c5_dodge.py [B]"""
import sys
import time

import numpy as np
import torch

sys.path.insert(0, ".")
from vehicle_motion_planning_with_obstacles_avoidance_using_mpc_amd import scenarios as sc                      # noqa: E402
from vehicle_motion_planning_with_obstacles_avoidance_using_mpc_amd.rollouts import DeviceRollouts, pack_worlds  # noqa: E402
from vehicle_motion_planning_with_obstacles_avoidance_using_mpc_amd.solver import SolverParams                   # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
worlds = [sc.make_world_c5(i, n_dyn=2) for i in range(B)]
base = None
for name, kw in (("default", {}), ("dodge off", dict(dodge=False)), ("retry_iter 200", dict(retry_iter=200)), ("retry_iter 150", dict(retry_iter=150)),
                 ("retry_iter 100", dict(retry_iter=100)), ("retry 150 patience 300", dict(retry_iter=150, patience=300))):
    w = pack_worlds(worlds)
    dr = DeviceRollouts(w, N=5, params=SolverParams(xL=getattr(w, "xL", (0.0, 0.0)), xU=getattr(w, "xU", (39.0, 10.0)), **kw))
    dr.run(1); torch.cuda.synchronize(); dr.reset(); torch.cuda.synchronize()
    best = None
    for _ in range(2):
        dr.reset(); torch.cuda.synchronize()
        t0 = time.perf_counter(); dr.run(); torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        best = dt if best is None else min(best, dt)
    o = {k: v.cpu().numpy() for k, v in dr.read().items()}
    ok = int(o["steps"].sum())
    it = o["iters"].astype(np.int64) * (o["variant"] > 0)
    per_rollout = it.sum(1)
    top = np.sort(it.ravel())[::-1][:8]
    stopped = np.flatnonzero(o["flags"] == 3)
    last = np.array([it[b, o["steps"][b]] for b in stopped]) if len(stopped) else np.zeros(1)
    if base is None:
        base = o
    same = int(sum(np.array_equal(o["x_closed"][b, :min(o["steps"][b], base["steps"][b]) + 1], base["x_closed"][b, :min(o["steps"][b], base["steps"][b]) + 1]) for b in range(B)))
    print("%-24s %.4f s, %d converged steps -> %.0f steps/s, %d stopped; iterations: total %.3f M, longest rollout %d, longest steps %s, stopping steps mean %.0f max %d; "
          "rollouts with the default's closed-loop words up to the shorter run: %d of %d"
          % (name, best, ok, ok / best, len(stopped), it.sum() / 1e6, per_rollout.max(), top.tolist(), last.mean(), last.max(), same, B))
