"""dev tool: launch time of small batches, four wavefronts per instance against one (the kernel choice of auto mode depends on
the shape only; this measures what a latency-minded caller gains by asking for mode "multiwave")"""
import sys, time
import numpy as np, torch
sys.path.insert(0, '.')
from vehicle_motion_planning_with_obstacles_avoidance_using_mpc_amd import scenarios as sc
from vehicle_motion_planning_with_obstacles_avoidance_using_mpc_amd.solver import BatchSolver, SolverParams
for B in (1, 64, 256, 257, 1024):
    b = sc.make_batch(B, 5)
    res = {}
    for m in ("wave", "multiwave"):
        s = BatchSolver(5, b["m"], max_batch=B, mode=m)
        ts = []
        for _ in range(4):
            torch.cuda.synchronize(); t = time.perf_counter()
            o = s.solve(b["variant"], b["x0"], b["u0"], b["xref"], b["A"], b["b"], b["Ts"], b["term"], SolverParams())
            torch.cuda.synchronize(); ts.append(time.perf_counter() - t)
        res[m] = (min(ts), o.xopt.cpu().numpy(), o.iters.cpu().numpy())
        s.close()
    print("B=%4d  wave %.2f ms  multiwave %.2f ms  identical %s" % (B, res["wave"][0] * 1e3, res["multiwave"][0] * 1e3,
          np.array_equal(res["wave"][1], res["multiwave"][1]) and np.array_equal(res["wave"][2], res["multiwave"][2])), flush=True)
