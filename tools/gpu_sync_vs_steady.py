"""dev tool: per-iteration cost of the headline kernel when every wavefront of the chip starts at the same time (B = 1024: one round, the
launch lasts as long as its longest instance) against the steady state (B = 8192: eight rounds, wavefronts of a CU at different phases
of the solve) -- and whether sustained load explains the difference (B = 1024 launched 200 times back to back).
    python tools/gpu_sync_vs_steady.py"""
import sys
import time

import numpy as np
import torch

sys.path.insert(0, ".")
from vehicle_motion_planning_with_obstacles_avoidance_using_mpc_amd import scenarios as sc                      # noqa: E402
from vehicle_motion_planning_with_obstacles_avoidance_using_mpc_amd.solver import BatchSolver, SolverParams      # noqa: E402

N = 5
full = sc.make_batch(8192, N)
for B, reps in ((1024, 5), (1024, 200), (2048, 50), (4096, 20), (8192, 10)):
    b = {k: (v[:B] if hasattr(v, "shape") and v.shape and v.shape[0] == 8192 else v) for k, v in full.items()}
    s = BatchSolver(N, b["m"], max_batch=B)
    dv = {k: torch.as_tensor(b[k], device="cuda") for k in ("variant", "x0", "u0", "xref", "A", "b", "Ts", "term")}
    out = s.solve(dv["variant"], dv["x0"], dv["u0"], dv["xref"], dv["A"], dv["b"], dv["Ts"], dv["term"], SolverParams())
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        out = s.solve(dv["variant"], dv["x0"], dv["u0"], dv["xref"], dv["A"], dv["b"], dv["Ts"], dv["term"], SolverParams(), out=out)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    it = out.iters.cpu().numpy().astype(float)
    nf = out.info[:, 3].cpu().numpy()
    rounds = B / 1024.0
    per_slot = it.sum() / 1024.0                       # iterations one SIMD slot works through if the load were perfectly balanced
    print("B=%5d x %3d launches: %.3f ms per launch; iterations mean %.2f max %d (factorisations mean %.1f max %d); "
          "us per iteration if balanced: %.1f; launch / longest instance: %.1f us per iteration of the longest"
          % (B, reps, dt * 1e3, it.mean(), it.max(), nf.mean(), nf.max(), dt * 1e6 / per_slot, dt * 1e6 / it.max()))
    s.close()
