"""dev tool: which shapes run faster with one wavefront per instance and the rows in an HBM workspace (global1) than on the
four-wavefront LDS kernel (auto) -- shapes beyond the one-wavefront LDS kernel (> 384 rows), C3 generator at several horizons."""
import sys, time
import numpy as np, torch
sys.path.insert(0, '.')
from vehicle_motion_planning_with_obstacles_avoidance_using_mpc_amd import scenarios as sc
from vehicle_motion_planning_with_obstacles_avoidance_using_mpc_amd.solver import BatchSolver, SolverParams
B = 8192
for N, gated in ((12, False), (16, False), (20, False), (26, False), (8, True), (10, True), (12, True), (14, True)):
    b = sc.make_batch_c3(B, N, gated=gated, procs=8)
    a = [b[k] for k in ("variant", "x0", "u0", "xref", "A", "b", "Ts", "term")]
    row = []
    for mode in ("auto", "global1"):
        s = BatchSolver(N, b["m"], max_batch=B)
        try:
            s.set_mode(mode)
        except RuntimeError as e:
            row.append("%s: n/a" % mode); s.close(); continue
        o = s.solve(*a, SolverParams()); torch.cuda.synchronize()
        t = time.perf_counter(); o = s.solve(*a, SolverParams()); torch.cuda.synchronize(); dt = time.perf_counter() - t
        ok = int(((o.status == 0) | (o.status == 1)).sum())
        row.append("%s %.1f ms (%d ok)" % (mode, dt * 1e3, ok))
        s.close()
    M = sum(b["m"]); nO = len(b["m"])
    R = 3 + 3 * N + 3 + 2 * (N + 1) + 4 * N + 2 + (N + 1) * (2 * nO + M + 4 * nO)
    print("N=%d %s (%d obstacles, %d rows/stage): R_max %d | %s" % (N, "gated" if gated else "free", nO, M, R, " | ".join(row)), flush=True)
