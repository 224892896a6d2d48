"""dev tool: auto mode's rule for shapes beyond the one-wavefront LDS kernel, the case not covered by the C3 generator: FOUR obstacles
(the gated instances with their last moving box dropped: 10 rows per stage), obca_mpc6 / obca_mpc8, N = 10 ... 20."""
import sys, time
import numpy as np, torch
sys.path.insert(0, '.')
from vehicle_motion_planning_with_obstacles_avoidance_using_mpc_amd import scenarios as sc
from vehicle_motion_planning_with_obstacles_avoidance_using_mpc_amd.solver import BatchSolver, SolverParams
B = 8192
for N in (10, 14, 20):
    b = sc.make_batch_c3(B, N, gated=True, procs=8)
    m = list(b["m"][:-1]); M = sum(m)
    A, bb = np.ascontiguousarray(b["A"][:, :, :M]), np.ascontiguousarray(b["b"][:, :, :M])
    for variant in (6, 8):
        a = [np.full(B, variant, np.int32), b["x0"], b["u0"], b["xref"], A, bb, b["Ts"], b["term"]]
        row = []
        for mode in ("multiwave", "global1"):
            s = BatchSolver(N, m, max_batch=B)
            s.set_mode(mode)
            o = s.solve(*a, SolverParams()); torch.cuda.synchronize()
            t = time.perf_counter(); o = s.solve(*a, SolverParams()); torch.cuda.synchronize(); dt = time.perf_counter() - t
            row.append("%s %.1f ms (%d ok)" % (mode, dt * 1e3, int(((o.status == 0) | (o.status == 1)).sum())))
            s.close()
        print("N=%d obca_mpc%d, 4 obstacles, %d rows/stage: %s" % (N, variant, M, " | ".join(row)), flush=True)
# ... and THREE obstacles with the fixed-time variants (both moving boxes dropped): is the three-obstacle rule a rule of the shape?
for N in (12, 20):
    b = sc.make_batch_c3(B, N, gated=True, procs=8)
    m = list(b["m"][:-2]); M = sum(m)
    A, bb = np.ascontiguousarray(b["A"][:, :, :M]), np.ascontiguousarray(b["b"][:, :, :M])
    for variant in (6, 8):
        a = [np.full(B, variant, np.int32), b["x0"], b["u0"], b["xref"], A, bb, b["Ts"], b["term"]]
        row = []
        for mode in ("multiwave", "global1"):
            s = BatchSolver(N, m, max_batch=B)
            s.set_mode(mode)
            o = s.solve(*a, SolverParams()); torch.cuda.synchronize()
            t = time.perf_counter(); o = s.solve(*a, SolverParams()); torch.cuda.synchronize(); dt = time.perf_counter() - t
            row.append("%s %.1f ms (%d ok)" % (mode, dt * 1e3, int(((o.status == 0) | (o.status == 1)).sum())))
            s.close()
        print("N=%d obca_mpc%d, 3 obstacles, %d rows/stage: %s" % (N, variant, M, " | ".join(row)), flush=True)
