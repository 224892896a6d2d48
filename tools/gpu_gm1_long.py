"""dev tool (round 6, advisor): one wavefront per instance with the rows in HBM (global1 = obca_ipm_kernel_gm1) against four wavefronts
(global = obca_ipm_kernel_gm) on LONG horizons with at most three obstacles -- the region auto mode sends to gm1 although round 5 only
measured N = 12 .. 26 -- at batch sizes 1, 64 and 2048."""
import sys, time
import numpy as np, torch
sys.path.insert(0, '.')
from vehicle_motion_planning_with_obstacles_avoidance_using_mpc_amd import scenarios as sc
from vehicle_motion_planning_with_obstacles_avoidance_using_mpc_amd.solver import BatchSolver, SolverParams
for N in (32, 40, 56, 74):
    bfull = sc.make_batch_c3(2048, N, gated=False, procs=8)
    for B in (1, 64, 2048):
        a = [np.ascontiguousarray(bfull[k][:B]) for k in ("variant", "x0", "u0", "xref", "A", "b", "Ts", "term")]
        row = []
        for mode in ("global", "global1"):
            s = BatchSolver(N, bfull["m"], max_batch=B)
            s.set_mode(mode)
            o = s.solve(*a, SolverParams()); torch.cuda.synchronize()
            t = time.perf_counter(); o = s.solve(*a, SolverParams()); torch.cuda.synchronize(); dt = time.perf_counter() - t
            ok = int(((o.status == 0) | (o.status == 1)).sum())
            row.append("%s %.1f ms (%d ok, %.0f it)" % (mode, dt * 1e3, ok, float(o.iters.float().mean())))
            s.close()
        print("N=%d B=%d: %s" % (N, B, " | ".join(row)), flush=True)
