"""dev tool: the gated / free-time halves of C3 (N = 20) on the four-wavefront LDS kernel (auto), the four-wavefront HBM-workspace
kernel (global) and the ONE-wavefront HBM-workspace kernel (global1): launch time, verdicts, and whether the words agree."""
import sys, time
import numpy as np, torch
sys.path.insert(0, '.')
from vehicle_motion_planning_with_obstacles_avoidance_using_mpc_amd import scenarios as sc
from vehicle_motion_planning_with_obstacles_avoidance_using_mpc_amd.solver import BatchSolver, SolverParams
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
for name, gated, N in (("C3 gated", True, 20), ("C3 free-time", False, 20), ("C2 (N = 5)", None, 5)):
    b = sc.make_batch(B, 5) if gated is None else sc.make_batch_c3(B, N, gated=gated, procs=8)
    a = [b[k] for k in ("variant", "x0", "u0", "xref", "A", "b", "Ts", "term")]
    ref = None
    for mode in ("auto", "global", "global1"):
        s = BatchSolver(N, b["m"], max_batch=B)
        s.set_mode(mode)
        if mode != "auto":
            s.set_two_sided_sweep(False)
        o = s.solve(*a, SolverParams()); torch.cuda.synchronize()
        t = time.perf_counter(); o = s.solve(*a, SolverParams()); torch.cuda.synchronize(); dt = time.perf_counter() - t
        st = o.status.cpu().numpy(); ok = np.isin(st, (0, 1))
        cur = {k: getattr(o, k).cpu().numpy() for k in ("xopt", "uopt", "ts_opt", "status", "iters")}
        same = "" if mode != "global1" else ("; words equal to 'global': %s" % all(np.array_equal(ref[k], cur[k]) for k in cur))
        if mode == "global":
            ref = cur
        print("%s %-8s: %.1f ms per %d, converged %d, lds %d B%s" % (name, mode, dt * 1e3, B, ok.sum(), s.lds_bytes, same), flush=True)
        s.close()
