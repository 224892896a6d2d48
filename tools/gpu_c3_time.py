"""dev tool: time of one launch of B config-C3 instances (N = 20) on the four-wavefront kernel, both halves; with a second
argument the results are also compared with the lane kernel (independent implementation of the same algorithm)"""
import sys, time
import numpy as np, torch
sys.path.insert(0, '.')
import os
from vehicle_motion_planning_with_obstacles_avoidance_using_mpc_amd import _lib, scenarios as sc
if os.environ.get("OBCA_LIB"):      # an alternative build of the library (file name inside the package)
    _lib.LIB_PATH = os.path.join(_lib.HERE, os.environ["OBCA_LIB"])
from vehicle_motion_planning_with_obstacles_avoidance_using_mpc_amd.solver import BatchSolver, SolverParams
B = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
check = len(sys.argv) > 2
N = 20
for gated in (False, True):
    b = sc.make_batch_c3(B, N, gated=gated, procs=8)
    dv = {k: torch.as_tensor(b[k], device="cuda") for k in ("variant", "x0", "u0", "xref", "A", "b", "Ts", "term")}
    res = {}
    for m in (["multiwave", "lane"] if check else ["multiwave"]):
        s = BatchSolver(N, b["m"], max_batch=B, mode=m)
        out, ts = None, []
        for _ in range(3 if m == "multiwave" else 1):
            torch.cuda.synchronize(); t = time.perf_counter()
            out = s.solve(dv["variant"], dv["x0"], dv["u0"], dv["xref"], dv["A"], dv["b"], dv["Ts"], dv["term"], SolverParams(), out=out)
            torch.cuda.synchronize(); ts.append(time.perf_counter() - t)
        it, st, x = out.iters.cpu().numpy(), out.status.cpu().numpy(), out.xopt.cpu().numpy()
        ok = np.isin(st, (0, 1))
        print("gated=%d %-9s %.1f ms  converged %.4f  -> %.0f converged solves/s, mean it %.1f" % (gated, m, min(ts) * 1e3, ok.mean(), ok.sum() / min(ts), it.mean()), flush=True)
        res[m] = (it, st, x)
        s.close()
    if check:
        a, l = res["multiwave"], res["lane"]
        same = a[0] == l[0]
        print("   vs lane: verdict differs %d, iters same %.3f, max|dx| where same %.2e" % ((np.isin(a[1], (0, 1)) != np.isin(l[1], (0, 1))).sum(), same.mean(), np.abs(a[2] - l[2])[same].max()))
