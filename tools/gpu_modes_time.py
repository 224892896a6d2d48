"""dev tool: time of one launch of B C2 instances per kernel mode (wave / twowave / multiwave), results compared"""
import sys, time
import numpy as np, torch
sys.path.insert(0, '.')
from vehicle_motion_planning_with_obstacles_avoidance_using_mpc_amd import scenarios as sc
from vehicle_motion_planning_with_obstacles_avoidance_using_mpc_amd.solver import BatchSolver, SolverParams
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
modes = sys.argv[2].split(",") if len(sys.argv) > 2 else ["wave", "twowave"]
b = sc.make_batch(B, 5)
dv = {k: torch.as_tensor(b[k], device="cuda") for k in ("variant", "x0", "u0", "xref", "A", "b", "Ts", "term")}
ref = None
for m in modes:
    s = BatchSolver(5, b["m"], max_batch=B, mode=m)
    out = None
    ts = []
    for _ in range(4):
        torch.cuda.synchronize(); t = time.perf_counter()
        out = s.solve(dv["variant"], dv["x0"], dv["u0"], dv["xref"], dv["A"], dv["b"], dv["Ts"], dv["term"], SolverParams(), out=out)
        torch.cuda.synchronize(); ts.append(time.perf_counter() - t)
    it, st, x = out.iters.cpu().numpy(), out.status.cpu().numpy(), out.xopt.cpu().numpy()
    line = "%-9s %.2f ms  ok %.4f mean it %.2f" % (m, min(ts) * 1e3, np.isin(st, (0, 1)).mean(), it.mean())
    if ref is not None:
        line += "  | iters differ %d, max|dx| where same %.2e" % ((it != ref[0]).sum(), np.abs(x - ref[1])[it == ref[0]].max())
    else:
        ref = (it, x)
    print(line)
    s.close()
