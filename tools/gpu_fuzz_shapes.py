"""dev tool: every kernel family on many small shapes (horizons 1..12, the three workload generators): verdict agreement,
NaNs, and crashes -- a smoke screen for shape-dependent code paths (short horizons, one-sided / two-sided sweep, row-slot counts)"""
import sys
import numpy as np, torch
sys.path.insert(0, '.')
from vehicle_motion_planning_with_obstacles_avoidance_using_mpc_amd import scenarios as sc
from vehicle_motion_planning_with_obstacles_avoidance_using_mpc_amd.solver import BatchSolver, SolverParams

B = 32
bad = 0
for N in range(1, 13):
    for name, gen in (("c2", lambda: sc.make_batch(B, N)), ("c2-3box", lambda: sc.make_batch(B, N, three_boxes=True)),
                      ("c3free", lambda: sc.make_batch_c3(B, N, gated=False)), ("c3gated", lambda: sc.make_batch_c3(B, N, gated=True))):
        try:
            b = gen()
        except Exception as e:
            print("N=%2d %-8s generator: %s" % (N, name, str(e)[:50])); continue
        res = {}
        for mode, two in (("wave", None), ("multiwave", False), ("multiwave", True), ("lane", None)):
            try:
                s = BatchSolver(N, b["m"], max_batch=B, mode=mode)
                s.set_two_sided_sweep(two)
                o = s.solve(b["variant"], b["x0"], b["u0"], b["xref"], b["A"], b["b"], b["Ts"], b["term"], SolverParams())
                torch.cuda.synchronize()
                res[(mode, two)] = (o.xopt.cpu().numpy(), o.status.cpu().numpy(), o.iters.cpu().numpy())
                s.close()
            except RuntimeError as e:
                res[(mode, two)] = None
        ref = res.get(("wave", None)) or res.get(("multiwave", False))
        line = "N=%2d %-8s" % (N, name)
        for k, v in res.items():
            if v is None:
                line += " | %s/%s n/a" % k; continue
            ok, okr = np.isin(v[1], (0, 1)), np.isin(ref[1], (0, 1))
            nan = int(np.isnan(v[0]).sum())
            flips = int((ok != okr).sum())
            both = ok & okr
            d = np.abs(v[0] - ref[0]).reshape(B, -1).max(1)
            far = int((d[both] > 1e-5).sum())
            bad += nan + (flips > 2)
            line += " | %s/%s ok %d flips %d far %d nan %d" % (k[0][:5], k[1], ok.sum(), flips, far, nan)
        print(line, flush=True)
print("suspicious:", bad)
