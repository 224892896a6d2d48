"""dev tool (GPU box): the workloads of bench.py on an alternative build of the library, selected by OBCA_LIB (file name inside
the package) -- written for libobca_mpc_x0.so (tools/build_x0_variant.sh: the EXPERIMENTAL x0 start of DESIGN.md section 9).
    python tools/gpu_x0_variant.py save /tmp/base.npz                       # the library in the tree, answers kept
    OBCA_LIB=libobca_mpc_x0.so python tools/gpu_x0_variant.py cmp /tmp/base.npz
Prints one JSON line."""
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from vehicle_motion_planning_with_obstacles_avoidance_using_mpc_amd import _lib, scenarios as sc   # noqa: E402
if os.environ.get("OBCA_LIB"):
    _lib.LIB_PATH = os.path.join(_lib.HERE, os.environ["OBCA_LIB"])
import bench                                                                                      # noqa: E402
from vehicle_motion_planning_with_obstacles_avoidance_using_mpc_amd.solver import BatchSolver, SolverParams   # noqa: E402


def main():
    mode, path = sys.argv[1], sys.argv[2]
    torch.cuda.set_device(0)
    B, N = 8192, 5
    b = sc.make_batch(B, N)
    s = BatchSolver(N, b["m"], max_batch=B)
    dv = {k: torch.as_tensor(b[k], device="cuda") for k in ("variant", "x0", "u0", "xref", "A", "b", "Ts", "term")}
    go = lambda: s.solve(dv["variant"], dv["x0"], dv["u0"], dv["xref"], dv["A"], dv["b"], dv["Ts"], dv["term"], SolverParams())
    o = go()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        o = go()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 3
    st, x, ts = o.status.cpu().numpy(), o.xopt.cpu().numpy(), o.ts_opt.cpu().numpy()
    ok = np.isin(st, (0, 1))
    res = {"library": os.path.basename(_lib.LIB_PATH),
           "headline": {"value": float(ok.sum() / dt), "ms_per_launch": dt * 1e3, "success_rate": float(ok.mean()), "mean_ipm_iters": float(o.iters.float().mean())}}
    if mode == "save":
        np.savez(path, st=st, x=x, ts=ts)
    else:
        z = np.load(path)
        both = ok & np.isin(z["st"], (0, 1))
        same = both & (np.abs(ts - z["ts"]) <= 1e-6 * np.maximum(1.0, np.abs(z["ts"]))) & (np.abs(x - z["x"]).reshape(B, -1).max(1) <= 1e-5)
        res["headline"]["same_optimum_as_the_library_in_the_tree"] = int(same.sum())
        res["headline"]["of_instances_both_solved"] = int(both.sum())
    c3 = bench.config_c3(B)
    res["config_c3"] = {k: {q: v[q] for q in ("value", "ms_per_launch", "success_rate", "mean_ipm_iters")} for k, v in c3.items() if isinstance(v, dict)}
    c5 = bench.closed_loop_c5(4096)
    res["closed_loop"] = {k: c5[k] for k in ("value", "seconds", "converged_steps", "attempted_steps", "rollouts_to_step_cap", "rollouts_stopped_infeasible", "mean_ipm_iters")}
    g = bench.reference_gif_leg()
    res["reference_gif_default_order"] = g["cold_start"]
    ol = bench.open_loop()
    res["open_loop"] = {k: v for k, v in ol.items() if isinstance(v, dict) and "seconds" in v}
    print(json.dumps(res))


if __name__ == "__main__":
    main()
