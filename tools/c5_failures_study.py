"""dev tool (round-5 review item 7): the C5 rollouts that stop although a feasible point exists (profiles/r05_bench_classify_all.json:
worlds 124 / 652 / 667, obca_mpc8, status 2).  Each world replayed on the host (structured core = the device's iterates); the last
call -- the obca_mpc8 that stopped the rollout -- per SLSQP start of the classifier, per start of the ladder and per dodge side, with
the product's spec (oracle/ipm_dense.py) where it is cheap.   python tools/c5_failures_study.py [world ...]"""
import json
import sys

import numpy as np

sys.path.insert(0, ".")
from oracle import c_oracle                                                                             # noqa: E402
from oracle.obca_nlp import Problem                                                                     # noqa: E402
from tests import independent as ind, native_build                                                      # noqa: E402
from vehicle_motion_planning_with_obstacles_avoidance_using_mpc_amd import scenarios as sc              # noqa: E402
from vehicle_motion_planning_with_obstacles_avoidance_using_mpc_amd.closed_loop import closedLoop       # noqa: E402
from vehicle_motion_planning_with_obstacles_avoidance_using_mpc_amd.solver import SolverParams          # noqa: E402


def last_call(i, n_dyn=2, **solver_kw):
    s = native_build.LpiObca()
    for k, v in solver_kw.items():
        setattr(s, k, v)
    cl = closedLoop(sc.make_world_c5(i, n_dyn=n_dyn), solver=s)
    cl.N_free = cl.N_fix = 5
    cl.closed_loop_mpc4()
    return cl, s


def problem_of(c):
    sp = SolverParams()
    v = c["variant"]
    W = (sp.Q_free, sp.R_free, sp.P_free) if v == 4 else (sp.Q_fix, sp.R_fix, sp.P_fix)
    return Problem(v, c["xref"].shape[1] - 1, c["m"], c["x0"], c["u0"], c["xref"], c["A"], c["b"], c["Ts"], W[0], W[1][0], W[1][1], W[2],
                   sp.xL, sp.xU, sp.uL, sp.uU, sp.ego, sp.dmin, term=c["term"] if v == 6 else None)


def study(i):
    cl, s = last_call(i)
    c = s.calls[-1]
    out = dict(world=i, steps=cl.k, variant=c["variant"], status=c["status"], iters=c["iters"], x0=np.round(c["x0"], 4).tolist(), Ts=c["Ts"],
               n_obs=len(c["m"]))
    if c["status"] in (0, 1):
        return out
    p = problem_of(c)
    out["slsqp"] = []
    for kind in ("window", "line", "right 1.5", "left 1.5", "right 3", "left 3"):
        r = ind.classify((p, np.zeros(p.n), (i,), (kind,)))
        out["slsqp"].append(dict(start=kind, viol=float(r["viol"]), f=float(r["f"]), nit=int(r["nit"])))
    arrs = (c["variant"], p.N, c["m"], c["x0"][None], c["u0"][None], c["xref"][None], c["A"][None], c["b"][None], [c["Ts"]], c["term"][None])
    out["ladder"] = []
    for order in ("window", "x0", "zeros"):
        for dodge in (False,):
            o = native_build.lpi_solve(*arrs, c_oracle.default_params(start_order=order, single_start=1, dodge=dodge))
            out["ladder"].append(dict(start=order, status=int(o["status"][0]), iters=int(o["iters"][0]), f=float(o["info"][0, 0]), elastic=float(o["info"][0, 1])))
    o = native_build.lpi_solve(*arrs, c_oracle.default_params())
    out["default"] = dict(status=int(o["status"][0]), iters=int(o["iters"][0]), elastic=float(o["info"][0, 1]))
    return out


if __name__ == "__main__":
    worlds = [int(a) for a in sys.argv[1:]] or [124, 652, 667]
    for i in worlds:
        print(json.dumps(study(i)))
