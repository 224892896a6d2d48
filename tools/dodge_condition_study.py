"""dev tool (round-5 review item 5: 'run the dodge passes only when the first order's best violation sits on a distance row -- the
condition that defines the symmetric stationary point'): on C5 worlds replayed on the host, every fixed-time call on which the order's
starts all fail and the rung is eligible: on which KIND of row the held answer's largest violation sits, and whether the rung then
finds a feasible point.   python tools/dodge_condition_study.py <first> <count> [procs]"""
import collections
import json
import sys

import numpy as np

sys.path.insert(0, ".")


def world_job(i):
    from oracle import c_oracle
    from oracle.obca_nlp import Problem
    from tests import native_build
    from vehicle_motion_planning_with_obstacles_avoidance_using_mpc_amd import scenarios as sc
    from vehicle_motion_planning_with_obstacles_avoidance_using_mpc_amd.closed_loop import closedLoop
    from vehicle_motion_planning_with_obstacles_avoidance_using_mpc_amd.solver import SolverParams
    sp = SolverParams()
    s = native_build.LpiObca()
    cl = closedLoop(sc.make_world_c5(i, n_dyn=2), solver=s)
    cl.N_free = cl.N_fix = 5
    cl.closed_loop_mpc4()
    rows = []
    for q in s.calls:
        v = q["variant"]
        if v == 4:
            continue
        single = 1 if v == 6 else 0
        arrs = (v, 5, q["m"], q["x0"][None], q["u0"][None], q["xref"][None], q["A"][None], q["b"][None], [q["Ts"]], q["term"][None])
        o0 = native_build.lpi_solve(*arrs, c_oracle.default_params(single_start=single, dodge=False), cert=True)
        if o0["status"][0] in (0, 1) or o0["iters"][0] == 0:        # solved by the order, or screened out
            continue
        o1 = native_build.lpi_solve(*arrs, c_oracle.default_params(single_start=single))
        if o1["iters"][0] == o0["iters"][0]:                          # rung not eligible (obca_mpc6 without room)
            continue
        p = Problem(v, 5, q["m"], q["x0"], q["u0"], q["xref"], q["A"], q["b"], q["Ts"], sp.Q_fix, sp.R_fix[0], sp.R_fix[1], sp.P_fix,
                    sp.xL, sp.xU, sp.uL, sp.uU, sp.ego, sp.dmin, term=q["term"] if v == 6 else None)
        z = o0["z"][0][:p.n]
        c, d = p.eq(z), p.ineq(z)
        lb, ub = p.ineq_bounds()
        viol = collections.defaultdict(float)
        for (kind, *_), val in zip(p.eq_layout(), c):
            viol[kind] = max(viol[kind], abs(val))
        for (kind, *_), val, lo, up in zip(p.ineq_layout(), d, lb, ub):
            viol[kind] = max(viol[kind], lo - val, val - up)
        top = max(viol, key=viol.get)
        rows.append(dict(world=i, variant=v, status0=int(o0["status"][0]), top=top, top_viol=float(viol[top]), rung_ok=bool(o1["status"][0] in (0, 1)),
                         rung_iters=int(o1["iters"][0] - o0["iters"][0])))
    return rows


if __name__ == "__main__":
    first, count = int(sys.argv[1]), int(sys.argv[2])
    procs = int(sys.argv[3]) if len(sys.argv) > 3 else 8
    import multiprocessing as mp
    with mp.get_context("spawn").Pool(procs) as pool:
        res = [r for rows in pool.map(world_job, range(first, first + count)) for r in rows]
    tab = collections.Counter((r["variant"], r["status0"], r["top"], r["rung_ok"]) for r in res)
    print(json.dumps(dict(first=first, count=count, calls=len(res),
                          table=[dict(variant=k[0], status_of_the_order=k[1], largest_violation_on=k[2], rung_found_a_plan=k[3], calls=v,
                                      rung_iterations=int(sum(r["rung_iters"] for r in res if (r["variant"], r["status0"], r["top"], r["rung_ok"]) == k)))
                                 for k, v in sorted(tab.items(), key=lambda kv: -kv[1])])))
