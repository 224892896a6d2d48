"""dev tool (round-5 review item 7), run on the GPU box: C5 on worlds first .. first + B - 1 through the product path, every stopped
rollout replayed on the host and classified (tests/independent.py), and for the SOLVER failures (a feasible point exists) a list of
further starts tried with the numpy specification of the product's method: which of them ends feasible.  first = 0: the worlds the
round-5 bench line names (124 / 652 / 667); first = 4096: worlds no rule of this build was looked at on.
    python tools/c5_heldout_failures.py <first> [B]  ->  one JSON line"""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, ".")


def candidates_job(job):
    i, n_dyn = job
    from oracle import ipm_dense as ipm
    from tools.c5_failures_study import last_call, problem_of
    cl, s = last_call(i, n_dyn)
    c = s.calls[-1]
    p = problem_of(c)
    out = dict(world=i, step=cl.k, variant=c["variant"], status=c["status"], solved_by=[])
    cands = [("dodge right mu 0.1", ipm.dodge_start(p, -1.0), 0.1, 1e4), ("dodge left mu 0.1", ipm.dodge_start(p, 1.0), 0.1, 1e4),
             ("window mu 0.1", ipm.window_start(p), 0.1, 1e4), ("x0 rho 1e3", ipm.x0_start(p), 0.1, 1e3), ("window rho 1e3", ipm.window_start(p), 1.0, 1e3),
             ("x0 mu 10", ipm.x0_start(p), 10.0, 1e4), ("zeros rho 1e3", None, 0.1, 1e3)]
    for name, st, mu, rho in cands:
        r = ipm._solve_once(p, dict(mu_init=mu, max_iter=350, rho=rho), x_start=st)
        if r.status in (0, 1):
            out["solved_by"].append(name)
    return out


if __name__ == "__main__":
    import torch                                                                                      # noqa: F401
    import bench
    first = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
    B = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
    t0 = time.time()
    res = bench.closed_loop_c5(B, first=first, classify=True)
    split = res.get("stopped_infeasible_split", {})
    fails = split.get("solver_failures_world_step_variant_status", [])
    rows = []
    if fails:
        import multiprocessing as mp
        from tools.c5_heldout_failures import candidates_job as job          # (importable name for the spawned workers)
        os.environ["OMP_NUM_THREADS"] = os.environ["OPENBLAS_NUM_THREADS"] = "1"
        with mp.get_context("spawn").Pool(max(1, min(len(fails), 64, os.cpu_count() or 1))) as pool:
            rows = pool.map(job, [(int(f[0]), 2) for f in fails])
    print(json.dumps(dict(first=first, B=B, steps_per_s=res["value"], stopped=res["rollouts_stopped_infeasible"], split=split, further_starts=rows,
                          seconds=time.time() - t0)))
