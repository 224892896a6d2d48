"""Answers of the product against an INDEPENDENT solver on the reference-pinned model (tests/independent.py: SciPy SLSQP,
analytic Jacobians) -- the strongest stand-in for the reference's `opti.solve()` (src/obca.py:1056) this
image allows.  Two questions per instance:
  (a) started AT the answer, can SLSQP find a feasible point with a lower objective?  It must not (beyond 1e-6 relative plus
      the duality gap an interior-point answer carries by construction).
  (b) started from the reference window and two perturbations of it, does it reach the same optimum?  On the free-time
      problems it does from every start; the fixed-time problems are non-convex enough that some starts end elsewhere --
      the shares are reported (bench.py prints them as `independent_solver`) and bounded here.
CPU part: the structured core on the host (same code as the lane kernel).  `-m gpu` part: the HIP kernels."""
import os

import numpy as np
import pytest

from tests import independent as ind, kkt_check, native_build
from tests.test_oracle_nlp import build
from vehicle_motion_planning_with_obstacles_avoidance_using_mpc_amd import scenarios as sc

PROCS = max(1, min(128, (os.cpu_count() or 1)))


def _check(jobs_polish, jobs_starts, what, max_better=0.0, min_same=0.0):
    pol = ind.pool_map(ind.polish, jobs_polish, PROCS)
    st = ind.pool_map(ind.from_starts, jobs_starts, PROCS) if jobs_starts else []
    summ = ind.summarise(pol, st)
    print(what, summ)
    bad = [(i, r) for i, r in enumerate(pol) if r["improved"]]
    assert not bad, (what, bad[:3])
    if st:
        assert summ["better_optimum"] <= max_better, (what, summ)
        assert summ["same_optimum"] >= min_same, (what, summ)
    return summ


def test_golden_answers_of_the_numpy_specification(nlp_golden):
    from oracle import ipm_dense
    jobs = []
    for case in nlp_golden:
        p = build(case)
        r = ipm_dense.solve(p)
        if r.feas:
            d = p.ineq(r.x)
            lb, ub = p.ineq_bounds()
            slack = np.minimum(np.where(np.isfinite(lb), d - lb, np.inf), np.where(np.isfinite(ub), ub - d, np.inf))
            yd = r.ye[ipm_dense.split_rows(p)[1].size:]
            jobs.append((p, r.x, float(np.sum(np.abs(yd) * np.maximum(slack, 0.0)))))
    assert len(jobs) == 8
    _check(jobs, [], "golden scenarios (numpy specification)")


def test_c2_answers_of_the_structured_core():
    B, N = 16, 5
    b = sc.make_batch(B, N)
    o = native_build.lpi_solve(b["variant"], N, b["m"], b["x0"], b["u0"], b["xref"], b["A"], b["b"], b["Ts"], b["term"], cert=True)
    ps = [kkt_check.problem_of(b, i, N) for i in range(B)]
    _check([(ps[i], o["z"][i], o["y"][i]) for i in range(B)], [(ps[i], o["z"][i], i) for i in range(B)], "C2, structured core",
           max_better=0.0, min_same=0.95)


def _gpu_solve(b, N):
    import torch
    from vehicle_motion_planning_with_obstacles_avoidance_using_mpc_amd.solver import BatchSolver, SolverParams
    B = len(b["variant"])
    s = BatchSolver(N, b["m"], max_batch=B)
    s.enable_certificates()
    out = s.solve(b["variant"], b["x0"], b["u0"], b["xref"], b["A"], b["b"], b["Ts"], b["term"], SolverParams())
    torch.cuda.synchronize()
    r = dict(st=out.status.cpu().numpy(), z=s.cert_z[:B].cpu().numpy(), y=s.cert_y[:B].cpu().numpy())
    s.close()
    return r


@pytest.mark.gpu
def test_gpu_answers_against_the_independent_solver():
    """64 C2 instances (both questions), 16 C3 free-time instances at N = 20 (both), 16 C3 gated instances at N = 20 (a) and
    16 at N = 8 (both: one SLSQP run on the 817 unknowns of N = 20 takes minutes)"""
    for what, b, N, starts, max_better, min_same in (
            ("C2 (64)", sc.make_batch(64, 5), 5, True, 0.0, 0.95),
            ("C3 free-time, N = 20 (16)", sc.make_batch_c3(16, 20, gated=False), 20, True, 0.0, 0.75),
            ("C3 gated, N = 20 (16)", sc.make_batch_c3(16, 20, gated=True), 20, False, 0.0, 0.0),
            ("C3 gated, N = 8 (16)", sc.make_batch_c3(16, 8, gated=True), 8, True, 0.25, 0.5)):
        r = _gpu_solve(b, N)
        ok = np.flatnonzero(np.isin(r["st"], (0, 1)))
        assert len(ok) >= 0.9 * len(r["st"])
        ps = {i: kkt_check.problem_of(b, i, N) for i in ok}
        _check([(ps[i], r["z"][i], r["y"][i]) for i in ok], [(ps[i], r["z"][i], int(i)) for i in ok] if starts else [], what,
               max_better=max_better, min_same=min_same)
