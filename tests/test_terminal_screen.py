"""The closed-form terminal-set screen of obca_mpc6 (include/obca_mpc.h: terminal_screen; rule: oracle/ipm_dense.py:
terminal_set_shortfall) and the dodge rung of the ladder (dodge; oracle/ipm_dense.py:dodge_start) on the CPU: numpy spec, dense C
oracle and the structured core follow ONE rule, and the screen never answers a call the solver would have solved."""
import numpy as np
import pytest

from oracle import c_oracle, ipm_dense
from oracle.obca_nlp import Problem
from tests import native_build
from vehicle_motion_planning_with_obstacles_avoidance_using_mpc_amd import scenarios as sc
from vehicle_motion_planning_with_obstacles_avoidance_using_mpc_amd.closed_loop import closedLoop
from vehicle_motion_planning_with_obstacles_avoidance_using_mpc_amd.solver import SolverParams


def _loop_calls(world, screen=True, dodge=True):
    s = native_build.LpiObca()
    s.terminal_screen, s.dodge = screen, dodge
    cl = closedLoop(sc.make_world_c5(world, n_dyn=2), solver=s)
    cl.N_free = cl.N_fix = 5
    cl.closed_loop_mpc4()
    return s.calls, cl


def _problem(c):
    sp = SolverParams()
    return Problem(6, c["xref"].shape[1] - 1, c["m"], c["x0"], c["u0"], c["xref"], c["A"], c["b"], c["Ts"], sp.Q_fix, sp.R_fix[0], sp.R_fix[1],
                   sp.P_fix, sp.xL, sp.xU, sp.uL, sp.uU, sp.ego, sp.dmin, term=c["term"])


@pytest.mark.parametrize("world", [0, 11, 40])
def test_screened_closed_loop_is_the_unscreened_one(world):
    """C5 worlds replayed on the host core with and without the screen: every pose, input and step length of the closed loop equal
    (a failed obca_mpc6 is discarded by the driver, src/closed_loop.py:393-398), the screened calls are exactly calls that ended
    'infeasible' when they were run, and they make up a third to nine tenths of the failing obca_mpc6 calls: the terminal set is x0 + 5 m
    (src/closed_loop.py:371) and five steps of Ts_opt uU are 5.000 m, so after the first turn of the wheel it is out of reach"""
    on, cl_on = _loop_calls(world, screen=True)
    off, cl_off = _loop_calls(world, screen=False)
    assert np.array_equal(np.asarray(cl_on.x_closed), np.asarray(cl_off.x_closed)) and np.array_equal(np.asarray(cl_on.T_closed), np.asarray(cl_off.T_closed))
    assert len(on) == len(off)
    screened = failing = 0
    for a, b in zip(on, off):
        assert a["variant"] == b["variant"] and (a["status"] in (0, 1)) == (b["status"] in (0, 1))
        if a["variant"] == 6 and b["status"] not in (0, 1):
            failing += 1
            if a["iters"] == 0:
                screened += 1
                assert a["status"] == 2 and a["info"][1] > 0.0 and ipm_dense.terminal_set_shortfall(_problem(a)) == pytest.approx(a["info"][1], abs=1e-15)
        elif a["variant"] == 6:
            assert a["iters"] > 0 and ipm_dense.terminal_set_shortfall(_problem(a)) <= 0.0       # a call that succeeds is never screened
    assert failing >= 5 and screened >= 0.3 * failing, (screened, failing)      # (world 11: 9 of 22 -- the rest has 1e-9 m to spare and the box ahead)


def test_the_three_cpu_implementations_screen_alike():
    calls, _ = _loop_calls(0)
    c = [q for q in calls if q["variant"] == 6 and q["iters"] == 0][0]
    p = _problem(c)
    r = ipm_dense.solve(p, dict(single_start=True))
    assert r.status == ipm_dense.STATUS_INFEASIBLE and r.iters == 0 and r.screened and np.array_equal(r.xopt, np.repeat(p.x0[:, None], p.N + 1, 1))
    sp = SolverParams()
    kw = dict(xL=sp.xL, xU=sp.xU, uL=sp.uL, uU=sp.uU, ego=sp.ego, dmin=sp.dmin, single_start=1, Qx=sp.Q_fix, Px=sp.P_fix, R1x=sp.R_fix[0], R2x=sp.R_fix[1])
    args = (6, p.N, c["m"], c["x0"][None], c["u0"][None], c["xref"][None], c["A"][None], c["b"][None], [c["Ts"]], c["term"][None])
    for engine in (c_oracle.solve_batch, native_build.lpi_solve):
        o = engine(*args, c_oracle.default_params(**kw))
        assert o["status"][0] == 2 and o["iters"][0] == 0 and o["info"][0, 1] == r.elastic and np.array_equal(o["xopt"][0], r.xopt) and not o["uopt"].any()
        o = engine(*args, c_oracle.default_params(terminal_screen=False, dodge=False, **kw))         # switched off: the solve runs, same verdict
        assert o["status"][0] == 2 and o["iters"][0] > 10
    assert ipm_dense.solve(p, dict(single_start=True, terminal_screen=False, dodge=False)).iters > 10


def test_shortfall_margin_scales_with_the_rows_elastic_slack():
    """the margin is what elastic variables of size feas_tol can add on the rows involved: a terminal set just out of reach by less
    than that is NOT screened (the solver decides), one beyond it is"""
    calls, _ = _loop_calls(0)
    c = dict([q for q in calls if q["variant"] == 6][0])
    p = _problem(c)
    reach = p.term[0] - ipm_dense.terminal_set_shortfall(p, 0.0)          # largest reachable x_N (margin 0)
    for extra, want in ((-1e-3, False), (5e-5, False), (5e-4, True)):
        c["term"] = np.array([reach + extra, 1.0, 9.0])
        assert (ipm_dense.terminal_set_shortfall(_problem(c)) > 0.0) == want, extra


def test_dodge_start_point(nlp_golden):
    from tests.test_oracle_nlp import build
    p = build([c for c in nlp_golden if c["name"] == "demo1_dyn_mpc6"][0])
    for side in (-1.0, 1.0):
        z = ipm_dense.dodge_start(p, side)
        xs, us = p.unpack_xu(z)
        assert np.array_equal(xs[:, 0], p.x0)
        off = np.hypot(xs[0, 1:] - p.xref[0, 1:], xs[1, 1:] - p.xref[1, 1:])
        np.testing.assert_allclose(off, [min(1.0, k / 3.0) * 3.0 for k in range(1, p.N + 1)], atol=1e-12)        # ramped in over three stages
        assert (us[0] >= p.uL[0]).all() and (us[0] <= p.uU[0]).all() and (us[1] >= p.uL[1]).all() and (us[1] <= p.uU[1]).all()
        for k in range(p.N + 1):                       # one lambda per (stage, obstacle), scaled to ||A'lambda|| = 1; the rotation equalities hold
            for i in range(p.nObs):
                lam = z[p.il(k) + p.off_m[i]:p.il(k) + p.off_m[i + 1]]
                assert np.count_nonzero(lam) == 1 and np.linalg.norm(p.A[k, p.off_m[i]:p.off_m[i + 1]].T @ lam) == pytest.approx(1.0)
        assert np.max(np.abs(p.eq(z)[[i for i, r in enumerate(p.eq_layout()) if r[0] == "rot"]])) < 1e-12
