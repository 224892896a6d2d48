"""The restart phase (oracle/ipm_dense.py:solve, include/obca_mpc.h: obca_params.restart) on the CPU: numpy specification,
C oracle and the structured core (csrc/obca_lpi_core.h, the code the lane kernel runs) apply the same rule and reach the
same points.  Anchor: SURVEY Appendix C's "mpc6 witness" (demo1, moving box advanced 8 steps) -- the reference's cold start
ends at an infeasible stationary point, a feasible plan with f = 0.029735 exists; the survey's criterion is feas = True
with f <= 0.02974."""
import numpy as np
import pytest

from oracle import c_oracle, ipm_dense
from tests import kkt_check, native_build
from tests.test_oracle_nlp import build
from vehicle_motion_planning_with_obstacles_avoidance_using_mpc_amd import scenarios as sc
from vehicle_motion_planning_with_obstacles_avoidance_using_mpc_amd.solver import pack_reference_call

F_WITNESS = 0.02974


def _packed(case, **extra):
    a = case["inputs"]
    m, x0, u0, xr, A, b, ts, term = pack_reference_call(case["variant"], a["Ts"], a["N"], a["x0"], a["xref"], a["nObs"], a["vObs"],
                                                        a["AObs"], a["bObs"], a["u0"], a.get("terminal_set"))
    R = [np.array(r) for r in a["R"]]
    kw = dict(xL=a["xL"][:2], xU=a["xU"][:2], uL=a["uL"], uU=a["uU"], ego=a["ego"], dmin=a["dmin"], **extra)
    kw.update(dict(Qf=a["Q"], Pf=a["P"], R1f=R[0], R2f=R[1]) if case["variant"] == 4 else dict(Qx=a["Q"], Px=a["P"], R1x=R[0], R2x=R[1]))
    return (case["variant"], a["N"], m, x0[None], u0[None], xr[None], A[None], b[None], [ts], term[None], c_oracle.default_params(**kw))


def test_mpc6_witness_is_met_by_all_three_cpu_implementations(nlp_golden):
    case = [c for c in nlp_golden if c["name"] == "demo1_dyn_mpc6"][0]
    p = build(case)
    r = ipm_dense.solve(p)
    assert r.feas and r.restarted and r.f <= F_WITNESS + 1e-6
    cert = ipm_dense.kkt_certificate(p, r)
    assert cert["primal"] < 1e-8 and cert["stationarity"] < 1e-6 and cert["complementarity"] < 1e-6
    assert r.xopt[1].max() > 7.9                                   # passes ABOVE the moving box (the witness's class)
    args = _packed(case)
    c = c_oracle.solve_batch(*args)
    g = native_build.lpi_solve(*args)
    for o in (c, g):
        assert o["status"][0] == 0 and o["info"][0, 0] <= F_WITNESS + 1e-6
        np.testing.assert_allclose(o["xopt"][0], r.xopt, rtol=0, atol=1e-8)
        np.testing.assert_allclose(o["uopt"][0], r.uopt, rtol=0, atol=1e-8)


def test_without_the_restart_phase_the_cold_start_ends_infeasible(nlp_golden):
    """the switch (obca_params.restart < 0; `no_restart` in the numpy specification): the first pass alone, status 2"""
    case = [c for c in nlp_golden if c["name"] == "demo1_dyn_mpc6"][0]
    r = ipm_dense.solve(build(case), dict(no_restart=True))
    assert r.status == ipm_dense.STATUS_INFEASIBLE and not getattr(r, "restarted", False)
    args = _packed(case, restart=-1)
    assert c_oracle.solve_batch(*args)["status"][0] == 2
    assert native_build.lpi_solve(*args)["status"][0] == 2


def test_a_genuinely_infeasible_problem_stays_infeasible(nlp_golden):
    """demo1 at N = 5 (the terminal pose collides): escalation and restart run, feas stays False"""
    case = [c for c in nlp_golden if c["name"] == "demo1_N5_mpc4_step0"][0]
    r = ipm_dense.solve(build(case))
    assert r.status == ipm_dense.STATUS_INFEASIBLE and r.restarted and r.elastic > 1e-3
    args = _packed(case)
    c, g = c_oracle.solve_batch(*args), native_build.lpi_solve(*args)
    assert c["status"][0] == 2 and g["status"][0] == 2
    assert c["iters"][0] == r.iters == g["iters"][0]               # the three passes, iterate for iterate


@pytest.mark.parametrize("name", ["demo1_dyn_mpc6", "demo9_N5_mpc4_step0", "slanted_asym_mpc4"])
def test_window_start_point(nlp_golden, name):
    p = build([c for c in nlp_golden if c["name"] == name][0])
    z = ipm_dense.window_start(p)
    xs, us = p.unpack_xu(z)
    assert np.array_equal(xs[:, 0], p.x0) and np.array_equal(xs[:, 1:], p.xref[:, 1:])
    assert (us[0] >= p.uL[0]).all() and (us[0] <= p.uU[0]).all() and (us[1] >= p.uL[1]).all() and (us[1] <= p.uU[1]).all()
    if p.variant == 4:
        T = z[p.iT()]
        assert 1.0 <= T <= max(1.0, p.Tmax)
        seg = np.hypot(*np.diff(xs[:2], axis=1))
        assert T == pytest.approx(min(max(1.0, seg.sum() / (p.N * 0.9 * p.uU[0] * p.Ts)), max(1.0, p.Tmax)))
    lam_mu = np.ones(p.n, bool)
    for k in range(p.N + 1):
        lam_mu[p.ip(k):p.ip(k) + (5 if k < p.N else 3)] = False
    if p.variant == 4:
        lam_mu[p.iT()] = False
    assert not z[lam_mu].any()


def test_c3_gated_batch_with_restarts_structured_core_against_dense_oracle():
    """config 3's fixed-time half at N = 8 (the dense oracle's reach): 32 instances through the structured core and the
    dense C oracle.  Where the first pass fails both restart; verdicts agree, every answer of the core is certified on
    the reference-pinned model, and the restart phase is what lifts the share of converged instances."""
    N, B = 8, 32
    b = sc.make_batch_c3(B, N, gated=True)
    args = (b["variant"], N, b["m"], b["x0"], b["u0"], b["xref"], b["A"], b["b"], b["Ts"], b["term"])
    cold = native_build.lpi_solve(*args, params=c_oracle.default_params(restart=-1))
    got = native_build.lpi_solve(*args, cert=True)
    ref = c_oracle.solve_batch(*args, threads=8)
    ok_cold, ok = np.isin(cold["status"], (0, 1)), np.isin(got["status"], (0, 1))
    assert (ok | ~ok_cold).all()                                   # nothing that converged cold is lost
    assert ok.sum() > ok_cold.sum() and ok.mean() >= 0.9
    assert (np.isin(ref["status"], (0, 1)) != ok).sum() <= 1       # long non-convex runs: one verdict may flip with roundoff
    for i in np.flatnonzero(ok):
        c = kkt_check.certificate(kkt_check.problem_of(b, i, N), got["z"][i], got["y"][i])
        for k in ("stationarity", "primal", "dual_sign", "complementarity"):
            assert c[k] <= 1e-6, (i, k, c)
