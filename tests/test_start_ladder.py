"""The start ladder (oracle/ipm_dense.py:solve, include/obca_mpc.h: start_order / single_start / patience / retry_iter) on the
CPU: numpy specification, C oracle and the structured core (csrc/obca_lpi_core.h, the code the lane kernel runs) apply the same
rule and reach the same points.  Anchor: SURVEY Appendix C's "mpc6 witness" (demo1, moving box advanced 8 steps) -- the first
start ends at an infeasible stationary point, a feasible plan with f = 0.029735 exists; the survey's criterion is feas = True
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


@pytest.mark.parametrize("order", ["default", "x0", "window", "zeros"])
def test_mpc6_witness_is_met_by_all_three_cpu_implementations_in_every_order(nlp_golden, order):
    case = [c for c in nlp_golden if c["name"] == "demo1_dyn_mpc6"][0]
    p = build(case)
    r = ipm_dense.solve(p, dict(start_order=order))
    assert r.feas and r.f <= F_WITNESS + 1e-6
    assert r.restarted == (order not in ("window", "default"))       # x0 and zeros end under the box; the window start (the default for obca_mpc6) finds the plan
    cert = ipm_dense.kkt_certificate(p, r)
    assert cert["primal"] < 1e-8 and cert["stationarity"] < 1e-6 and cert["complementarity"] < 1e-6
    assert r.xopt[1].max() > 7.9                                   # passes ABOVE the moving box (the witness's class)
    args = _packed(case, start_order=order)
    c = c_oracle.solve_batch(*args)
    g = native_build.lpi_solve(*args)
    for o in (c, g):
        assert o["status"][0] == 0 and o["info"][0, 0] <= F_WITNESS + 1e-6
        np.testing.assert_allclose(o["xopt"][0], r.xopt, rtol=0, atol=1e-8)
        np.testing.assert_allclose(o["uopt"][0], r.uopt, rtol=0, atol=1e-8)


@pytest.mark.parametrize("order", ["x0", "zeros"])
def test_single_start_ends_where_the_first_start_ends(nlp_golden, order):
    """obca_params.single_start = 1: the first start of the order alone -- on the witness it ends at the infeasible stationary
    point (status 2), which is what a driver with its own fallback (obca_mpc8 after obca_mpc6) asks for"""
    case = [c for c in nlp_golden if c["name"] == "demo1_dyn_mpc6"][0]
    r = ipm_dense.solve(build(case), dict(start_order=order, single_start=True, dodge=False))
    assert r.status == ipm_dense.STATUS_INFEASIBLE and not r.restarted
    args = _packed(case, start_order=order, single_start=1, dodge=False)
    assert c_oracle.solve_batch(*args)["status"][0] == 2
    assert native_build.lpi_solve(*args)["status"][0] == 2
    # ... and with the ladder's last rung (the default) the same call finds a plan around the box: the dodge passes follow a
    # single start as well -- what the reference's demo11 run needs from the closed loop's obca_mpc6
    args = _packed(case, start_order=order, single_start=1)
    r = ipm_dense.solve(build(case), dict(start_order=order, single_start=True))
    c, g = c_oracle.solve_batch(*args), native_build.lpi_solve(*args)
    assert r.status == 0 and r.dodged and c["status"][0] == 0 and g["status"][0] == 0
    np.testing.assert_allclose(c["xopt"][0], r.xopt, rtol=0, atol=1e-8)
    np.testing.assert_allclose(g["xopt"][0], r.xopt, rtol=0, atol=1e-8)


@pytest.mark.parametrize("order", ["default", "x0", "window", "zeros"])
def test_a_genuinely_infeasible_problem_stays_infeasible(nlp_golden, order):
    """demo1 at N = 5 (the terminal pose collides): all three starts run, each with its penalty escalation,
    feas stays False -- and the whole sequence is the same in the three implementations, iterate for iterate"""
    case = [c for c in nlp_golden if c["name"] == "demo1_N5_mpc4_step0"][0]
    r = ipm_dense.solve(build(case), dict(start_order=order))
    assert r.status == ipm_dense.STATUS_INFEASIBLE and r.starts_used == 3 and r.elastic > 1e-3
    args = _packed(case, start_order=order)
    c, g = c_oracle.solve_batch(*args), native_build.lpi_solve(*args)
    assert c["status"][0] == 2 and g["status"][0] == 2
    assert c["iters"][0] == r.iters == g["iters"][0]
    singles = [native_build.lpi_solve(*_packed(case, start_order=o, single_start=1))["iters"][0] for o in ("x0", "window", "zeros")]
    assert sum(singles) == r.iters                                 # the ladder is the three starts one after the other


def test_iteration_limits_of_the_ladder(nlp_golden):
    """include/obca_mpc.h: max_iter_* bounds each pass; while further starts remain the first start's passes stop after
    `patience`, the later ones after `retry_iter`; with single_start only max_iter_* applies (the advisor's round-3 finding:
    the caps must not override the caller's max_iter silently).  demo1_dyn_mpc6: the x0 start needs 59 iterations to reach its
    infeasible stationary point, the window start 49 to the plan."""
    case = [c for c in nlp_golden if c["name"] == "demo1_dyn_mpc6"][0]
    for engine in (c_oracle.solve_batch, native_build.lpi_solve):
        x0f = dict(start_order="x0", dodge=False)                                # (x0 first: the default order starts obca_mpc6 at the window; the rung after the order is switched off: its passes would add their own retry_iter iterations)
        it0 = engine(*_packed(case, single_start=1, **x0f))["iters"][0]        # the x0 start alone, to its end
        o = engine(*_packed(case, patience=20, **x0f))                           # first start abandoned after 20 iterations
        assert o["status"][0] == 0 and 20 < o["iters"][0] < it0 + 49
        o = engine(*_packed(case, patience=20, retry_iter=10, **x0f))            # ... and the later starts after 10 each
        assert o["status"][0] == -1 and o["iters"][0] <= 20 + 10 + 10 + 3
        o = engine(*_packed(case, single_start=1, patience=20, max_iter_fixed=2000, **x0f))
        assert o["status"][0] == 2 and o["iters"][0] == it0                      # single start: patience does not apply
        o = engine(*_packed(case, single_start=1, max_iter_fixed=30, **x0f))
        assert o["status"][0] == -1 and o["iters"][0] <= 31                      # ... max_iter does


def test_patience_above_its_default_is_honoured():
    """A first start that crawls: step 51 of the reference's demo9 run replayed with the literal zero start first (tests/
    reference_gif.py) -- the zero start converges to an infeasible stationary point, its repetition with the raised penalty never
    converges, the window start solves the problem in 28 iterations.  The default patience (500 + 10 N = 550) hands over after 550
    iterations of that pass; a caller who sets patience = 2000 gets 2000."""
    from tests import reference_gif
    s = native_build.LpiObca()
    s.start_order = "zeros"
    reference_gif.replay(s, 51)
    c = s.calls[50]
    assert c["status"] == 0 and 550 < c["iters"] < 550 + 350
    arrs = (4, 5, c["m"], c["x0"][None], c["u0"][None], c["xref"][None], c["A"][None], c["b"][None], [c["Ts"]], c["term"][None])
    kw = dict(xL=[0, 0], xU=[40, 60], Qf=0.5 * np.eye(3), Pf=0.5 * np.eye(3), start_order="zeros")
    base = native_build.lpi_solve(*arrs, c_oracle.default_params(**kw))
    assert base["status"][0] == 0 and base["iters"][0] == c["iters"]
    long = native_build.lpi_solve(*arrs, c_oracle.default_params(patience=2000, **kw))
    assert long["status"][0] == 0 and long["iters"][0] == c["iters"] - 550 + 2000
    np.testing.assert_allclose(long["xopt"], base["xopt"], atol=1e-9)
    capped = native_build.lpi_solve(*arrs, c_oracle.default_params(patience=2000, max_iter_free=300, **kw))
    assert capped["iters"][0] < c["iters"]                                      # never above max_iter_free


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


def test_x0_start_point(nlp_golden):
    p = build([c for c in nlp_golden if c["name"] == "demo9_N5_mpc4_step0"][0])
    z = ipm_dense.x0_start(p)
    xs, us = p.unpack_xu(z)
    assert np.array_equal(xs, np.repeat(np.asarray(p.x0, float)[:, None], p.N + 1, 1)) and not us.any() and z[p.iT()] == 1.0
    z0 = p.start_point()                                           # the reference's literal start: zeros, Topt = 1 (src/obca.py:856)
    diff = np.flatnonzero(z != z0)
    assert set(diff) <= {p.ip(k) + j for k in range(p.N + 1) for j in range(3)}


def test_c3_gated_batch_with_the_ladder_structured_core_against_dense_oracle():
    """config 3's fixed-time half at N = 8 (the dense oracle's reach): 32 instances through the structured core and the
    dense C oracle.  Where the first start fails both go on to the next; verdicts agree, every answer of the core is certified
    on the reference-pinned model, and the ladder is what lifts the share of converged instances."""
    N, B = 8, 32
    b = sc.make_batch_c3(B, N, gated=True)
    args = (b["variant"], N, b["m"], b["x0"], b["u0"], b["xref"], b["A"], b["b"], b["Ts"], b["term"])
    cold = native_build.lpi_solve(*args, params=c_oracle.default_params(single_start=1, start_order="x0", dodge=False))     # the x0 start alone
    got = native_build.lpi_solve(*args, cert=True)                     # default: the window first for obca_mpc6, then x0, then zeros
    ref = c_oracle.solve_batch(*args, threads=8)
    ok_cold, ok = np.isin(cold["status"], (0, 1)), np.isin(got["status"], (0, 1))
    assert (ok | ~ok_cold).all()                                   # nothing that converged cold is lost
    assert ok.sum() > ok_cold.sum() and ok.mean() >= 0.9
    assert (np.isin(ref["status"], (0, 1)) != ok).sum() <= 1       # long non-convex runs: one verdict may flip with roundoff
    for i in np.flatnonzero(ok):
        c = kkt_check.certificate(kkt_check.problem_of(b, i, N), got["z"][i], got["y"][i])
        for k in ("stationarity", "primal", "dual_sign", "complementarity"):
            assert c[k] <= 1e-6, (i, k, c)


def test_out_of_range_start_fields_are_rejected_by_every_cpu_implementation(nlp_golden):
    """include/obca_mpc.h: start_order outside OBCA_START_*, single_start outside 0 / 1 -> OBCA_E_INVAL.  The dense C oracle and the
    host build of the core answer the same inputs with an error too (the oracle used to map them to defaults silently), and the numpy
    spec raises: a parity test with a mistyped option cannot pass on one side only."""
    case = [c for c in nlp_golden if c["name"] == "demo1_dyn_mpc6"][0]
    for bad in (dict(start_order=4), dict(start_order=-1), dict(single_start=2)):
        with pytest.raises(ValueError):
            c_oracle.solve_batch(*_packed(case, **bad))
        with pytest.raises(AssertionError):
            native_build.lpi_solve(*_packed(case, **bad))
    with pytest.raises(KeyError):
        ipm_dense.solve(build(case), dict(start_order=7))


@pytest.mark.parametrize("order,first,second", [("default", "window", "x0"), ("x0", "x0", "window"), ("zeros", "zeros", "window")])
def test_an_exhausted_ladder_returns_the_most_informative_pass(nlp_golden, order, first, second):
    """Which pass's status / iterate an exhausted ladder returns (csrc/obca_device.h: OBCA_LADDER_REPLACES; until round 6: the last
    pass's, whatever it was).  demo1 at N = 5 has no feasible point and every start converges there with elastic
    variables left, so:
      * all passes converge: the answer is the FIRST start's (its last penalty level) -- the same words as that start run alone;
      * the first start cut off after 20 iterations (MAXITER), the second converges: the second's answer, status 2 -- 'no feasible
        point' says more than 'ran out of iterations';
      * the first converges, the later ones are cut off: the first's answer stays (before: status -1, the last pass's iterate).
    Same rule, same words in the numpy specification, the C oracle and the structured core."""
    case = [c for c in nlp_golden if c["name"] == "demo1_N5_mpc4_step0"][0]
    p = build(case)
    engines = (c_oracle.solve_batch, native_build.lpi_solve)
    alone = {k: [e(*_packed(case, start_order=k, single_start=1)) for e in engines] for k in (first, second)}
    spec_alone = {k: ipm_dense.solve(p, dict(start_order=k, single_start=True)) for k in (first, second)}
    assert all(o["status"][0] == 2 for k in alone for o in alone[k])

    def same(o, ref, status=2):
        assert o["status"][0] == status
        assert np.array_equal(o["xopt"], ref["xopt"]) and np.array_equal(o["uopt"], ref["uopt"]) and np.array_equal(o["ts_opt"], ref["ts_opt"])
        assert np.array_equal(o["info"][0, :3], ref["info"][0, :3])

    for i, e in enumerate(engines):
        full = e(*_packed(case, start_order=order))
        same(full, alone[first][i])
        assert full["iters"][0] > alone[first][i]["iters"][0]                    # ... the count is the whole sequence's
        cut = e(*_packed(case, start_order=order, patience=20))
        assert cut["iters"][0] > 20
        # (the second start's passes run under retry_iter = 350 here, under max_iter alone: it converges well below either)
        same(cut, alone[second][i])
        late = e(*_packed(case, start_order=order, retry_iter=10))
        same(late, alone[first][i])
        assert late["iters"][0] <= alone[first][i]["iters"][0] + 2 * 3 * 11
    r = ipm_dense.solve(p, dict(start_order=order))
    assert r.status == 2 and np.array_equal(r.xopt, spec_alone[first].xopt) and r.start_index == 0
    r = ipm_dense.solve(p, dict(start_order=order, patience=20))
    assert r.status == 2 and np.array_equal(r.xopt, spec_alone[second].xopt) and r.start_index == 1
    r = ipm_dense.solve(p, dict(start_order=order, retry_iter=10))
    assert r.status == 2 and np.array_equal(r.xopt, spec_alone[first].xopt) and r.start_index == 0


def test_second_level_of_the_dodge_rung_on_c5_world_667():
    """csrc/obca_device.h: OBCA_DODGE_LEVEL2_MU (round 6; obca_mpc8 only).  C5 world 667, step 20: obca_mpc6 is screened out, obca_mpc8 ends at
    an infeasible stationary point (one mu >= 0 row short by 8 mm) from x0, window, zeros and both dodge starts at mu = 1 -- the rollout
    stopped there although SLSQP finds a feasible plan from the window moved to the right (profiles/r05_bench_classify_all.json named it).
    The same dodge start at IPOPT's own mu_init 0.1 ends on that plan (f = 0.029957, SLSQP's 0.029955).  Numpy specification, dense C oracle
    and structured core: the same answer; obca_mpc6 does not get the second level (its failure has an answer: obca_mpc8)."""
    from oracle.obca_nlp import Problem
    from vehicle_motion_planning_with_obstacles_avoidance_using_mpc_amd.closed_loop import closedLoop
    from vehicle_motion_planning_with_obstacles_avoidance_using_mpc_amd.solver import SolverParams
    s = native_build.LpiObca()
    cl = closedLoop(sc.make_world_c5(667, n_dyn=2), solver=s)
    cl.N_free = cl.N_fix = 5
    cl.closed_loop_mpc4()
    assert cl.k > 20                                                         # (round 5: stopped at step 20)
    # the obca_mpc8 call of step 20: the first one whose order and first dodge level all fail
    sp = SolverParams()
    hit = None
    for q in s.calls:
        if q["variant"] != 8 or q["status"] not in (0, 1):
            continue
        arrs = (8, 5, q["m"], q["x0"][None], q["u0"][None], q["xref"][None], q["A"][None], q["b"][None], [q["Ts"]], q["term"][None])
        if native_build.lpi_solve(*arrs, c_oracle.default_params(dodge=False))["status"][0] == 2 and abs(q["x0"][0] - 17.3473) < 1e-3:
            hit = (q, arrs)
    assert hit is not None
    q, arrs = hit
    p = Problem(8, 5, q["m"], q["x0"], q["u0"], q["xref"], q["A"], q["b"], q["Ts"], sp.Q_fix, sp.R_fix[0], sp.R_fix[1], sp.P_fix,
                sp.xL, sp.xU, sp.uL, sp.uU, sp.ego, sp.dmin)
    for side in (-1.0, 1.0):                                                 # first level: neither side
        r = ipm_dense._solve_once(p, dict(mu_init=ipm_dense.RESTART_MU, max_iter=350), x_start=ipm_dense.dodge_start(p, side))
        assert r.status == ipm_dense.STATUS_INFEASIBLE and 5e-3 < r.elastic < 1e-2
    r = ipm_dense.solve(p)
    assert r.status == 0 and r.dodged and r.f == pytest.approx(0.029956, abs=3e-6)
    c, g = c_oracle.solve_batch(*arrs, c_oracle.default_params()), native_build.lpi_solve(*arrs, c_oracle.default_params())
    assert c["iters"][0] == r.iters                        # dense C oracle: the specification's iterates
    for o, tol in ((c, 1e-8), (g, 1e-6)):                  # (structured core: other linear algebra, the failing passes take a few iterations more)
        assert o["status"][0] == 0
        np.testing.assert_allclose(o["xopt"][0], r.xopt, rtol=0, atol=tol)
    # obca_mpc6 on the same inputs (a terminal set within reach, so that the rung is tried): two dodge passes, not four
    p6 = (6,) + arrs[1:9] + (np.array([[q["x0"][0] + 2.0, 1.0, 9.0]]),)
    o6a = native_build.lpi_solve(*p6, c_oracle.default_params())
    o6b = native_build.lpi_solve(*p6, c_oracle.default_params(dodge=False))
    o8b = native_build.lpi_solve(*arrs, c_oracle.default_params(dodge=False))
    if o6a["status"][0] == 2:
        assert o6a["iters"][0] - o6b["iters"][0] < g["iters"][0] - o8b["iters"][0]
