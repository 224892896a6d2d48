"""The reference's own run of its checked-in setting ``demo11`` (src/demo_setting.py:248-269) -- the fifth solver output the
reference repository holds, and the one no default of this build was chosen on.  Two records of the SAME run: the four frame titles
of report Figure 11 (cumulative ``Ts_opt`` to 0.01 s; tests/golden/reference_report_figures.json) and the closed-loop markers of
``images/OBCA_dynObs_demo11.gif`` (55 poses to ~0.15 m; tests/golden/reference_gif_demo11.json).

What the markers showed (round 5): up to pose 22 this build's run lay on them; at steps 21-25 -- the car runs head-on into the first
moving box -- obca_mpc6 of this build ended 'infeasible' from all three starts and obca_mpc8 braked in front of the box, while the
reference drove around it.  The problems ARE feasible (SLSQP from the reference's marker poses finds a point, below): a solver
failure of this build, not IPOPT's.  The l1 penalty problem has a stationary point symmetric about the window there; the ladder's
dodge rung (include/obca_mpc.h: dodge) starts beside the window and finds the plan.  With it the run takes the reference's 52 steps
to the end of the fixed-time phase instead of 53 and shows the fourth title to 0.01 s instead of 0.076 s."""
import numpy as np
import pytest

from tests import independent, native_build, reference_report


@pytest.fixture(scope="module")
def fx():
    return reference_report.fixture()


@pytest.fixture(scope="module")
def gif():
    return reference_report.gif_demo11()


def run_errors(cum, x_closed, fx, gif):
    """-> [(step, |title - cumulative time|) x 4], distance of every marker to its pose"""
    titles = sorted(f["spend_time"] for f in fx["figure11_demo11"]["frames"])
    hits = reference_report.match(cum, titles)
    M, first = np.array(gif["markers_xy"]), gif["first_marker_is_pose"]
    X = np.asarray(x_closed)[first:first + len(M), :2]
    return hits, np.hypot(*(X - M).T)


def check_run(cum, x_closed, fx, gif):
    """what the build DOES show at reading precision: the four steps, three titles, the 19 markers of the straight part -- plus
    regression guards on the rest (the fourth title, the markers of the dodge), which it does not: see the strict xfails below"""
    hits, d = run_errors(cum, x_closed, fx, gif)
    assert [k for k, _ in hits] == reference_report.DEMO11_TITLE_STEPS, hits
    assert max(e for _, e in hits[:3]) <= reference_report.TIME_TOL, hits
    assert d[:19].max() <= reference_report.DEMO11_MARKER_ACCURACY + 0.005          # poses 2 .. 20: the straight part, before the first obstacle is met
    assert hits[3][1] <= reference_report.DEMO11_FOURTH_GUARD, hits                  # guard, not a tolerance (tests/reference_report.py)
    assert d.max() <= reference_report.DEMO11_MARKER_GUARD_MAX and d.mean() <= reference_report.DEMO11_MARKER_GUARD_MEAN, (d.max(), d.mean())
    return hits, d


def check_fourth_title_at_reading_precision(cum, x_closed, fx, gif):
    hits, _ = run_errors(cum, x_closed, fx, gif)
    assert hits[3][1] <= reference_report.TIME_TOL, "fourth title of Figure 11: %.4f s off (reading precision %.4f s)" % (hits[3][1], reference_report.TIME_TOL)


def check_markers_at_reading_precision(cum, x_closed, fx, gif):
    """one documented exception: markers 21 and 22 overlap in the recording, whose own spacing there (23.91 -> 25.04 m) is more than
    the 0.9974 m a step of 1.6623 s at 0.6 m/s allows -- either of them is off by at least 0.13 m in the fixture itself"""
    _, d = run_errors(cum, x_closed, fx, gif)
    first = gif["first_marker_is_pose"]
    keep = np.array([first + i not in (21, 22) for i in range(len(d))])
    assert d[keep].max() <= reference_report.DEMO11_MARKER_ACCURACY, "markers of the demo11 recording: up to %.2f m off (accuracy 0.15 m), %d of %d beyond it" % (
        d[keep].max(), int((d[keep] > reference_report.DEMO11_MARKER_ACCURACY).sum()), int(keep.sum()))


@pytest.fixture(scope="module")
def default_run():
    s = native_build.LpiObca()
    cum, cl = reference_report.replay(reference_report.demo11_setting(), s, 61)
    return cum, cl.x_closed


@pytest.mark.xfail(strict=True, reason="measured 0.0099 s against the 0.0055 s a title can be read to: the car is 0.02 m further on than IPOPT's when the fixed-time phase ends")
def test_fourth_title_of_figure_11_at_reading_precision(fx, gif, default_run):
    check_fourth_title_at_reading_precision(*default_run, fx, gif)


@pytest.mark.xfail(strict=True, reason="measured: 18 of 53 markers 0.16-0.33 m from their pose (accuracy 0.15 m) -- the dodge around the two boxes is driven up to 0.33 m beside the reference's")
def test_markers_of_the_demo11_recording_at_reading_precision(fx, gif, default_run):
    check_markers_at_reading_precision(*default_run, fx, gif)


@pytest.mark.parametrize("engine", ["lpi", "oracle"])
def test_demo11_run_shows_figure_11_and_lies_on_the_gif_markers(fx, gif, engine):
    s = native_build.LpiObca(engine)
    cum, cl = reference_report.replay(reference_report.demo11_setting(), s, 61)
    check_run(cum, cl.x_closed, fx, gif)
    v = [c["variant"] for c in s.calls]
    assert v[:15] == [4] * 15 and v[15:52] == [6] * 37 and v[52:60] == [4] * 8      # every obca_mpc6 of the dodge succeeds: no obca_mpc8
    assert all(c["status"] in (0, 1) for c in s.calls)


def test_without_the_dodge_rung_the_run_is_a_step_late_on_problems_that_are_feasible(fx, gif):
    """the round-4 state, kept as evidence: obca_mpc6 'infeasible' at steps 21, 23, 25 (and 29) although a feasible point exists"""
    from oracle.obca_nlp import Problem
    from vehicle_motion_planning_with_obstacles_avoidance_using_mpc_amd.solver import SolverParams
    s = native_build.LpiObca()
    s.dodge = False
    cum, cl = reference_report.replay(reference_report.demo11_setting(), s, 61)
    hits = reference_report.match(cum, sorted(f["spend_time"] for f in fx["figure11_demo11"]["frames"]))
    assert [k for k, _ in hits] == [23, 29, 39, 59] and 0.05 < hits[3][1] < 0.1          # the fourth title 0.076 s off
    M, first = np.array(gif["markers_xy"]), gif["first_marker_is_pose"]
    X = np.asarray(cl.x_closed)[first:first + len(M), :2]
    assert np.hypot(*(X - M).T)[21:].max() > 0.45                                        # ... and the poses off the markers from step 23 on
    failed6 = [i for i, c in enumerate(s.calls) if c["variant"] == 6 and c["status"] == 2]
    assert failed6[:3] == [20, 22, 24]
    c = s.calls[22]
    sp = SolverParams(xL=cl.xL[:2], xU=cl.xU[:2])
    p = Problem(6, 6, c["m"], c["x0"], c["u0"], c["xref"], c["A"], c["b"], c["Ts"], sp.Q_fix, sp.R_fix[0], sp.R_fix[1], sp.P_fix, sp.xL, sp.xU,
                sp.uL, sp.uU, sp.ego, sp.dmin, term=c["term"])
    pts = np.zeros((3, 7))
    pts[:2] = M[21 - first:28 - first].T                                                 # the reference's poses 21 .. 27 as the trajectory guess
    d = np.diff(pts[:2], axis=1)
    pts[2, :6] = np.arctan2(d[1], d[0])
    pts[2, 6] = pts[2, 5]
    r = independent.slsqp(p, independent.trajectory_start(p, pts), maxiter=400)
    assert r["viol"] <= independent.FEAS_TOL                                             # feasible: the 'infeasible' verdict was this build's failure
    # the default ladder now answers the same call with a plan at least as good as the one SLSQP reaches from the reference's poses
    o = native_build.lpi_solve(6, 6, c["m"], c["x0"][None], c["u0"][None], c["xref"][None], c["A"][None], c["b"][None], [c["Ts"]], c["term"][None],
                               __import__("oracle.c_oracle", fromlist=["x"]).default_params(xL=sp.xL, xU=sp.xU, uL=sp.uL, uU=sp.uU, ego=sp.ego, dmin=sp.dmin, single_start=1,
                                                                                           Qx=sp.Q_fix, Px=sp.P_fix, R1x=sp.R_fix[0], R2x=sp.R_fix[1]))
    assert o["status"][0] == 0 and o["info"][0, 0] <= r["f"] + 1e-6


def test_along_the_dodge_this_builds_optimum_is_the_one_the_references_poses_lead_to(gif):
    """the residual (fourth title 0.0099 s, markers up to 0.33 m) is accumulated state, not another local optimum: at steps of the
    fixed-time phase -- around the first box, at the peak beside the second, on the way back -- SLSQP on the pinned model started from
    the REFERENCE's marker poses (moved to this build's current pose) reaches this build's optimum (round 5: the same at 17 of 18
    steps tried, a HIGHER one at the 18th)"""
    from oracle.obca_nlp import Problem
    from vehicle_motion_planning_with_obstacles_avoidance_using_mpc_amd.solver import SolverParams
    M, first = np.array(gif["markers_xy"]), gif["first_marker_is_pose"]
    s = native_build.LpiObca()
    cum, cl = reference_report.replay(reference_report.demo11_setting(), s, 48)
    sp = SolverParams(xL=cl.xL[:2], xU=cl.xU[:2])
    for step in (30, 38, 46):
        c = s.calls[step]
        assert c["variant"] == 6 and c["status"] == 0
        p = Problem(6, 6, c["m"], c["x0"], c["u0"], c["xref"], c["A"], c["b"], c["Ts"], sp.Q_fix, sp.R_fix[0], sp.R_fix[1], sp.P_fix, sp.xL, sp.xU,
                    sp.uL, sp.uU, sp.ego, sp.dmin, term=c["term"])
        pts = np.zeros((3, 7))
        pts[:2] = M[step - first:step - first + 7].T
        pts[:2] += (c["x0"][:2] - pts[:2, 0])[:, None]
        d = np.diff(pts[:2], axis=1)
        pts[2, :6] = np.arctan2(d[1], d[0])
        pts[2, 6] = pts[2, 5]
        r = independent.slsqp(p, independent.trajectory_start(p, pts), maxiter=400)
        assert r["viol"] <= independent.FEAS_TOL and abs(r["f"] - c["info"][0]) <= 1e-4 * max(1.0, abs(c["info"][0])), (step, r["f"], c["info"][0])
