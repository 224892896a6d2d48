"""Solver-boundary pins on reference-held outputs NO default of this build was chosen on (found by the round-5 review, first used
in round 6):

* the two state-history plots of the project report (PDF objects 47 and 48; tests/golden/make_report_state_fixture.py): x, y and
  THETA of the N = 50 open-loop plan and of the demo9 closed loop -- theta is pinned by nothing else;
* every frame of the demo9 GIF (tests/golden/make_gif_pose_fixture.py): the car rectangle (pose with heading) and the magenta
  open-loop plan IPOPT returned at that step (six poses).

What they show (asserted below, dense C oracle and structured core alike):
  - steps 0-69 of the closed loop and all 51 knots of the plan lie ON the drawn curves: <= 1 px to the ink, read-outs <= 2 px
    (2 px = 0.04 rad on the closed-loop theta panel, 0.027 rad on the plan's);
  - the GIF's car boxes: <= 0.33 m / 0.08 rad on steps 0-69, and -- the run's tail -- every one of the 84 poses within 0.5 m, every
    heading after step 70 within 0.15 rad: after the one step where the two differ (70) this build drives the GIF's route;
  - the GIF's plans: frames 1-68, a pose of this build's plan on every marker (<= 1.8 px) and no ink off the plan (<= 4 px);
  - the reference's TWO records of its own run part from each other where they part from this build: at step 72 the state plot
    swings the heading to 2.5 rad and backs up, the GIF drives on at 0.9 rad (test_the_two_reference_records_part_at_step_71)."""
import numpy as np
import pytest

from tests import native_build, reference_gif, reference_openloop, reference_state as rs


@pytest.fixture(scope="module")
def closed():
    return rs.fixture("closedloop")


@pytest.fixture(scope="module")
def opened():
    return rs.fixture("openloop")


@pytest.fixture(scope="module")
def poses():
    return reference_gif.pose_fixture()


def test_fixture_shapes(closed, opened, poses):
    assert closed["steps"] == 86 and opened["steps"] == 51            # x_opt of 85 closed-loop steps; N + 1 = 51 knots
    for fx in (closed, opened):
        assert [p["name"] for p in fx["panels"]] == ["x", "y", "theta"]
        for p in fx["panels"]:
            assert max(p["tick_fit_residual_px"]) <= 0.8 and len(p["readout"]) == fx["steps"]
    assert abs(rs.panel(closed, "theta")["value_per_pixel"] - 0.0198) < 2e-4 and abs(rs.panel(opened, "theta")["value_per_pixel"] - 0.01366) < 2e-4
    b = np.asarray(poses["car_box"]["poses"])
    assert b.shape == (84, 4) and b[:, 3].max() < 1.0                   # every rectangle found: mean outline-to-ink distance below a pixel
    assert np.allclose(b[0, :3], [1.0, 5.0, 0.0], atol=0.2)             # the start pose of demo9 (the axes clip the rear of the box there: the worst fit of the run)
    assert len(poses["plan_ink"]["runs"]) == 84 and not poses["plan_ink"]["runs"][83]


def check_closed_loop_states(fx, xs, n=70):
    """closed-loop states xs (steps, 3) of this build on the reference's state plot, steps 0 .. n - 1"""
    r = rs.compare(fx, np.asarray(xs)[:n].T, np.arange(n))
    for name, (ink, read, n_flat) in r.items():
        assert ink <= 1.0 and read <= rs.PIXEL_TOL, (name, ink, read)
    assert r["theta"][2] >= 50 and r["x"][2] >= 65                     # the read-out rule leaves most steps in
    return r


def check_plan_states(fx, xopt):
    r = rs.compare(fx, np.asarray(xopt), np.arange(51))
    for name, (ink, read, n_flat) in r.items():
        assert ink <= 1.0 and read <= rs.PIXEL_TOL, (name, ink, read)
    assert r["theta"][1] <= 1.0 and r["theta"][2] >= 40               # theta: 1 px = 0.014 rad on 43 knots; the other 8 are the three flanks
    return r


def check_gif_frames(pf, xs, plans, n=70):
    dxy, dth = reference_gif.box_errors(pf, xs)
    assert dxy[:n].max() <= reference_gif.BOX_XY_TOL and dth[:n].max() <= reference_gif.BOX_THETA_TOL, (dxy[:n].max(), dth[:n].max())
    e = np.array([reference_gif.plan_errors(pf, k, plans[k]) for k in range(1, n - 1)])      # frame 0: the start marker hides pose 0
    assert e[:, 0].max() <= 1.8 and e[:, 1].max() <= 4.0, e.max(0)
    return dxy, dth, e


@pytest.mark.parametrize("engine", ["lpi", "oracle"])
def test_closed_loop_lies_on_the_references_state_plot_and_gif_frames(closed, poses, engine):
    s = native_build.LpiObca(engine)
    cum, xs, cl = reference_gif.replay(s, 70)
    check_closed_loop_states(closed, xs)
    check_gif_frames(poses, xs, cl.x_openLoop)


def test_open_loop_plan_lies_on_the_references_state_plot(opened):
    cl = reference_openloop.plan(native_build.LpiObca())
    assert cl.feas
    check_plan_states(opened, cl.xOpt)


def test_the_comparisons_have_teeth(closed, opened, poses):
    """the same checks fail for a run that is not the reference's: the plan with the checked-in Q = 0.1 I instead of the figure's
    0.5 I; the closed loop shifted by one step; IPOPT's plan of step 70 (the one this build does not return)"""
    cl = reference_openloop.plan(native_build.LpiObca(), q=0.1)
    r = rs.compare(opened, np.asarray(cl.xOpt), np.arange(51))
    assert max(v[0] for v in r.values()) > 3.0
    s = native_build.LpiObca()
    cum, xs, cl = reference_gif.replay(s, 71)
    r = rs.compare(closed, xs[1:71].T, np.arange(70))
    assert r["theta"][0] > 3.0 and r["y"][0] > 0.8          # y: one step = 1 m = 2.6 px of that panel
    e69 = reference_gif.plan_errors(poses, 69, cl.x_openLoop[69])
    assert e69[0] > 1.8 and e69[1] > 6.0                                # step 70: IPOPT's plan (Ts_opt 2.11 s) is another one (1.63 s here)


def test_whole_run_follows_the_gifs_route(poses, closed):
    """all 84 poses of the GIF's run: after step 70 (where IPOPT's plan takes 2.11 s and this build's 1.63 s, tests/test_reference_gif.py)
    the build is back on the GIF's boxes -- position <= 0.5 m at every pose, heading <= 0.15 rad from pose 71 on"""
    cum, xs, cl = reference_gif.replay(native_build.LpiObca(), 120)
    assert cl.goal_reached() and len(xs) == 85
    dxy, dth = reference_gif.box_errors(poses, xs)
    assert len(dxy) == 84 and dxy.max() <= 0.5, dxy.max()
    assert dth[71:].max() <= 0.15 and 0.25 < dth[70] < 0.4             # pose 70 itself: 0.26 rad in the GIF, 0.57 here
    e = np.array([reference_gif.plan_errors(poses, k, cl.x_openLoop[k]) for k in range(74, 83)])
    assert e[:, 0].max() <= 1.8 and e[:, 1].max() <= 4.0, e.max(0)     # ... and the plans from frame 74 on are the GIF's again


def test_the_two_reference_records_part_at_step_71(closed, poses):
    """The state plot (85 steps) and the GIF (84 steps) are two runs of the reference's own code.  They agree with each other -- and
    with this build -- up to step 70; from step 71 the state plot's run swings the heading to 2.5 rad (pose 72) while backing up
    (y falls from 53.9 to 53.5 m), the GIF's run turns to 0.9 rad and drives on to the goal.  What IPOPT returns on these steps is
    not one answer: the GIF's is the one this build follows."""
    th, flat = rs.readout(rs.panel(closed, "theta"))
    b = np.asarray(poses["car_box"]["poses"])
    assert np.abs(th[:70][flat[:70]] - b[:70, 2][flat[:70]]).max() <= 0.1          # the two records, steps 0-69: same headings
    assert abs(th[70] - b[70, 2]) <= 0.2                                            # both turn less than this build at step 70 (0.41 / 0.26 / 0.57)
    p = rs.panel(closed, "theta")
    peak = (min(c[1] for c in p["curve"]) - p["row_of_value"]["at_value_0"]) / p["row_of_value"]["per_unit"]
    assert peak > 2.5 and b[68:80, 2].max() < 1.15                                  # the swing exists in one record only
    y, _ = rs.readout(rs.panel(closed, "y"))
    assert y[73] < y[71] - 0.2 and np.all(np.diff(b[70:80, 1]) > 0)                 # ... and so does the backing up
