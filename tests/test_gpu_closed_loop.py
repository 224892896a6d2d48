"""Closed-loop drop-in on the GPU: the reference-style driver (closedLoop.closed_loop_mpc4) with the MI355X
solver behind obca(), and the batched driver."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module", autouse=True)
def built():
    import __graft_entry__ as ge
    ge.build()


def test_demo8_closed_loop_reaches_fixed_time_phase():
    from vehicle_motion_planning_with_obstacles_avoidance_using_mpc_amd.closed_loop import closedLoop
    from vehicle_motion_planning_with_obstacles_avoidance_using_mpc_amd.demo_setting import problemSetting
    cl = closedLoop(problemSetting("demo8"))
    cl.N_free = cl.N_fix = 5
    x_open, x_closed, u_closed, T_closed = cl.closed_loop_mpc4()
    assert len(x_closed) >= 3                                   # several steps were solved
    # step 0 is the analytic demo8 solve: Ts_opt = 2.0, first move 1.2 m along +x
    assert T_closed[0] == pytest.approx(2.0, abs=1e-6)
    np.testing.assert_allclose(x_closed[1], [4.2, 4.0, 0.0], atol=1e-6)
    np.testing.assert_allclose(u_closed[0], [0.6, 0.0], atol=1e-6)
    xs = np.asarray(x_closed)
    # makes progress along the corridor (a fixed-time plan may legitimately wait or back up for a moving box: u in [-0.6, 0.6])
    assert xs[-1, 0] > xs[0, 0] + 5.0 and np.all(np.diff(xs[:, 0]) > -0.6 * 2.5)
    assert np.all((xs[:, 1] > 1.75) & (xs[:, 1] < 8.25))        # stays clear of both walls


def test_demo1_first_steps_match_known_answer():
    from vehicle_motion_planning_with_obstacles_avoidance_using_mpc_amd.closed_loop import closedLoop
    from vehicle_motion_planning_with_obstacles_avoidance_using_mpc_amd.demo_setting import problemSetting
    cl = closedLoop(problemSetting("demo1"))
    assert cl.step()
    assert cl.T_closed[0] == pytest.approx(2.0378864, abs=2e-6)          # demo1, N=6 (Appendix C)
    np.testing.assert_allclose(cl.x0, [4.222732, 4.0, 0.332888], atol=2e-5)


def test_batched_rollouts_equal_single_rollouts():
    from vehicle_motion_planning_with_obstacles_avoidance_using_mpc_amd.closed_loop import BatchClosedLoop, closedLoop
    from vehicle_motion_planning_with_obstacles_avoidance_using_mpc_amd.demo_setting import problemSetting
    demos = ["demo8", "demo1", "demo8", "demo5"]

    def fresh(d):
        c = closedLoop(problemSetting(d))
        if d == "demo8":
            c.N_free = c.N_fix = 5
        return c

    batch = BatchClosedLoop([fresh(d) for d in demos]).run(max_steps=4)
    singles = [fresh(d) for d in demos]
    for c in singles:
        for _ in range(4):
            if not c.step():
                break
    assert batch.steps_solved >= 8
    for a, b in zip(batch.rollouts, singles):
        assert a.k == b.k
        np.testing.assert_array_equal(np.asarray(a.x_closed), np.asarray(b.x_closed))   # same kernel, same inputs
        assert a.T_closed == b.T_closed


def test_open_loop_free_time_planner_demo1():
    """reference `mpc_openLoop_freeTime` (src/closed_loop.py:113-120) at the recommended N_free = 10: needs the penalty
    escalation (rho 1e4 -> 1e6); the plan goes round the box and ends at the goal"""
    import numpy as np
    from vehicle_motion_planning_with_obstacles_avoidance_using_mpc_amd.closed_loop import closedLoop
    from vehicle_motion_planning_with_obstacles_avoidance_using_mpc_amd.demo_setting import problemSetting
    cl = closedLoop(problemSetting("demo1"))
    cl.N_free = 10
    cl.mpc_openLoop_freeTime()
    assert cl.feas == True  # noqa: E712
    assert cl.xOpt[1].max() > 5.5 and abs(cl.xOpt[0, -1] - 38.0) < 1e-6 and abs(cl.xOpt[1, -1] - 4.0) < 1e-6
    h = cl.Ts_opt
    x, u = cl.xOpt, cl.uOpt
    nxt = x[:, :-1] + h * np.stack([u[0] * np.cos(x[2, :-1]), u[0] * np.sin(x[2, :-1]), u[1]])
    assert np.max(np.abs(nxt - x[:, 1:])) < 1e-7
