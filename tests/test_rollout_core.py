"""The device harness core (csrc/obca_rollout_core.h, compiled for the CPU) against the Python ``closedLoop``
mirror of the reference's loop (pinned to the reference by tests/test_harness.py, fixtures F1-F8): same worlds,
same solver code (CPU build of csrc/obca_lpi_core.h), step-by-step equality of what the solver is given and of
the closed-loop trajectory."""
import copy

import numpy as np
import pytest

from oracle import c_oracle
from tests import native_build
from vehicle_motion_planning_with_obstacles_avoidance_using_mpc_amd.closed_loop import closedLoop
from vehicle_motion_planning_with_obstacles_avoidance_using_mpc_amd.demo_setting import problemSetting
from vehicle_motion_planning_with_obstacles_avoidance_using_mpc_amd.rollouts import pack_worlds
from vehicle_motion_planning_with_obstacles_avoidance_using_mpc_amd.scenarios import make_world_c5

TOL = 1e-9          # fp64; libm vs numpy cos/sin/atan2 may differ in the last bit, the solves amplify that slightly


def host_rollout(setting, N, n_steps, start_order="default"):
    solver = native_build.LpiObca()
    solver.start_order = start_order
    cl = closedLoop(setting, solver=solver)
    cl.N_free = cl.N_fix = N
    steps = 0
    while steps < n_steps and not cl.goal_reached():
        steps += 1
        if not cl.step():
            break
    return cl, solver


def compare(settings, N, n_steps, start_order="default"):
    w = pack_worlds(copy.deepcopy(settings))
    out = native_build.rollout_run(w, N, c_oracle.default_params(start_order=start_order), n_steps)
    for i, st in enumerate(settings):
        cl, solver = host_rollout(copy.deepcopy(st), N, n_steps, start_order)
        k_host = cl.k
        assert out["steps"][i] == k_host, (i, out["steps"][i], k_host)
        # what the solver was given, step by step (the last host call may be a failed one)
        calls = [c for c in solver.calls]
        j = 0
        for k in range(min(n_steps, len(out["variant"][i]))):
            v = int(out["variant"][i, k])
            if v == 0:
                break
            c = calls[j]
            if v == 8:                                   # the mirror called obca_mpc6 first, then obca_mpc8
                assert c["variant"] == 6 and calls[j + 1]["variant"] == 8
                j += 1
                c = calls[j]
            assert c["variant"] == v, (i, k, c["variant"], v)
            np.testing.assert_allclose(out["xref"][i, k], c["xref"], rtol=0, atol=TOL, err_msg="xref %d %d" % (i, k))
            j += 1
        xc = np.asarray(cl.x_closed)
        np.testing.assert_allclose(out["x_closed"][i, :k_host + 1], xc[:k_host + 1], rtol=0, atol=1e-7)
        if k_host:
            np.testing.assert_allclose(out["T_closed"][i, :k_host], np.asarray(cl.T_closed), rtol=0, atol=1e-7)
            np.testing.assert_allclose(out["u_closed"][i, :k_host], np.asarray(cl.u_closed), rtol=0, atol=1e-7)
    return out


@pytest.mark.parametrize("demo", ["demo8", "demo1"])
def test_reference_demos_follow_the_mirror(demo):
    out = compare([problemSetting(demo)], 6, 8)
    assert out["steps"][0] >= 1
    if demo == "demo8":
        assert set(out["variant"][0, :out["steps"][0]].tolist()) >= {4, 6}     # reaches the fixed-time phase


def test_monte_carlo_worlds_follow_the_mirror():
    out = compare([make_world_c5(i) for i in range(4)], 5, 6)
    assert out["steps"].sum() >= 8


@pytest.mark.parametrize("order", ["window", "x0", "zeros"])
def test_other_start_orders_follow_the_mirror(order):
    """obca_params.start_order (include/obca_mpc.h): the harness hands obca_mpc6 the first start of the order only
    (single_start), obca_mpc4 / obca_mpc8 the whole ladder -- as the Python mirror does per call"""
    out = compare([make_world_c5(i) for i in range(5)], 5, 8, start_order=order)
    assert out["steps"].sum() >= 12
    compare([problemSetting("demo8")], 6, 8, start_order=order)
    if order == "window":
        assert out["iters"][out["variant"] == 4].mean() < 40


def test_warm_start_option_reaches_the_same_plans_in_fewer_iterations():
    """obca_rollouts_set_warm_start (NOT reference behaviour, off by default): steps start from the shifted previous
    plan.  On static worlds the closed-loop trajectory is the cold-start one to solver tolerance, in fewer
    interior-point iterations."""
    w = pack_worlds([make_world_c5(i, n_dyn=0) for i in range(6)])
    cold = native_build.rollout_run(w, 5, c_oracle.default_params(), 12)
    warm = native_build.rollout_run(w, 5, c_oracle.default_params(), 12, warm_mu=0.1)
    assert np.array_equal(cold["steps"], warm["steps"]) and np.all(cold["steps"] == 12)
    np.testing.assert_allclose(warm["x_closed"][:, :13], cold["x_closed"][:, :13], rtol=0, atol=1e-5)
    np.testing.assert_array_equal(warm["iters"][:, 0], cold["iters"][:, 0])        # step 0 is always a cold start
    # (the cold start of round 5 is the reference window: 16 iterations per step against the warm start's 12; with x0 first it was 40)
    assert warm["iters"][:, 1:12].mean() < 0.85 * cold["iters"][:, 1:12].mean()


def test_history_export_matches_the_mirror_lists():
    """N4: the batched history converts to the lists the reference's closedLoop hands to its plot routine"""
    from vehicle_motion_planning_with_obstacles_avoidance_using_mpc_amd.rollouts import reference_lists
    settings = [make_world_c5(i) for i in range(2)]
    w = pack_worlds(copy.deepcopy(settings))
    out = native_build.rollout_run(w, 5, c_oracle.default_params(), 5)
    for i, st in enumerate(settings):
        cl, _ = host_rollout(copy.deepcopy(st), 5, 5)
        lists = reference_lists(out, w, i)
        assert len(lists["x_openLoop"]) == len(cl.x_openLoop) == cl.k
        for a, b in zip(lists["x_openLoop"], cl.x_openLoop):
            np.testing.assert_allclose(a, b, rtol=0, atol=1e-7)
        np.testing.assert_allclose(np.array(lists["x_closed"]), np.array(cl.x_closed), rtol=0, atol=1e-7)
        np.testing.assert_allclose(lists["Ts_opt"], cl.T_closed, rtol=0, atol=1e-7)
        assert len(lists["dyn_loc"]) == len(cl.dyn_loc)
        for a, b in zip(lists["dyn_loc"], cl.dyn_loc):
            assert len(a) == len(b)
            for va, vb in zip(a, b):
                np.testing.assert_allclose(np.array(va[:5]), np.array(vb[:5]), rtol=0, atol=1e-9)
                assert va[5] == vb[5]


def test_fixed_time_horizon_twice_the_free_time_one():
    """H5 with a resampling ratio of 2 (N_free = 5, N_fix = 10; reference src/closed_loop.py:570-587: the first N_free
    segments of the window become int(N_fix/N_free) points each).  The device harness follows the Python mirror step by step;
    the mirror's update_path is the fixture-pinned restatement of the reference's.  (6/12 cannot run in the reference: its
    shift-in of the previous plan, :363-364, reads column 7 of a 7-column free-time plan -- IndexError.)"""
    settings = [problemSetting("demo8")]
    w = pack_worlds(copy.deepcopy(settings))
    out = native_build.rollout_run(w, 5, c_oracle.default_params(), 8, N_fix=10)
    solver = native_build.LpiObca()
    cl = closedLoop(copy.deepcopy(settings[0]), solver=solver)
    cl.N_free, cl.N_fix = 5, 10
    steps = 0
    while steps < 8 and not cl.goal_reached():
        steps += 1
        if not cl.step():
            break
    assert out["steps"][0] == cl.k and cl.k >= 3
    var = out["variant"][0, :cl.k].tolist()
    assert 4 in var and (6 in var or 8 in var)                       # both problem sizes were solved
    j = 0
    for k in range(cl.k):
        v = var[k]
        c = solver.calls[j]
        if v == 8:
            j += 1
            c = solver.calls[j]
        assert c["variant"] == v
        n1 = c["xref"].shape[1]
        assert n1 == (6 if v == 4 else 11)
        np.testing.assert_allclose(out["xref"][0, k][:, :n1], c["xref"], rtol=0, atol=TOL)
        j += 1
    np.testing.assert_allclose(out["x_closed"][0, :cl.k + 1], np.asarray(cl.x_closed)[:cl.k + 1], rtol=0, atol=1e-7)
    np.testing.assert_allclose(out["T_closed"][0, :cl.k], np.asarray(cl.T_closed), rtol=0, atol=1e-7)
