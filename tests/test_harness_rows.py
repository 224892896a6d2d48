"""Rows S4 / S5 of SURVEY 8a checked DIRECTLY on the device harness core: the half-space rows the harness
hands the solver -- static obstacles, then the moving rectangles predicted over the horizon (rebuild_lObs +
obstacle_H_Represent, reference src/demo_setting.py:457-473, src/model_obstacle.py:37-102) -- against fixture F2F3, captured
from the reference's own functions (tests/golden/make_golden.py), bit for bit.  The fixture's procedure is
update_obstacle(0, Ts), update_obstacle(1, Ts), update_obstacle_constraint(N, Ts, dynObs_exist); the harness runs the same
through its test hook: the harness part of step 0, then of step 1, with the inherited step length Ts and a pose from which the
lidar gate sees every moving obstacle.  CPU build of csrc/obca_rollout_core.h here, the GPU in the -m gpu twin below."""
import numpy as np
import pytest

from tests import native_build
from vehicle_motion_planning_with_obstacles_avoidance_using_mpc_amd.demo_setting import problemSetting
from vehicle_motion_planning_with_obstacles_avoidance_using_mpc_amd.rollouts import pack_worlds


def _cases(harness_golden):
    for c in harness_golden["F2F3_predict_hrep"]:
        w = pack_worlds([problemSetting(c["demo"])])
        M = sum(v - 1 for v in c["vObs"])
        A = np.array(c["AObs"]).reshape(c["N"] + 1, M, 2)
        b = np.array(c["bObs"]).reshape(c["N"] + 1, M)
        if c["dynObs_exist"]:
            centres = np.array([d[:2] for d in c["dyn_obs_info"]])
            pose = [centres[:, 0].mean() - 3.0, centres[:, 1].mean(), 0.0]      # the gate sees every moving obstacle from here
            yield c, w, [0, 1], pose, len(c["dyn_obs_info"]), 6, A, b
        else:
            yield c, w, [0], None, 0, 4, A, b


def test_cpu_build_of_the_harness_hands_the_solver_the_fixture_rows(harness_golden):
    n = 0
    for c, w, ks, pose, g, variant, A, b in _cases(harness_golden):
        var, Ad, bd = native_build.harness_rows(w, c["N"], ks, c["Ts"], pose, g)
        assert var == variant, (c["demo"], c["N"], c["Ts"], var)
        np.testing.assert_array_equal(Ad, A, err_msg="A %s N=%d Ts=%g" % (c["demo"], c["N"], c["Ts"]))
        np.testing.assert_array_equal(bd, b, err_msg="b %s N=%d Ts=%g" % (c["demo"], c["N"], c["Ts"]))
        n += 1
    assert n == 16


def test_demo1_moving_box_rows_of_survey_q11(harness_golden):
    """demo1's moving box at k = 0 gives rows [-1,0,-21], [0,1,1.5], [1,0,24], [0,-1,1.5]; only b changes
    with k (exact vertical / horizontal branches although cos(pi/2) is 6e-17)"""
    w = pack_worlds([problemSetting("demo1")])
    var, A, b = native_build.harness_rows(w, 5, [0, 1], 0.0, [19.0, 1.0, 0.0], 1)     # Ts = 0: the box stays where it appears
    assert var == 6
    rows = np.concatenate([A[0, -4:], b[0, -4:, None]], 1)
    np.testing.assert_array_equal(rows, np.array([[-1, 0, -21], [0, 1, 1.5], [1, 0, 24], [0, -1, 1.5]], float))
    assert all(np.array_equal(A[k, -4:], A[0, -4:]) for k in range(6))


@pytest.mark.gpu
def test_device_harness_hands_the_solver_the_fixture_rows(harness_golden):
    import torch
    from vehicle_motion_planning_with_obstacles_avoidance_using_mpc_amd.rollouts import DeviceRollouts
    n = 0
    for c, w, ks, pose, g, variant, A, b in _cases(harness_golden):
        dr = DeviceRollouts(w, N=c["N"])
        for k in ks:
            var, Ad, bd = dr.debug_harness(k, c["Ts"], pose, g)
        torch.cuda.synchronize()
        assert var[0] == variant, (c["demo"], c["N"], c["Ts"], var)
        np.testing.assert_array_equal(Ad[0], A, err_msg="A %s N=%d Ts=%g" % (c["demo"], c["N"], c["Ts"]))
        np.testing.assert_array_equal(bd[0], b, err_msg="b %s N=%d Ts=%g" % (c["demo"], c["N"], c["Ts"]))
        dr.close()
        n += 1
    assert n == 16
    w = pack_worlds([problemSetting("demo1")])
    dr = DeviceRollouts(w, N=5)
    for k in (0, 1):
        var, Ad, bd = dr.debug_harness(k, 0.0, [19.0, 1.0, 0.0], 1)
    rows = np.concatenate([Ad[0, 0, -4:], bd[0, 0, -4:, None]], 1)
    np.testing.assert_array_equal(rows, np.array([[-1, 0, -21], [0, 1, 1.5], [1, 0, 24], [0, -1, 1.5]], float))
