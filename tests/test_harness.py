"""Host-side harness (problem setup, H-representation, prediction, A*, windows, lidar gate, fixed-time
preparation, driver dispatch) against input/output pairs captured from the reference itself
(tests/golden/harness.json, produced by tests/golden/make_golden.py; SURVEY.md section 8a fixtures F1-F9)."""
import numpy as np
import pytest

from vehicle_motion_planning_with_obstacles_avoidance_using_mpc_amd.closed_loop import closedLoop
from vehicle_motion_planning_with_obstacles_avoidance_using_mpc_amd.demo_setting import problemSetting
from vehicle_motion_planning_with_obstacles_avoidance_using_mpc_amd.model_obstacle import obstacleModel


class NoSolver:
    pass


def make(demo):
    return closedLoop(problemSetting(demo), solver=NoSolver())


def test_F1_rectangle_vertices(harness_golden):
    st = problemSetting("demo1")
    for c in harness_golden["F1_get_obstacle"]:
        np.testing.assert_array_equal(np.array(st.get_obstacle(*c["rect"])), np.array(c["vertices"]))


def test_F2_F3_prediction_and_hrep(harness_golden):
    for c in harness_golden["F2F3_predict_hrep"]:
        cl = make(c["demo"])
        cl.update_obstacle(0, c["Ts"])
        cl.update_obstacle(1, c["Ts"])
        cl.update_obstacle_constraint(c["N"], c["Ts"], c["dynObs_exist"])
        assert cl.nObs == c["nObs"] and [int(v) for v in cl.vObs] == c["vObs"]
        assert len(cl.lObs) == len(c["lObs"])
        for a, b in zip(cl.lObs, c["lObs"]):
            np.testing.assert_array_equal(np.array(a), np.array(b))
        np.testing.assert_array_equal(cl.AObs, np.array(c["AObs"]))
        np.testing.assert_array_equal(cl.bObs, np.array(c["bObs"]))


def test_F3b_hrep_branches(harness_golden):
    c = harness_golden["F3b_hrep_polys"]
    v = np.array([len(p) for p in c["polys"]], dtype=int)
    A, b = obstacleModel().obstacle_H_Represent(len(c["polys"]), v, c["polys"])
    np.testing.assert_array_equal(A, np.array(c["A"]))
    np.testing.assert_array_equal(b, np.array(c["b"]))


def test_F9_astar_routes_and_grid(harness_golden):
    for c in harness_golden["F9_astar"]:
        st = problemSetting(c["demo"])
        np.testing.assert_array_equal(np.array(st.org_gridMap).astype(int), np.array(c["grid"]))
        cl = closedLoop(st, solver=NoSolver())
        ref = cl.update_path(0, cl.x0, cl.xF, 0, "A_star")
        np.testing.assert_array_equal(ref, np.array(c["ref"]))


def test_F4_windows(harness_golden):
    refs = {c["demo"]: np.array(c["ref"]) for c in harness_golden["F9_astar"]}
    cl = make("demo1")
    for c in harness_golden["F4_windows"]:
        w = cl.update_reference_trajectory(c["N"], refs[c["demo"]], c["pose"])
        np.testing.assert_array_equal(w, np.array(c["window"]))


def test_F5_F6_advance_and_lidar_gate(harness_golden):
    for c in harness_golden["F5F6_sensor_advance"]:
        cl = make(c["demo"])
        for s in c["steps"]:
            cl.x0 = np.array(s["pose"], dtype=float)
            cl.update_obstacle(s["k"], s["Ts_opt"])
            verts = [[list(p) for p in o[:5]] for o in cl.dyn_loc[-1]]
            np.testing.assert_array_equal(np.array(verts), np.array(s["dyn_vertices"]))
            cl.sensor()
            assert [int(o[5]) for o in cl.dyn_loc[-1]] == s["flags"]
            assert cl.fixtime == s["fixtime"]
            assert cl.setting.dyn_nObs == s["dyn_nObs"]
            np.testing.assert_array_equal(np.array(cl.dyn_orignal_info), np.array(s["dyn_table"]))
            np.testing.assert_array_equal(np.array(cl.setting.dyn_obs_info).reshape(-1),
                                          np.array(s["sensed_info"]).reshape(-1))


def test_F7_fixed_time_preparation(harness_golden):
    for c in harness_golden["F7_fixtime_prep"]:
        cl = make(c["demo"])
        cl.N_free = cl.N_fix = c["N_free"]
        ref = np.array(c["ref"])
        cl.x0 = np.array(c["x0"])
        cl.xOpt = np.array(c["xOpt_prev"])
        cl.Ts_opt, cl.Ts = c["Ts_opt_in"], 0.1
        cl.xref = cl.update_reference_trajectory(cl.N_fix, ref, cl.x0)
        np.testing.assert_array_equal(cl.xref, np.array(c["window"]))
        for i in range(cl.N_fix - 5):
            cl.xref[:, i] = cl.xOpt[:, i + 1]
        cl.xref = cl.update_path(0, 0, 0, allAviable=1, type="")
        np.testing.assert_allclose(cl.xref, np.array(c["xref"]), rtol=0, atol=0)
        assert cl.N_fix == c["N_fix"] and cl.Ts_opt == c["Ts_opt"] and cl.Ts == c["Ts"]


class ScriptedSolver:
    """the same stand-in the fixture generator used: follows xref, Ts_opt = 2.0 for free-time solves"""

    def __init__(self):
        self.calls = []

    def _ret(self, name, Ts, N, x0, xref, nObs, vObs, AObs, bObs, u0, free, term=None):
        self.calls.append(dict(variant=name, Ts=float(Ts), N=int(N), x0=np.asarray(x0, float), u0=np.asarray(u0, float),
                               xref=np.asarray(xref, float), nObs=int(nObs), vObs=[int(v) for v in vObs],
                               AObs_shape=list(np.shape(AObs)), bObs_sum=float(np.sum(bObs)),
                               AObs_first=np.asarray(AObs)[:int(sum(vObs[:nObs]) - nObs)], terminal_set=term))
        xo = np.array(xref, dtype=float)[:, :N + 1].copy()
        xo[:, 0] = np.asarray(x0, float)
        return xo, np.tile(np.array([[0.5], [0.01]]), (1, N)), True, (2.0 if free else Ts)

    def obca_mpc4(self, Ts, P, Q, R, N, x0, xL, xU, uL, uU, xref, nObs, vObs, AObs, bObs, dmin, ego, u0):
        return self._ret("mpc4", Ts, N, x0, xref, nObs, vObs, AObs, bObs, u0, True)

    def obca_mpc6(self, Ts, P, Q, R, N, x0, xL, xU, uL, uU, xref, nObs, vObs, AObs, bObs, dmin, ego, u0, uOpt, ts):
        return self._ret("mpc6", Ts, N, x0, xref, nObs, vObs, AObs, bObs, u0, False, ts)

    def obca_mpc8(self, *a):
        raise AssertionError("not expected")


@pytest.mark.parametrize("i", [0, 1, 2])
def test_F8_driver_trace(harness_golden, i):
    c = harness_golden["F8_driver_trace"][i]
    s = ScriptedSolver()
    cl = closedLoop(problemSetting(c["demo"]), solver=s)
    cl.N_free = cl.N_fix = c["N"]
    cl.closed_loop_mpc4()
    assert [q["variant"] for q in s.calls] == [q["variant"] for q in c["calls"]]
    for mine, ref in zip(s.calls, c["calls"]):
        assert mine["N"] == ref["N"] and mine["nObs"] == ref["nObs"] and mine["vObs"] == ref["vObs"]
        assert mine["AObs_shape"] == ref["AObs_shape"]
        assert mine["Ts"] == ref["Ts"]                                     # q7: Ts = 2.0 after the first fixed-time step
        np.testing.assert_array_equal(mine["x0"], np.array(ref["x0"]))
        np.testing.assert_array_equal(mine["u0"], np.array(ref["u0"]))
        np.testing.assert_allclose(mine["xref"], np.array(ref["xref"]), rtol=0, atol=0)
        np.testing.assert_array_equal(mine["AObs_first"], np.array(ref["AObs_first"]))
        assert mine["bObs_sum"] == ref["bObs_sum"]
        if ref["terminal_set"] is not None:
            np.testing.assert_array_equal(np.asarray(mine["terminal_set"]), np.array(ref["terminal_set"]))
    np.testing.assert_array_equal(cl.xOpt, np.array(c["x_closed"]))
    assert cl.Ts_opt == c["Ts_opt_list"]


def test_update_path_unknown_type_returns_zeros_like_the_reference():
    """src/closed_loop.py:529-566: a reference type the routine does not know falls through every branch and the all-zero ref_x is
    returned; the mirror does the same (with a warning), it does not raise"""
    import warnings
    cl = closedLoop(problemSetting("demo1"), solver=NoSolver())
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        ref = cl.update_path(0, cl.x0, cl.xF, 0, "no_such_type")
    assert len(w) == 1 and not np.asarray(ref).any()
