"""Row N3 of SURVEY.md 8(f): the reference's open-loop two-stage planner (src/simulation.py:20-62 ->
closedLoop.mpc_openLoop_freeTime, src/closed_loop.py:113-120, then mpc_openLoop_fixTime, :122-140: the free-time plan
resampled to N_fix points, obca_mpc6 with the setting's terminal set, obca_mpc8 where that fails) and the long horizons of
src/simulation.py:225-231 (N = 10: 3.69 s, N = 74: 136.7 s published), on the GPU through the drop-in `obca` class.

Checker: the same driver on the CPU build of the structured core (tests/native; same algorithm, pinned to the dense C
oracle by tests/test_lpi_core_cpu.py) -- equal feas flags, trajectories to 1e-6 -- plus model-level properties
(dynamics, bounds, terminal constraints, geometric clearance) that do not depend on any solver."""
import time

import numpy as np
import pytest

from tests import kkt_check, native_build
from vehicle_motion_planning_with_obstacles_avoidance_using_mpc_amd.closed_loop import closedLoop
from vehicle_motion_planning_with_obstacles_avoidance_using_mpc_amd.demo_setting import problemSetting

pytestmark = pytest.mark.gpu
EGO, DMIN = (1.7, 0.75, 1.7, 0.75), 0.05


def _gpu_solver():
    from vehicle_motion_planning_with_obstacles_avoidance_using_mpc_amd.obca import obca
    return obca()


def _dyn_res(x, u, h):
    nxt = x[:, :-1] + h * np.stack([u[0] * np.cos(x[2, :-1]), u[0] * np.sin(x[2, :-1]), u[1]])
    return float(np.max(np.abs(nxt - x[:, 1:])))


def _static_rows(cl, N):
    from vehicle_motion_planning_with_obstacles_avoidance_using_mpc_amd.solver import pack_reference_call
    m, _, _, _, A, b, _, _ = pack_reference_call(4, cl.Ts, N, cl.x0, np.zeros((3, N + 1)), cl.nObs, cl.vObs, cl.AObs, cl.bObs, cl.u0)
    return m, A, b


@pytest.mark.parametrize("demo,n_free,n_fix", [("demo1", 10, 20), ("demo8", 5, 10)])
def test_two_stage_open_loop_planner(demo, n_free, n_fix):
    """the reference's recommended open-loop setting (N_free = 10, N_fix = 20: resampling ratio 2, src/simulation.py:23-25)"""
    runs = {}
    for name, solver in (("gpu", _gpu_solver()), ("cpu", native_build.LpiObca())):
        cl = closedLoop(problemSetting(demo), solver=solver)
        cl.N_free, cl.N_fix = n_free, n_fix
        cl.mpc_openLoop_freeTime()
        free = (np.array(cl.xOpt), np.array(cl.uOpt), bool(cl.feas), float(cl.Ts_opt))
        m, A, b = _static_rows(cl, n_free)
        cl.mpc_openLoop_fixTime()
        runs[name] = dict(free=free, fix=(np.array(cl.xOpt), np.array(cl.uOpt), bool(cl.feas), float(cl.Ts_opt)), N_fix=cl.N_fix,
                          xref=np.array(cl.xref), term=np.array(cl.terminal_set, float), rows=(m, A, b), goal=cl.xF)
    g, c = runs["gpu"], runs["cpu"]
    for stage in ("free", "fix"):
        assert g[stage][2] == c[stage][2]
        assert g[stage][2], stage                                            # both stages succeed on these demos
        np.testing.assert_allclose(g[stage][0], c[stage][0], rtol=0, atol=1e-6)
        np.testing.assert_allclose(g[stage][1], c[stage][1], rtol=0, atol=1e-6)
        assert g[stage][3] == pytest.approx(c[stage][3], abs=1e-8)
    assert g["N_fix"] == n_fix and g["fix"][0].shape == (3, n_fix + 1)      # resampled: N_free segments x ratio + 1 points
    xf, uf, _, ts = g["free"]
    assert _dyn_res(xf, uf, ts) < 1e-7 and np.max(np.abs(xf[:, -1] - np.asarray(g["goal"], float))) < 1e-6   # reaches the goal
    m, A, b = g["rows"]
    assert kkt_check.min_clearance(xf, EGO, m, A, b) >= DMIN - 1e-6
    x2, u2, _, ts2 = g["fix"]
    assert ts2 == pytest.approx(n_free * ts / n_fix, rel=1e-12)             # Ts_opt <- N_free Ts_opt / N_fix (:586)
    assert _dyn_res(x2, u2, ts2) < 1e-7
    assert np.abs(u2[0]).max() <= 0.6 + 1e-7 and np.abs(u2[1]).max() <= np.pi / 6 + 1e-7


@pytest.mark.parametrize("demo,N", [("demo1", 40), ("demo1", 74), ("demo9", 74), ("demo9", 10)])
def test_long_horizon_free_time_solves(demo, N):
    """src/simulation.py:225-231: `mpc.N_free = 10 # np.size(a_start_path, 0)` -- 3.69 s at N = 10 and 136.7 s at N = 74
    on demo9.  Cold start, start/goal-only reference.  demo9 at N = 10 is infeasible by construction (the time-scale bound
    max_Topt allows a path of (dx + dy) + 0.6 m in ten straight segments, the obstacles need a detour) and is reported as
    such by GPU and CPU alike; the other three converge to the same plan."""
    runs = {}
    for name, solver in (("gpu", _gpu_solver()), ("cpu", native_build.LpiObca())):
        cl = closedLoop(problemSetting(demo), solver=solver)
        cl.N_free = N
        t = time.perf_counter()
        cl.mpc_openLoop_freeTime()
        runs[name] = (np.array(cl.xOpt), np.array(cl.uOpt), bool(cl.feas), float(cl.Ts_opt), time.perf_counter() - t, cl)
    g, c = runs["gpu"], runs["cpu"]
    print("%s N=%d: GPU %.3f s, one CPU core %.3f s (reference, unspecified hardware: %s)" %
          (demo, N, g[4], c[4], {10: "3.69 s", 74: "136.7 s"}.get(N, "not published")))
    assert g[2] == c[2]
    if (demo, N) == ("demo9", 10):
        assert not g[2]
        return
    assert g[2]
    np.testing.assert_allclose(g[0], c[0], rtol=0, atol=1e-5)
    assert g[3] == pytest.approx(c[3], abs=1e-7)
    x, u, _, ts = g[:4]
    cl = g[5]
    assert _dyn_res(x, u, ts) < 1e-7 and np.max(np.abs(x[:, -1] - np.asarray(cl.xF, float))) < 1e-6
    m, A, b = _static_rows(cl, N)
    assert kkt_check.min_clearance(x, EGO, m, A, b) >= DMIN - 1e-6
    assert np.abs(u[0]).max() <= 0.6 + 1e-7
