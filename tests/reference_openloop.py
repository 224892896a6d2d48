"""The reference's open-loop plan of demo9 at N = 50 (fixture tests/golden/reference_openloop_demo9.json, read off the picture
the reference repository holds -- see tests/golden/make_openloop_fixture.py) solved through the ``closedLoop`` mirror exactly as
``simulation.run_aStar`` does (src/simulation.py:114-123): ``N_free = 50``, ``mpc_openLoop_freeTime()`` (src/closed_loop.py:113-120),
with the weights the project report gives for that figure (Q = 0.5 I; the input weights are the checked-in ones)."""
import json
import os

import numpy as np

from vehicle_motion_planning_with_obstacles_avoidance_using_mpc_amd.closed_loop import closedLoop
from vehicle_motion_planning_with_obstacles_avoidance_using_mpc_amd.demo_setting import problemSetting

HERE = os.path.dirname(os.path.abspath(__file__))
# a marker centre is known to a pixel (0.091 m) -- plus the rounding of a three-pixel dot's centre of mass
MARKER_TOL = 0.12


def fixture():
    with open(os.path.join(HERE, "golden", "reference_openloop_demo9.json")) as f:
        return json.load(f)


def plan(solver=None, q=0.5, N=50):
    """-> the closedLoop mirror after the open-loop free-time solve (xOpt (3, N+1), feas, Ts_opt)"""
    cl = closedLoop(problemSetting("demo9"), solver=solver)
    cl.Q_free = q * np.eye(3)
    cl.P_free = cl.Q_free
    cl.N_free = N
    cl.mpc_openLoop_freeTime()
    return cl


def marker_distances(fx, xopt):
    """for every marker of the picture: distance to the nearest pose of the plan, and that pose's index"""
    m = np.asarray(fx["markers_xy"])
    X = np.asarray(xopt)[:2].T
    d = np.sqrt(((m[:, None, :] - X[None, :, :]) ** 2).sum(-1))
    return d.min(1), d.argmin(1)
