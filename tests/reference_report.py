"""Closed-loop frames of the reference's project report (fixture tests/golden/reference_report_figures.json, see
tests/golden/make_report_fixture.py) replayed through the ``closedLoop`` mirror: Figure 12 = demo1 as checked in, Figure 11 = a
corridor reconstructed from its frames.  Like the demo9 GIF the runs are longer than 30 steps, so the k = 30 stop of
src/closed_loop.py:426-427 is lifted."""
import json
import os

import numpy as np

from vehicle_motion_planning_with_obstacles_avoidance_using_mpc_amd.closed_loop import closedLoop
from vehicle_motion_planning_with_obstacles_avoidance_using_mpc_amd.demo_setting import problemSetting

HERE = os.path.dirname(os.path.abspath(__file__))
TIME_TOL = 0.005 + 5e-4          # a title is rounded to 0.01 s (tests/reference_gif.py)


def fixture():
    with open(os.path.join(HERE, "golden", "reference_report_figures.json")) as f:
        return json.load(f)


class _NoStop(closedLoop):
    def finish_step(self, result):
        go_on = super().finish_step(result)
        if not go_on and self.feas == True and not self.goal_reached():  # noqa: E712  (the k == 30 stop)
            self.done = False
            return True
        return go_on


def demo1_setting():
    return problemSetting("demo1")


def corridor_setting(fx):
    c = fx["figure11_corridor"]["setting"]
    xU = c["xU"]
    static = [[[xU[0], xU[1] - 1], [0, xU[1] - 1]], [[0, 1], [xU[0], 1]]]                       # demo8's walls (src/demo_setting.py:332-335)
    grid = [[[xU[0], xU[1] - 1], [0, xU[1] - 1], [0, xU[1]], [xU[0], xU[1]]], [[0, 1], [xU[0], 1], [xU[0], 0], [0, 0]]]
    ang = {"pi/2": np.pi / 2, "-pi/2": -np.pi / 2}
    dyn = [[ang.get(v, v) if isinstance(v, str) else v for v in d] for d in c["dyn"]]
    return problemSetting.from_world(xU, c["start"], c["goal"], static, grid, dyn, np.array(c["terminal_set"]), name="report_fig11")


def replay(setting, solver, n_steps):
    """-> cumulative spent time after step 1..n, the loop object"""
    cl = _NoStop(setting, solver=solver)
    for _ in range(n_steps):
        if not cl.step():
            break
    return np.cumsum(cl.T_closed), cl


def match(cum, titles):
    """for every title: (step whose cumulative time is nearest, distance)"""
    return [(int(np.argmin(np.abs(cum - t))) + 1, float(np.min(np.abs(cum - t)))) for t in titles]
