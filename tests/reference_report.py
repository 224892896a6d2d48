"""Closed-loop frames of the reference's project report (fixture tests/golden/reference_report_figures.json, see
tests/golden/make_report_fixture.py) replayed through the ``closedLoop`` mirror: Figure 12 = demo1, Figure 11 = demo11, both as
checked in (src/demo_setting.py).  The demo11 run is also the one recorded in images/OBCA_dynObs_demo11.gif, whose closed-loop
markers tests/golden/reference_gif_demo11.json holds.  Like the demo9 GIF the runs are longer than 30 steps, so the k = 30 stop of
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


def demo11_setting():
    return problemSetting("demo11")


def gif_demo11():
    with open(os.path.join(HERE, "golden", "reference_gif_demo11.json")) as f:
        return json.load(f)


# Figure 11 / the demo11 GIF: what this build's run shows (ONE place; tests/test_reference_demo11.py, tests/test_gpu_reference_demo11.py
# and bench.py read it).  Steps whose cumulative time is the title's; the first three lie in the fixed-time phase, where the step
# length is inherited -- they pin the fifteen free-time solves before it and the step count.  The fourth lies six
# free-time solves after the dodge: it measures WHERE the car is when the phase ends, to about 0.05 m.
DEMO11_TITLE_STEPS = [23, 29, 39, 59]
# What this build does NOT reproduce at the reading precision (stated as such: the strict-xfail tests of tests/test_reference_demo11.py
# and tests/test_gpu_reference_demo11.py fail on these with TIME_TOL / 0.15 m, and say so in every test report):
#   * the fourth title: 98.540 s here against 98.55 s -- 0.0099 s, 1.8 x the reading precision of a title (0.0055 s); before the
#     dodge rung of the ladder it was 0.076 s, a whole step late;
#   * 20 of the 55 markers: 0.16 .. 0.33 m away against a marker accuracy of 0.15 m (all in the dodge around the two boxes, poses 21-51;
#     the straight part before it, poses 2-20, is within 0.15 m).
# The two numbers below are REGRESSION GUARDS on that measured state -- not tolerances of the comparison.
DEMO11_FOURTH_MEASURED = 0.0099
DEMO11_FOURTH_GUARD = 0.011
DEMO11_MARKER_ACCURACY = 0.15
DEMO11_MARKER_GUARD_MAX, DEMO11_MARKER_GUARD_MEAN = 0.35, 0.15


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
