"""Scenario definitions and obstacle prediction -- host-side mirror of the reference's ``demo_setting.py``.

``problemSetting(demo_name)`` exposes the attributes and methods the reference's driver touches
(reference src/demo_setting.py:13-70, :374-473): ``startPose, goalPose, xL, xU, static_lObs, static_vObs,
dyn_obs_info, dyn_lObs, dyn_vObs, dyn_nObs, terminal_set, senseDis, org_gridMap`` and
``get_obstacle / add_dynamic_obstacle / combine_obstacle / rebuild_lObs``.  The demo table below is data
(map size, poses, wall/box vertices, moving-box tuples) taken from src/demo_setting.py:82-341.

Dynamic-obstacle tuple: [cx, cy, theta, length, width, speed, end_x, end_y, end_theta, t_start, t_end]
(11 entries; the reference's docstring lists 12 names, the data has 11 -- SURVEY.md H2).
"""
import math

import numpy as np

from .model_map import mapModel
from .model_obstacle import rectangle_vertices

PI = math.pi


def _corridor(xmax, box=None):
    top = [[xmax, 9], [0, 9]]
    bot = [[0, 1], [xmax, 1]]
    top_g = [[xmax, 9], [0, 9], [0, 10], [xmax, 10]]
    bot_g = [[0, 1], [xmax, 1], [xmax, 0], [0, 0]]
    if box is None:
        return [top, bot], [top_g, bot_g]
    return [top, box, bot], [top_g, box, bot_g]


_BOX_A = [[10, 1], [10, 5], [15, 5], [15, 1], [10, 1]]
_BOX_B = [[25, 8], [25, 3], [20, 3], [20, 8], [25, 8]]


def _dyn(cx, cy, th, v, ex, ey, t1, size=3):
    return [cx, cy, th, size, size, v, ex, ey, th, 0, t1]


# name -> (xU, start, goal, static polygons builder, dynamic tuples, terminal set)
_DEMOS = {
    "demo1": ((39, 10), [3, 4, 0], [38, 4, 0], lambda: _corridor(39, _BOX_A), [_dyn(22.5, 0, PI / 2, 0.2, 22.5, 9, 55)], [[25, 39], [1, 9]]),
    "demo2": ((39, 10), [3, 4, 0], [38, 4, 0], lambda: _corridor(39, _BOX_B), [_dyn(18.5, 0, PI / 2, 0.2, 18.5, 9, 55)], [[25, 39], [1, 9]]),
    "demo3": ((39, 10), [3, 4, 0], [38, 4, 0], lambda: _corridor(39, _BOX_B), [_dyn(18.5, 0, PI / 2, 0.15, 18.5, 9, 55)], [[25, 39], [1, 9]]),
    "demo4": ((39, 10), [3, 4, 0], [38, 4, 0], lambda: _corridor(39, _BOX_B), [_dyn(18.5, 0, PI / 2, 0.1, 18.5, 9, 55)], [[25, 39], [1, 9]]),
    "demo5": ((39, 10), [3, 4, 0], [38, 4, 0], lambda: _corridor(39, _BOX_A), [_dyn(22.5, 0, PI / 2, 0.1, 22.5, 9, 55)], [[25, 39], [1, 9]]),
    "demo6": ((39, 10), [3, 4, 0], [38, 4, 0], lambda: _corridor(39),
              [_dyn(13.5, 0, PI / 2, 0.2, 13.5, 9, 100), _dyn(22.5, 0, PI / 2, 0.1, 22.5, 9, 200)], [[25, 39], [1, 9]]),
    "demo7": ((39, 10), [3, 4, 0], [38, 4, 0], lambda: _corridor(39),
              [_dyn(13.5, 0, PI / 2, 0.1, 13.5, 9, 100), _dyn(22.5, 0, PI / 2, 0.05, 22.5, 9, 200)], [[28, 39], [1, 9]]),
    "demo8": ((39, 10), [3, 4, 0], [38, 4, 0], lambda: _corridor(39),
              [_dyn(13.5, 0, PI / 2, 0.1, 13.5, 9, 100), _dyn(22.5, 9, -PI / 2, 0.1, 22.5, 0, 200)], [[25, 39], [2, 6]]),
    "demo10": ((99, 10), [3, 4, 0], [98, 4, 0], lambda: _corridor(99), [_dyn(99, 5, -PI, 0.5, 0, 5, 100)], [[60, 99], [1, 9]]),
    "demo11": ((80, 10), [3, 4, 0], [77, 4, 0], lambda: _corridor(80),
               [_dyn(30.5, 0, PI / 2, 0.1, 30.5, 9, 100), _dyn(39.5, 9, -PI / 2, 0.1, 39.5, 0, 200)], [[25, 39], [2, 6]]),
}


def _demo9():
    lobs = [[[8, 0], [8, 6], [40, 6]],
            [[12, 30], [34, 30], [34, 14], [12, 14], [12, 30]],
            [[13, 49], [34, 49], [34, 34], [13, 34], [13, 49]],
            [[4, 60], [4, 10], [0, 10]],
            [[33, 60], [33, 55], [4, 55]]]
    grid = [[[8, 6], [40, 6], [40, 0], [8, 0]],
            [[12, 30], [34, 30], [34, 14], [12, 14]],
            [[12, 50], [34, 50], [34, 34], [12, 34]],
            [[0, 60], [4, 60], [4, 10], [0, 10]],
            [[4, 60], [34, 60], [34, 54], [4, 54]]]
    return lobs, grid


_DEMOS["demo9"] = ((40, 60), [1, 5, 0], [37, 58, PI / 2], _demo9, [_dyn(8, 50, -PI / 2, 0.5, 8, 10, 100, size=2)],
                   [[34, 40], [54, 60]])


class problemSetting:
    def __init__(self, demo_name):
        if demo_name not in _DEMOS:
            raise KeyError("unknown demo %r" % (demo_name,))
        xU, start, goal, polys, dyn, term = _DEMOS[demo_name]
        self._setup(demo_name, xU, start, goal, polys, dyn, term)

    @classmethod
    def from_world(cls, xU, start, goal, static_lObs, static_gridlObs, dyn, terminal_set, ref_path=None,
                   name="custom"):
        """A setting that is not in the reference's demo table (Monte-Carlo worlds, config C5).  ``ref_path``
        (3,P) replaces the A* reference when given."""
        self = cls.__new__(cls)
        self._setup(name, xU, start, goal, lambda: (static_lObs, static_gridlObs), dyn, terminal_set)
        self.ref_path = None if ref_path is None else np.asarray(ref_path, float)
        return self

    ref_path = None

    def _setup(self, demo_name, xU, start, goal, polys, dyn, term):
        self.demo_name = demo_name
        self.xL = [0, 0]
        self.xU = list(xU)
        self.map_size = [(self.xU[0] - self.xL[0]) + 1, (self.xU[1] - self.xL[1]) + 1]
        self.startPose = list(start)
        self.goalPose = list(goal)
        self.static_lObs, self.static_gridlObs = polys()
        self.terminal_set = np.array(term)
        self.static_nObs = len(self.static_lObs)
        self.static_vObs = np.array([len(p) for p in self.static_lObs], dtype=int)
        self.nObs, self.vObs, self.lObs, self.obs_info = 0, 0, 0, []
        self.add_dynamic_obstacle([list(d) for d in dyn])
        self.resolution = 1
        self.mapClass = mapModel(self.map_size, self.resolution)
        self.org_gridMap = self.mapClass.shape2grid([], self.static_gridlObs)
        self.grid_map = self.org_gridMap
        self.senseDis = 10

    # reference src/demo_setting.py:405-429
    def get_obstacle(self, center_x, center_y, theta, length, width):
        return rectangle_vertices(center_x, center_y, theta, length, width)

    # reference src/demo_setting.py:374-403
    def add_dynamic_obstacle(self, dyn_obs_info):
        self.dyn_lObs = [self.get_obstacle(d[0], d[1], d[2], d[3], d[4]) for d in dyn_obs_info]
        self.dyn_nObs = len(self.dyn_lObs)
        self.dyn_vObs = np.ones(self.dyn_nObs, dtype=int) * 5
        self.dyn_obs_info = dyn_obs_info

    # reference src/demo_setting.py:431-455 -- static first, then ALL stored dynamic vertex lists (q8: the lists
    # are not filtered by the lidar gate, only dyn_obs_info / dyn_nObs are)
    def combine_obstacle(self, dynObs_exist):
        nObs = self.static_nObs
        vObs = [int(v) for v in self.static_vObs]
        lObs = list(self.static_lObs)
        obs_info = [[0] * 11 for _ in range(self.static_nObs)]
        if dynObs_exist == 1:
            vObs += [int(v) for v in self.dyn_vObs]
            lObs += list(self.dyn_lObs)
            for i in range(self.dyn_nObs):
                obs_info.append(self.dyn_obs_info[i])
                nObs = self.static_nObs + self.dyn_nObs
        return nObs, vObs, lObs, obs_info

    # reference src/demo_setting.py:457-473: obstacle i at step k is translated by Ts*v*(cos th, sin th)*k
    def rebuild_lObs(self, N, Ts, dynObs_exist):
        nObs, vObs, lObs, obs_info = self.combine_obstacle(dynObs_exist)
        out = []
        for k in range(N + 1):
            for i in range(nObs):
                sx = Ts * obs_info[i][5] * np.cos(obs_info[i][2]) * k
                sy = Ts * obs_info[i][5] * np.sin(obs_info[i][2]) * k
                out.append([[lObs[i][j][0] + sx, lObs[i][j][1] + sy] for j in range(vObs[i])])
        self.nObs, self.vObs, self.lObs, self.obs_info = nObs, vObs, out, obs_info
