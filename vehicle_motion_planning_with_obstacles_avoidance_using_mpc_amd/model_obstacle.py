"""Obstacle half-space representation -- host-side mirror of the reference's ``model_obstacle.py``.

``obstacleModel.obstacle_H_Represent`` keeps the reference's name, arguments and output layout
(reference src/model_obstacle.py:37-102): consecutive (clockwise) vertex pairs become rows ``A p <= b``;
vertical and horizontal edges take the exact unit-normal branches, every other edge the slope form,
which is NOT normalised -- kept deliberately, because ``||A' lambda|| <= 1`` and
``dmin`` are expressed in those units by the reference.
"""
import numpy as np


def edge_halfspace(v1, v2):
    """One edge -> (a0, a1, b).  Branch order and the exact float comparisons follow the reference."""
    if v1[0] == v2[0]:                         # vertical edge
        if v2[1] < v1[1]:
            return 1.0, 0.0, float(v1[0])
        return -1.0, 0.0, -float(v1[0])
    if v1[1] == v2[1]:                         # horizontal edge
        if v1[0] < v2[0]:
            return 0.0, 1.0, float(v1[1])
        return 0.0, -1.0, -float(v1[1])
    a = (v2[1] - v1[1]) / (v2[0] - v1[0])
    b = v1[1] - a * v1[0]
    if v1[0] < v2[0]:
        return -a, 1.0, b
    return a, -1.0, -b


class obstacleModel:
    def obstacle_H_Represent(self, nOb, vOb, obstacle_vertex):
        """rows for obstacle 0, then obstacle 1, ...; returns A (sum(vOb)-nOb, 2), b (sum(vOb)-nOb, 1)."""
        rows = int(sum(int(v) for v in vOb[:nOb]) - nOb)
        A = np.zeros((rows, 2))
        b = np.zeros((rows, 1))
        r = 0
        for i in range(nOb):
            poly = obstacle_vertex[i]
            for j in range(int(vOb[i]) - 1):
                A[r, 0], A[r, 1], b[r, 0] = edge_halfspace(poly[j], poly[j + 1])
                r += 1
        return A, b


def rectangle_vertices(cx, cy, theta, length, width):
    """Clockwise rectangle, first vertex repeated (reference ``problemSetting.get_obstacle``,
    src/demo_setting.py:405-429)."""
    l, w = length / 2, width / 2
    c, s = np.cos(theta), np.sin(theta)
    v1 = [cx - l * c - w * s, cy - l * s + w * c]
    v2 = [cx + l * c - w * s, cy + l * s + w * c]
    v3 = [cx + l * c + w * s, cy + l * s - w * c]
    v4 = [cx - l * c + w * s, cy - l * s - w * c]
    return [v1, v2, v3, v4, v1]
