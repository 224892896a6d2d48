"""Global planner -- host-side mirror of the reference's ``a_star`` (src/a_star.py:16-200).

8-connected grid A* with Euclidean step cost and heuristic.  To return the reference's routes exactly the
search keeps its conventions: the open list is a binary heap of ``(f, (row, col))`` tuples (ties broken by the
cell tuple), neighbours are expanded in the order E, W, S, N, SE, SW, NE, NW of (row, col) offsets, a closed
cell is re-opened only on a strictly better g, and the returned chain runs goal -> start WITHOUT the start
cell (src/a_star.py:56-61).
"""
import heapq
import math

import numpy as np

_NEIGHBOURS = ((0, 1), (0, -1), (1, 0), (-1, 0), (1, 1), (1, -1), (-1, 1), (-1, -1))


class a_star:
    def __init__(self, array, start, goal):
        self.neighbors = list(_NEIGHBOURS)
        self._reset(start, goal)

    def _reset(self, start, goal):
        self.close_set = set()
        self.came_from = {}
        self.gscore = {start: 0}
        self.fscore = {start: self.heuristic(start, goal)}
        self.oheap = [(self.fscore[start], start)]

    @staticmethod
    def heuristic(a, b):
        return np.sqrt((b[0] - a[0]) ** 2 + (b[1] - a[1]) ** 2)

    def solve(self, array, start, goal):
        rows, cols = array.shape
        open_cells = {start}
        while self.oheap:
            current = heapq.heappop(self.oheap)[1]
            if current == goal:
                chain = []
                while current in self.came_from:
                    chain.append(current)
                    current = self.came_from[current]
                return chain
            self.close_set.add(current)
            for di, dj in self.neighbors:
                nb = (current[0] + di, current[1] + dj)
                g = self.gscore[current] + self.heuristic(current, nb)
                if not (0 <= nb[0] < rows and 0 <= nb[1] < cols) or array[nb[0]][nb[1]] == 1:
                    continue
                if nb in self.close_set and g >= self.gscore.get(nb, 0):
                    continue
                in_open = any(e[1] == nb for e in self.oheap)
                if g < self.gscore.get(nb, 0) or not in_open:
                    self.came_from[nb] = current
                    self.gscore[nb] = g
                    self.fscore[nb] = g + self.heuristic(nb, goal)
                    heapq.heappush(self.oheap, (self.fscore[nb], nb))
                    open_cells.add(nb)
        return False

    def rebuild_path(self, route):
        """goal->start chain of (row, col) cells -> start->goal list of [x, y] (src/a_star.py:137-147)"""
        return [[c[1], c[0]] for c in reversed([tuple(r) for r in np.asarray(route).tolist()])]

    def create_reference_path(self, path):
        """append the yaw of each segment; the last point repeats the previous yaw (src/a_star.py:189-200)"""
        out = []
        for i in range(len(path) - 1):
            yaw = np.arctan2(path[i + 1][1] - path[i][1], path[i + 1][0] - path[i][0])
            out.append([path[i][0], path[i][1], yaw])
        out.append([path[-1][0], path[-1][1], out[-1][2]])
        return out
