"""Occupancy grid -- host-side mirror of the reference's ``model_map.mapModel`` (src/model_map.py:14-101).
Only what the A* global planner needs (``shape2grid``); the morphology helpers of the reference are unused
there (and need skimage), so they are not provided."""
import numpy as np


class mapModel:
    def __init__(self, map_size, resolution):
        rows = int((map_size[1] - 1) / resolution) + 1
        cols = int((map_size[0] - 1) / resolution) + 1
        self.grid_map = np.zeros((rows, cols))
        self.resolution = resolution

    def shape2grid(self, org_gridMap, obstacle_location):
        """Mark the axis-aligned bounding box of every obstacle polygon (cells are 1 m squares)."""
        grid = self.grid_map if (isinstance(org_gridMap, list) and len(org_gridMap) == 0) else org_gridMap
        for poly in obstacle_location:
            xs = [p[0] / self.resolution for p in poly]
            ys = [p[1] / self.resolution for p in poly]
            x0, y0 = int(min(xs)), int(min(ys))
            nx = int(max(xs) - min(xs)) + 1
            ny = int(max(ys) - min(ys)) + 1
            grid[y0:y0 + ny, x0:x0 + nx] = 1
        return grid
