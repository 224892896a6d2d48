"""Batched global planner on the GPU -- B grid A* searches at once (reference src/a_star.py:16-102 followed by
rebuild_path / create_reference_path, :137-200), producing the (3,P) reference trajectories the closed loop tracks
(reference src/closed_loop.py:340: ``update_path(..., type="A_star")``) directly in device memory.

``a_star.a_star`` stays the readable per-rollout mirror of the reference class; tests compare the two."""
import ctypes

import numpy as np

from . import _lib


def yaw_table():
    """arctan2(dy, dx) for the nine lattice steps, evaluated by numpy like the reference's create_reference_path"""
    t = np.zeros(9)
    for dy in (-1, 0, 1):
        for dx in (-1, 0, 1):
            t[(dy + 1) * 3 + (dx + 1)] = np.arctan2(float(dy), float(dx))
    return t


def plan_batch(grids, starts, goals, path_max=None, device=None):
    """grids [B,rows,cols] (1 = occupied; ``setting.org_gridMap``), starts / goals [B,2] as (row, col) =
    (pose_y, pose_x).  Returns device tensors path [B,3,path_max] (x, y, yaw; padded with the last point) and
    path_len [B] (negative: -1 no route, -2 open list overflow, -3 path_max too small)."""
    import torch
    if not torch.cuda.is_available():
        raise RuntimeError("plan_batch needs a ROCm GPU; there is no CPU fallback on the product path")
    lib = _lib.load()
    dev = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
    if isinstance(grids, torch.Tensor):                      # e.g. the output of rasterise_batch: stays on the device
        g = grids.to(device=dev, dtype=torch.uint8).contiguous()
    else:
        g = torch.as_tensor(np.ascontiguousarray(grids), device=dev).to(torch.uint8).contiguous()
    B, rows, cols = g.shape
    st = torch.as_tensor(np.ascontiguousarray(starts), device=dev).to(torch.int32).contiguous()
    go = torch.as_tensor(np.ascontiguousarray(goals), device=dev).to(torch.int32).contiguous()
    P = int(path_max) if path_max is not None else rows * cols
    path = torch.empty(B, 3, P, dtype=torch.float64, device=dev)
    plen = torch.empty(B, dtype=torch.int32, device=dev)
    need = int(lib.obca_astar_workspace_bytes(B, rows, cols))
    if need < 0:
        raise ValueError("grid too large for the planner (rows*cols <= 65535)")
    work = torch.empty(need, dtype=torch.uint8, device=dev)
    yaw = (ctypes.c_double * 9)(*yaw_table().tolist())
    ptr = lambda t: ctypes.c_void_p(t.data_ptr())
    _lib.check(lib.obca_astar_batch(ptr(g), B, rows, cols, ptr(st), ptr(go), yaw, P, ptr(path), ptr(plen), ptr(work),
                                    need, ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)))
    # the launch is asynchronous: the tensors it reads must outlive it.  They are tied to the RESULT (a second call
    # before the stream drains must not free the first call's workspace), and the caching allocator keeps a freed block
    # out of other streams' hands until this stream has passed the free point.
    path._obca_keep = (g, st, go, work)
    for t in (g, st, go, work):
        t.record_stream(torch.cuda.current_stream(dev))
    return path, plen


def rasterise_batch(obstacle_lists, map_size, resolution=1.0, device=None):
    """Occupancy grids of B worlds in one launch (reference ``mapModel.shape2grid``, src/model_map.py:21-56).
    obstacle_lists: per world a list of polygons (vertex lists, e.g. ``setting.static_gridlObs``); map_size = (x, y) as in
    ``problemSetting`` (src/demo_setting.py:67-68).  Returns a device tensor [B, rows, cols] uint8 = ``org_gridMap``."""
    import torch
    if not torch.cuda.is_available():
        raise RuntimeError("rasterise_batch needs a ROCm GPU; there is no CPU fallback on the product path")
    lib = _lib.load()
    dev = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
    B = len(obstacle_lists)
    K = max(1, max(len(o) for o in obstacle_lists))
    boxes = np.full((B, K, 4), np.nan)
    for b, polys in enumerate(obstacle_lists):
        for k, poly in enumerate(polys):                       # reOrderVertex: the polygon's bounding box
            xs, ys = [float(p[0]) for p in poly], [float(p[1]) for p in poly]
            boxes[b, k] = (min(xs), min(ys), max(xs), max(ys))
    rows = int((map_size[1] - 1) / resolution) + 1
    cols = int((map_size[0] - 1) / resolution) + 1
    bx = torch.as_tensor(boxes, device=dev)
    grid = torch.empty(B, rows, cols, dtype=torch.uint8, device=dev)
    _lib.check(lib.obca_rasterise_batch(ctypes.c_void_p(bx.data_ptr()), B, K, float(resolution), rows, cols,
                                        ctypes.c_void_p(grid.data_ptr()), ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)))
    bx.record_stream(torch.cuda.current_stream(dev))
    return grid
