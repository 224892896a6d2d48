"""Seeded synthetic workloads of SURVEY.md section 8(d) (configs C2 / C4): random start poses in the demo1
corridor with one random box, A*-like lattice reference, N+1-point window, obca_mpc4 inputs.

World: corridor ``xL=[0,0]``, ``xU=[39,10]`` with walls y >= 9 and y <= 1 (one half-space each, exactly the
rows the reference derives for demo1: src/demo_setting.py:93-95 -> src/model_obstacle.py:37-102) and one
axis-aligned box.  Instance ``i`` uses ``numpy.random.default_rng(seed0 + i)``; windows whose start or terminal
pose is in collision, or whose time-scale bound ``max_Topt`` (signed sum, src/obca.py:961-962) cannot cover
the window length, are redrawn from the same generator, so the batch is reproducible from ``seed0`` alone.
"""
import math
import os

import numpy as np

from .model_obstacle import obstacleModel, rectangle_vertices

SEED0 = 20260928
XL, XU = (0.0, 0.0), (39.0, 10.0)
EGO = (1.7, 0.75, 1.7, 0.75)
DMIN = 0.05
TS = 0.1
V_MAX = 0.6


def lattice_path(y0, box, y_goal=None):
    """8-connected polyline of 1 m steps from x=3 to x=38, detouring round ``box`` = (cx, cy, l, w)."""
    cx, cy, l, w = box
    x_lo, x_hi = cx - l / 2, cx + l / 2
    y_lo, y_hi = cy - w / 2, cy + w / 2
    need = 0.75 + DMIN + 0.55                       # half width + clearance + margin
    blocked = (y_lo - need) < y0 < (y_hi + need)
    y_det = y0
    if blocked:
        up = math.ceil(y_hi + need)
        dn = math.floor(y_lo - need)
        cands = [y for y in (up, dn) if 3 <= y <= 7]     # rows 2 and 8 leave no room to turn next to the walls
        if not cands:
            return None
        y_det = min(cands, key=lambda y: abs(y - y0))
    pts = []
    x, y = 3, y0
    x_start_det = math.floor(x_lo - 2.5) - abs(y_det - y0)        # diagonal finishes ~2.5 m before the box
    x_end_det = math.ceil(x_hi + 2.5)
    while x <= 38:
        pts.append((float(x), float(y)))
        if blocked and x >= x_start_det and x < x_end_det and y != y_det:
            y += 1 if y_det > y else -1
        elif blocked and x >= x_end_det and y != y0:
            y += 1 if y0 > y else -1
        x += 1
    if blocked and x_start_det < 3:
        return None
    path = np.zeros((3, len(pts)))
    path[0] = [p[0] for p in pts]
    path[1] = [p[1] for p in pts]
    for i in range(len(pts) - 1):                                  # yaw rule of a_star.create_reference_path
        path[2, i] = math.atan2(path[1, i + 1] - path[1, i], path[0, i + 1] - path[0, i])
    path[2, -1] = path[2, -2]
    return path


def window(path, pose, N):
    """closest-point window (reference closedLoop.update_reference_trajectory, src/closed_loop.py:502-528)"""
    d = (pose[0] - path[0]) ** 2 + (pose[1] - path[1]) ** 2
    i0 = int(np.argmin(d))                                          # first minimum, like the strict '<' loop
    P = path.shape[1]
    idx = np.minimum(i0 + np.arange(N + 1), P - 1)
    return path[:, idx].copy()


def _car_corners(pose):
    x, y, th = pose
    c, s = math.cos(th), math.sin(th)
    out = []
    for dx, dy in ((EGO[0], EGO[1]), (EGO[0], -EGO[3]), (-EGO[2], -EGO[3]), (-EGO[2], EGO[1])):
        out.append((x + c * dx - s * dy, y + s * dx + c * dy))
    return np.array(out)


def _poly_distance(Pa, Pb):
    """distance between two convex polygons (vertex arrays); 0 when they overlap"""
    best = 0.0
    for poly, other in ((Pa, Pb), (Pb, Pa)):
        n = len(poly)
        for i in range(n):
            e = poly[(i + 1) % n] - poly[i]
            nrm = np.array([e[1], -e[0]])
            ln = np.linalg.norm(nrm)
            if ln == 0:
                continue
            nrm = nrm / ln
            if np.max((poly - poly[i]) @ nrm) > 1e-12:
                nrm = -nrm
            gap = np.min((other - poly[i]) @ nrm)
            best = max(best, gap)
    if best > 0:                                                    # separated: refine with vertex-edge distances
        dmin = np.inf
        for A, B in ((Pa, Pb), (Pb, Pa)):
            for p in A:
                for i in range(len(B)):
                    a, b = B[i], B[(i + 1) % len(B)]
                    t = np.clip(np.dot(p - a, b - a) / max(np.dot(b - a, b - a), 1e-300), 0, 1)
                    dmin = min(dmin, np.linalg.norm(p - (a + t * (b - a))))
        return dmin
    return 0.0


def clearance(pose, box):
    car = _car_corners(pose)
    cx, cy, l, w = box
    bx = np.array([[cx - l / 2, cy - w / 2], [cx - l / 2, cy + w / 2], [cx + l / 2, cy + w / 2],
                   [cx + l / 2, cy - w / 2]])
    d = _poly_distance(car, bx)
    d = min(d, 9.0 - np.max(car[:, 1]), np.min(car[:, 1]) - 1.0)
    return d


def make_instance(i, N=5, seed0=SEED0, three_boxes=False):
    rng = np.random.default_rng(seed0 + i)
    om = obstacleModel()
    for _ in range(200):
        box = (rng.uniform(12, 30), rng.uniform(2.5, 7.5), rng.uniform(2, 5), rng.uniform(2, 5))
        y0 = int(round(rng.uniform(3, 7)))
        path = lattice_path(y0, box)
        noise = rng.uniform([-0.3, -0.3, -0.2], [0.3, 0.3, 0.2])
        u0 = np.array([rng.uniform(0, 0.6), rng.uniform(-0.1, 0.1)])
        if path is None:
            continue
        i0 = int(rng.integers(0, path.shape[1] - N - 1))
        x0 = path[:, i0] + noise
        xref = window(path, x0, N)
        seg = np.diff(xref[:2], axis=1)
        length = float(np.sum(np.hypot(seg[0], seg[1]))) + float(np.hypot(*(xref[:2, 0] - x0[:2])))
        dis = (xref[0, N] - x0[0]) + (xref[1, N] - x0[1])
        tmax = dis / (N * V_MAX * TS) + 1.0
        if tmax < 1.15 * length / (N * V_MAX * TS):
            continue
        if clearance(x0, box) < DMIN + 0.1 or clearance(xref[:, N], box) < DMIN + 0.1:
            continue
        if abs(xref[2, N] - x0[2]) > 1.0:
            continue
        rect = rectangle_vertices(box[0], box[1], 0.0, box[2], box[3])
        if three_boxes:
            polys = [[[39, 9], [0, 9], [0, 10], [39, 10], [39, 9]], rect, [[0, 1], [39, 1], [39, 0], [0, 0], [0, 1]]]
        else:
            polys = [[[39, 9], [0, 9]], rect, [[0, 1], [39, 1]]]          # demo1-style walls (1 row each)
        v = [len(p) for p in polys]
        A, b = om.obstacle_H_Represent(len(polys), v, polys)
        return dict(x0=x0, u0=u0, xref=xref, A=A, b=b[:, 0], m=[n - 1 for n in v], box=box, path=path)
    raise RuntimeError("could not draw a feasible instance for seed %d" % (seed0 + i))


def _c2_chunk(args):
    first, count, N, three_boxes = args
    return [make_instance(i, N, SEED0, bool(three_boxes)) for i in range(first, first + count)]


def _spawn_chunks(kind, spans, extra):
    """worker PROCESSES started from scratch (python -m ...scenarios KIND FIRST COUNT EXTRA... OUT), one per (first, count)
    span: never a fork of a process that holds a HIP context, and no dependence on how the caller's __main__ was started.
    Returns the unpickled results in span order."""
    import pickle
    import subprocess
    import sys
    import tempfile
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    res = []
    with tempfile.TemporaryDirectory() as tmp:
        jobs = []
        for j, (first, count) in enumerate(spans):
            out = os.path.join(tmp, "%s_%d.pkl" % (kind, j))
            jobs.append((out, subprocess.Popen([sys.executable, "-m", __name__, kind, str(first), str(count)] +
                                               [str(e) for e in extra] + [out], cwd=root)))
        for out, pr in jobs:
            if pr.wait() != 0:
                raise RuntimeError("scenario worker failed")
            with open(out, "rb") as f:
                res.append(pickle.load(f))
    return res


def make_batch(B, N=5, seed0=SEED0, three_boxes=False, first=0, procs=1):
    """Arrays in the layout of include/obca_mpc.h for instances first .. first+B-1 (variant 4).  procs > 1 draws them in that
    many worker processes (same result: every instance depends on its own seed only; default seed0 only)."""
    if procs > 1 and B >= 4 * procs and seed0 == SEED0:
        step = (B + procs - 1) // procs
        spans = [(first + lo, min(step, B - lo)) for lo in range(0, B, step)]
        ins = [q for part in _spawn_chunks("c2", spans, (N, int(three_boxes))) for q in part]
    else:
        ins = [make_instance(first + i, N, seed0, three_boxes) for i in range(B)]
    m = ins[0]["m"]
    M = sum(m)
    out = dict(
        m=m,
        variant=np.full(B, 4, dtype=np.int32),
        x0=np.stack([q["x0"] for q in ins]),
        u0=np.stack([q["u0"] for q in ins]),
        xref=np.stack([q["xref"] for q in ins]),
        A=np.stack([np.broadcast_to(q["A"], (N + 1, M, 2)) for q in ins]).copy(),
        b=np.stack([np.broadcast_to(q["b"], (N + 1, M)) for q in ins]).copy(),
        Ts=np.full(B, TS),
        term=np.zeros((B, 3)),
    )
    return out


# ---------------------------------------------------------------------------------------------------------------
# Config C3: N = 20, walls + one static box + two moving 3x3 boxes crossing the corridor, lidar
# gated -- gated instances are fixed-time solves (obca_mpc6, fallback obca_mpc8) with time-varying obstacle rows,
# the others free-time solves (obca_mpc4) against the static obstacles only.
def make_instance_c3(i, N=20, seed0=SEED0 + 10 ** 6, sense_dis=10.0):
    rng = np.random.default_rng(seed0 + i)
    om = obstacleModel()
    for _ in range(400):
        box = (rng.uniform(14, 28), rng.uniform(2.5, 7.5), rng.uniform(2, 4), rng.uniform(2, 4))
        y0 = int(round(rng.uniform(3, 7)))
        path = lattice_path(y0, box)
        dyn = []
        for _d in range(2):
            up = rng.uniform() < 0.5
            dyn.append(dict(cx=rng.uniform(10, 35), cy=rng.uniform(-6, -2) if up else rng.uniform(12, 16),
                            th=math.pi / 2 if up else -math.pi / 2, v=rng.uniform(0.1, 0.5)))
        noise = rng.uniform([-0.2, -0.2, -0.1], [0.2, 0.2, 0.1])
        ts_fix = rng.uniform(1.8, 2.5)
        if path is None or path.shape[1] < N + 3:
            continue
        i0 = int(rng.integers(0, path.shape[1] - N - 1))
        x0 = path[:, i0] + noise
        xref = window(path, x0, N)
        dis = (xref[0, N] - x0[0]) + (xref[1, N] - x0[1])
        seg = np.diff(xref[:2], axis=1)
        length = float(np.sum(np.hypot(seg[0], seg[1])))
        if dis / (N * V_MAX * TS) + 1.0 < 1.15 * length / (N * V_MAX * TS):
            continue
        if clearance(x0, box) < DMIN + 0.15 or clearance(xref[:, N], box) < DMIN + 0.15:     # (the expensive test last)
            continue
        rect = rectangle_vertices(box[0], box[1], 0.0, box[2], box[3])
        static = [[[39, 9], [0, 9]], rect, [[0, 1], [39, 1]]]
        front = (x0[0] + EGO[0] * math.cos(x0[2]), x0[1] + EGO[0] * math.sin(x0[2]))
        gated = False
        per_step = []
        for k in range(N + 1):
            polys = list(static)
            for d in dyn:
                cx = d["cx"] + ts_fix * d["v"] * math.cos(d["th"]) * k
                cy = d["cy"] + ts_fix * d["v"] * math.sin(d["th"]) * k
                verts = rectangle_vertices(cx, cy, d["th"], 3.0, 3.0)
                polys.append(verts)
                if k == 0 and any(math.hypot(front[0] - p[0], front[1] - p[1]) <= sense_dis for p in verts[:4]):
                    gated = True
            per_step.append(polys)
        v_all = [len(p) for p in per_step[0]]
        A = np.zeros((N + 1, sum(v_all) - len(v_all), 2))
        b = np.zeros((N + 1, sum(v_all) - len(v_all)))
        for k in range(N + 1):
            Ak, bk = om.obstacle_H_Represent(len(per_step[k]), v_all, per_step[k])
            A[k], b[k] = Ak, bk[:, 0]
        return dict(x0=x0, u0=np.array([0.5, 0.0]), xref=xref, A=A, b=b, m=[n - 1 for n in v_all], gated=gated,
                    Ts_fix=ts_fix, term=np.array([x0[0] + 5.0, 1.0, 9.0]), box=box, dyn=dyn)
    raise RuntimeError("could not draw a C3 instance for seed %d" % (seed0 + i))


def _c3_chunk(args):
    first, count, N = args
    out = []
    for i in range(first, first + count):
        try:
            out.append((i, make_instance_c3(i, N)))
        except RuntimeError:                 # no admissible window for this seed: the seed is skipped
            pass
    return out


def make_batch_c3(B, N=20, first=0, gated=True, procs=1):
    """gated=True: the fixed-time sub-batch (5 obstacles, variant 6); False: the free-time one (3 static, variant 4).
    Instances are the seeds first, first+1, ... whose lidar gate matches, in seed order; procs > 1 draws them in that
    many worker processes (same result -- every instance depends on its own seed only)."""
    ins, i = [], first
    if procs > 1:
        chunk = 64
        while len(ins) < B:
            parts = _spawn_chunks("c3", [(i + j * chunk, chunk) for j in range(procs)], (N,))
            i += procs * chunk
            for part in parts:
                ins += [q for _, q in part if q["gated"] == gated]
        ins = ins[:B]
    while len(ins) < B:
        part = _c3_chunk((i, 1, N))
        i += 1
        if part and part[0][1]["gated"] == gated:
            ins.append(part[0][1])
    if gated:
        m = ins[0]["m"]
        A = np.stack([q["A"] for q in ins])
        b = np.stack([q["b"] for q in ins])
        var, Ts = 6, np.array([q["Ts_fix"] for q in ins])
    else:
        m = ins[0]["m"][:3]
        M = sum(m)
        A = np.stack([q["A"][:, :M] for q in ins])
        b = np.stack([q["b"][:, :M] for q in ins])
        var, Ts = 4, np.full(B, TS)
    return dict(m=m, variant=np.full(B, var, dtype=np.int32), x0=np.stack([q["x0"] for q in ins]),
                u0=np.stack([q["u0"] for q in ins]), xref=np.stack([q["xref"] for q in ins]), A=A, b=b, Ts=Ts,
                term=np.stack([q["term"] for q in ins]))


# ---------------------------------------------------------------------------------------------------------------
# Config C5: closed-loop Monte-Carlo worlds -- C2 corridor with one random box and ``n_dyn`` 3x3
# boxes crossing it; each world is a ``problemSetting`` so that the per-rollout ``closedLoop`` mirror and the
# device-resident ``DeviceRollouts`` consume the same object.
def _one_way_path(y0, box):
    """lattice_path that detours upwards only and stays on the detour row: the reference's time-scale bound is a
    SIGNED sum of the window's x and y extents (src/obca.py:961-962, quirk q3), so its free-time problem is
    infeasible on windows that run towards smaller y -- the reference's own demos only ever climb."""
    path = lattice_path(y0, box)
    if path is None:
        return None
    y = path[1].copy()
    if np.any(np.diff(y) < 0):
        top = int(np.argmax(y))
        if top == 0 or np.any(np.diff(y[:top + 1]) < 0):
            return None                              # detour goes down first
        y[top:] = y[top]
        path = path.copy()
        path[1] = y
        for i in range(path.shape[1] - 1):
            path[2, i] = math.atan2(path[1, i + 1] - path[1, i], path[0, i + 1] - path[0, i])
        path[2, -1] = path[2, -2]
    return path


def make_world_c5(i, n_dyn=2, seed0=SEED0 + 2 * 10 ** 6):
    from .demo_setting import problemSetting
    rng = np.random.default_rng(seed0 + i)
    for _ in range(200):
        box = (rng.uniform(14, 30), rng.uniform(2.5, 7.5), rng.uniform(2, 5), rng.uniform(2, 5))
        y0 = int(round(rng.uniform(3, 7)))
        path = _one_way_path(y0, box)
        dyn = []
        for _d in range(n_dyn):
            up = rng.uniform() < 0.5
            cx, v = rng.uniform(10, 35), rng.uniform(0.1, 0.5)
            th = math.pi / 2 if up else -math.pi / 2
            dyn.append([cx, 0.0 if up else 9.0, th, 3, 3, v, cx, 9.0 if up else 0.0, th, 0, 100])
        if path is None:
            continue
        start = [float(path[0, 0]), float(path[1, 0]), 0.0]
        if clearance(np.array(start), box) < DMIN + 0.1:
            continue
        rect = rectangle_vertices(box[0], box[1], 0.0, box[2], box[3])
        static = [[[39, 9], [0, 9]], rect, [[0, 1], [39, 1]]]
        grid = [[[39, 9], [0, 9], [0, 10], [39, 10]], rect[:4], [[0, 1], [39, 1], [39, 0], [0, 0]]]
        goal = [float(path[0, -1]), float(path[1, -1]), 0.0]
        return problemSetting.from_world((39, 10), start, goal, static, grid, dyn, [[25, 39], [1, 9]], ref_path=path,
                                         name="c5_%d" % i)
    raise RuntimeError("could not draw a C5 world for seed %d" % (seed0 + i))


if __name__ == "__main__":          # worker of make_batch / make_batch_c3 (procs > 1): python -m ...scenarios KIND FIRST COUNT N [..] OUT
    import pickle
    import sys
    if len(sys.argv) == 6 and sys.argv[1] == "c3":
        with open(sys.argv[5], "wb") as f:
            pickle.dump(_c3_chunk((int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]))), f)
    elif len(sys.argv) == 7 and sys.argv[1] == "c2":
        with open(sys.argv[6], "wb") as f:
            pickle.dump(_c2_chunk((int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5]))), f)
