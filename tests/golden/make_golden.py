#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ from the reference itself.

Runs ONLY in the build container (needs /root/reference); the GPU box and the
test-suite read the committed *.json files and never this script's imports.

Two families of vectors are captured:

1. ``nlp_eval.json`` -- the reference's own model-building code
   (src/obca.py:828-1071, 1361-1562, 1564-1758) executed with every decision
   variable bound to a number (tests/golden/_numeric_casadi.py), giving the
   objective value and the (lb, value, ub) triple of every constraint, in the
   reference's order, at random points.  Pins the NLP definition.
2. ``harness.json`` -- input/output pairs of the importable host-side harness
   (SURVEY.md section 8a rows S4, S5, H2-H6, F1-F9): rectangle vertices,
   N+1-step prediction, H-representation, reference windows, lidar gate,
   dynamic-obstacle advance, fixed-time preparation, A* routes, and a driver
   trace with a scripted stand-in solver.

No reference source text is stored; only numbers.
"""
import contextlib
import io
import json
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference/src"


def _install_stubs():
    import matplotlib
    matplotlib.use("Agg")
    sys.dont_write_bytecode = True
    sys.path.insert(0, REF)
    sys.path.insert(0, HERE)
    import _numeric_casadi as nc
    cas = types.ModuleType("casadi")
    for k in ("Opti", "MX", "cos", "sin", "pi"):
        setattr(cas, k, getattr(nc, k))
    cas.np = np                       # closed_loop.py gets `np` through `from casadi import *`
    sys.modules["casadi"] = cas
    sk = types.ModuleType("skimage")
    skm = types.ModuleType("skimage.morphology")
    skm.erosion = skm.dilation = skm.disk = None
    sk.morphology = skm
    sys.modules["skimage"] = sk
    sys.modules["skimage.morphology"] = skm
    tt = types.ModuleType("ttictoc")
    tt.tic = lambda: None
    tt.toc = lambda: 0.0
    sys.modules["ttictoc"] = tt
    _size = np.size

    def size(a, axis=None):           # NumPy >= 1.24 rejects ragged lists
        try:
            return _size(a, axis)
        except ValueError:
            assert axis == 0
            return len(a)
    np.size = size
    return nc


def tolist(a):
    if isinstance(a, np.ndarray):
        return a.tolist()
    if isinstance(a, (list, tuple)):
        return [tolist(x) for x in a]
    if isinstance(a, (np.floating, np.integer)):
        return a.item()
    return a


@contextlib.contextmanager
def quiet():
    with contextlib.redirect_stdout(io.StringIO()):
        yield


# ---------------------------------------------------------------------------
# 1. NLP evaluation vectors
# ---------------------------------------------------------------------------

def eval_variant(nc, solver, variant, args, feed):
    nc.Opti.feed = feed
    with quiet():
        getattr(solver, "obca_mpc%d" % variant)(*args)
    o = nc.Opti.last
    cons = [{"kind": c.kind, "val": c.val, "lb": c.lb, "ub": c.ub} for c in o.cons]
    return {"objective": o.objective, "cons": cons}


def random_point(rng, N, xref, rowsA, nObs, variant):
    x = np.asarray(xref)[:, :N + 1] + rng.uniform(-0.7, 0.7, (3, N + 1))
    u = rng.uniform(-0.6, 0.6, (2, N))
    lam = rng.uniform(0.0, 1.0, (rowsA, N + 1))
    mu = rng.uniform(0.0, 1.0, (4 * nObs, N + 1))
    feed = [x, u, lam, mu]
    if variant == 4:
        # the N+1 Topt copies are tied by Topt[k] == Topt[k+1] (src/obca.py:911) and start equal, so every
        # IPM iterate keeps them equal; the restatement collapses them, hence equal values here.
        feed.append(np.full((N + 1, 1), rng.uniform(0.5, 3.0)))
    return feed


def nlp_cases(nc):
    from demo_setting import problemSetting
    from closed_loop import closedLoop
    from model_obstacle import obstacleModel
    rng = np.random.default_rng(20260928)
    out = []

    def add(name, variant, kw, n_points=3):
        order4 = ["Ts", "P", "Q", "R", "N", "x0", "xL", "xU", "uL", "uU", "xref", "nObs", "vObs",
                  "AObs", "bObs", "dmin", "ego", "u0"]
        order = {4: order4, 6: order4 + ["uOpt", "terminal_set"], 8: order4 + ["uOpt"]}[variant]
        args = [kw[k] for k in order]
        from obca import obca
        pts = []
        for _ in range(n_points):
            feed = random_point(rng, kw["N"], kw["xref"], np.shape(kw["AObs"])[0], kw["nObs"], variant)
            r = eval_variant(nc, obca(), variant, args, feed)
            r["point"] = {"x": tolist(feed[0]), "u": tolist(feed[1]), "l": tolist(feed[2]),
                          "mu": tolist(feed[3]), "Topt": tolist(feed[4]) if variant == 4 else None}
            pts.append(r)
        out.append({"name": name, "variant": variant,
                    "inputs": {k: tolist(kw[k]) for k in order}, "points": pts})

    def base_kw(c, N, free=True):
        return dict(Ts=c.Ts, P=c.P_free if free else c.P_fix, Q=c.Q_free if free else c.Q_fix,
                    R=c.R_free if free else c.R_fix, N=N, x0=list(c.x0), xL=c.xL, xU=c.xU,
                    uL=c.uL, uU=c.uU, xref=c.xref, nObs=c.nObs, vObs=[int(v) for v in c.vObs],
                    AObs=c.AObs, bObs=c.bObs, dmin=c.dmin, ego=c.ego, u0=list(c.u0))

    # --- mpc4 on the shipped demos, exactly the first closed-loop call -----
    for demo, N in (("demo1", 6), ("demo9", 5), ("demo8", 5), ("demo1", 5)):
        with quiet():
            c = closedLoop(problemSetting(demo))
            ref = c.update_path(0, c.x0, c.xF, 0, "A_star")
            c.update_obstacle(0, c.Ts_opt)
            c.sensor()
            c.update_obstacle_constraint(N, c.Ts, 0)
            c.xref = c.update_reference_trajectory(N, ref, c.x0)
        add("%s_N%d_mpc4_step0" % (demo, N), 4, base_kw(c, N))

    # --- mpc4 with slanted obstacle, asymmetric ego (offset != 0), full weights, nonzero u0
    with quiet():
        c = closedLoop(problemSetting("demo1"))
        om = obstacleModel()
        st = c.setting
        tri = [[12.0, 2.0], [13.5, 5.5], [17.0, 4.0], [15.0, 1.5], [12.0, 2.0]]  # clockwise quad, no axis-aligned edge
        lobs = [st.static_lObs[0], tri, st.static_lObs[2]]
        N = 4
        full = []
        for _ in range(N + 1):
            full += lobs
        vfull = np.array([len(o) for o in full], dtype=int)
        A, b = om.obstacle_H_Represent(len(full), vfull, full)
    W = rng.uniform(-1, 1, (3, 3))
    Qf = W @ W.T * 0.05 + 0.05 * np.eye(3)
    W2 = rng.uniform(-1, 1, (2, 2))
    R1 = W2 @ W2.T * 0.01 + 0.01 * np.eye(2)
    W3 = rng.uniform(-1, 1, (2, 2))
    R2 = W3 @ W3.T * 0.1 + 0.05 * np.eye(2)
    W4 = rng.uniform(-1, 1, (3, 3))
    Pf = W4 @ W4.T * 0.1 + 0.02 * np.eye(3)
    xref = np.array([[4.0, 5.0, 6.0, 7.0, 8.0], [6.0, 6.0, 6.5, 7.0, 7.0], [0.0, 0.3, 0.4, 0.2, 0.0]])
    kw = dict(Ts=0.13, P=Pf, Q=Qf, R=[R1, R2], N=N, x0=[3.9, 6.1, -0.05], xL=[0, 0, -np.pi], xU=[39, 10, np.pi],
              uL=[-0.6, -np.pi / 6], uU=[0.6, np.pi / 6], xref=xref, nObs=3, vObs=[2, 5, 2], AObs=A, bObs=b,
              dmin=0.07, ego=[2.5, 0.8, 1.0, 0.7], u0=[0.3, -0.05])
    add("slanted_asym_mpc4", 4, kw)

    # --- mpc6 / mpc8: demo1 with the moving box advanced (time-varying rows) ----
    with quiet():
        c = closedLoop(problemSetting("demo1"))
        for k in range(9):
            c.update_obstacle(k, 2.0378864)
        c.x0 = np.array([17.0, 6.0, 0.0])
        c.sensor()
        c.Ts_opt = 2.0378864
        c.N_fix = 6
        c.update_obstacle_constraint(c.N_fix, c.Ts_opt, 1)
    xref = np.array([[17.0 + i for i in range(7)], [6.0] * 7, [0.0] * 7])
    c.xref = xref
    c.u0 = [0.6, 0.0]
    kw = base_kw(c, 6, free=False)
    kw["Ts"] = 2.0378864
    kw["uOpt"] = np.zeros((2, 6))
    kw["terminal_set"] = np.array([[22.0, 99.0], [1.0, 9.0]])
    assert np.shape(kw["AObs"]) == (70, 2) and kw["nObs"] == 4, (np.shape(kw["AObs"]), kw["nObs"])
    add("demo1_dyn_mpc6", 6, kw)
    add("demo1_dyn_mpc8", 8, kw)

    # --- mpc6 with full weights / asymmetric ego / slanted rows that change with k
    full = []
    N = 3
    for k in range(N + 1):
        moved = [[p[0] + 0.3 * k, p[1] - 0.2 * k] for p in tri]
        full += [st.static_lObs[0], moved, st.static_lObs[2]]
    vfull = np.array([len(o) for o in full], dtype=int)
    A, b = om.obstacle_H_Represent(len(full), vfull, full)
    kw = dict(Ts=1.7, P=Pf, Q=Qf, R=[R1, R2], N=N, x0=[3.9, 6.1, -0.05], xL=[0, 0, -np.pi], xU=[39, 10, np.pi],
              uL=[-0.6, -np.pi / 6], uU=[0.6, np.pi / 6], xref=xref[:, :N + 1] if False else
              np.array([[4.0, 5.0, 6.0, 7.0], [6.0, 6.0, 6.5, 7.0], [0.0, 0.3, 0.4, 0.2]]),
              nObs=3, vObs=[2, 5, 2], AObs=A, bObs=b, dmin=0.07, ego=[2.5, 0.8, 1.0, 0.7], u0=[0.3, -0.05],
              uOpt=np.zeros((2, N)), terminal_set=np.array([[5.5, 99.0], [2.0, 8.5]]))
    add("slanted_asym_mpc6", 6, kw)
    add("slanted_asym_mpc8", 8, kw)
    return out


# ---------------------------------------------------------------------------
# 2. Harness vectors
# ---------------------------------------------------------------------------

def harness_cases():
    from demo_setting import problemSetting
    from closed_loop import closedLoop
    from model_obstacle import obstacleModel
    out = {}

    # F1 rectangle -> vertices
    with quiet():
        st = problemSetting("demo1")
    f1 = []
    for rect in ([22.5, 0.0, np.pi / 2, 3, 3], [8, 50, -np.pi / 2, 2, 2], [99, 5, -np.pi, 3, 3],
                 [10.25, 3.5, 0.0, 4, 2], [5.0, 5.0, 0.3, 4.0, 1.5], [7.0, 2.0, -2.1, 1.0, 3.0]):
        f1.append({"rect": tolist(rect), "vertices": tolist(st.get_obstacle(*rect))})
    out["F1_get_obstacle"] = f1

    # F2/F3 prediction + H-rep for the shipped demos (static, and with dynamic obstacles)
    f23 = []
    for demo in ("demo1", "demo8", "demo9", "demo10"):
        for N, Ts, dyn in ((5, 0.1, 0), (6, 0.1, 0), (6, 2.0378864, 1), (5, 1.3, 1)):
            with quiet():
                st = problemSetting(demo)
                c = closedLoop(st)
                c.update_obstacle(0, Ts)
                c.update_obstacle(1, Ts)
                c.update_obstacle_constraint(N, Ts, dyn)
            f23.append({"demo": demo, "N": N, "Ts": Ts, "dynObs_exist": dyn,
                        "static_lObs": tolist(st.static_lObs), "dyn_lObs": tolist(st.dyn_lObs),
                        "dyn_obs_info": tolist(st.dyn_obs_info),
                        "nObs": int(c.nObs), "vObs": [int(v) for v in c.vObs],
                        "lObs": tolist(c.lObs), "AObs": tolist(c.AObs), "bObs": tolist(c.bObs)})
    out["F2F3_predict_hrep"] = f23

    # F3b H-rep branch coverage on hand-made polygons (general slopes both directions)
    om = obstacleModel()
    polys = [[[12.0, 2.0], [13.5, 5.5], [17.0, 4.0], [15.0, 1.5], [12.0, 2.0]],
             [[0.0, 0.0], [0.0, 2.0], [2.0, 2.0], [2.0, 0.0], [0.0, 0.0]],
             [[1.0, 1.0], [2.0, 3.0], [3.0, 1.0], [1.0, 1.0]],
             [[39, 9], [0, 9]], [[0, 1], [39, 1]], [[8, 0], [8, 6], [40, 6]]]
    v = np.array([len(p) for p in polys], dtype=int)
    A, b = om.obstacle_H_Represent(len(polys), v, polys)
    out["F3b_hrep_polys"] = {"polys": polys, "A": tolist(A), "b": tolist(b)}

    # F9 A* + F4 windows
    f9, f4 = [], []
    for demo in ("demo1", "demo8", "demo9", "demo10"):
        with quiet():
            st = problemSetting(demo)
            c = closedLoop(st)
            grid = np.array(st.org_gridMap)
            ref = c.update_path(0, c.x0, c.xF, 0, "A_star")
        f9.append({"demo": demo, "grid": tolist(grid.astype(int)), "start": tolist(st.startPose),
                   "goal": tolist(st.goalPose), "static_gridlObs": tolist(st.static_gridlObs),
                   "map_size": tolist(st.map_size), "ref": tolist(ref)})
        P = ref.shape[1]
        poses = [list(c.x0), [ref[0, P // 2] + 0.3, ref[1, P // 2] - 0.2, 0.1],
                 [ref[0, P - 3] + 0.1, ref[1, P - 3], 0.0], [ref[0, P - 1], ref[1, P - 1] + 0.4, 0.0],
                 [ref[0, 3] + 0.49, ref[1, 3] + 0.49, 0.0]]
        for N in (5, 6):
            for pose in poses:
                with quiet():
                    w = c.update_reference_trajectory(N, ref, pose)
                f4.append({"demo": demo, "N": N, "pose": tolist(pose), "window": tolist(w)})
    out["F9_astar"] = f9
    out["F4_windows"] = f4

    # F6 dynamic-obstacle advance + F5 lidar gate along scripted poses
    f56 = []
    for demo, poses in (("demo1", [[3, 4, 0], [8, 4.5, 0.3], [12.6, 6.0, 0.2], [14, 6, 0.0], [20, 6, -0.2]]),
                        ("demo8", [[3, 4, 0], [5, 4, 0], [9, 4, 0.1], [15, 4, 0], [21, 4, 0]]),
                        ("demo9", [[1, 5, 0], [3, 7, 0.9], [6, 12, 1.2], [7, 30, 1.5], [8, 42, 1.57]])):
        with quiet():
            st = problemSetting(demo)
            c = closedLoop(st)
        steps = []
        for k, pose in enumerate(poses):
            ts_opt = 0.1 if k == 0 else 1.5 + 0.25 * k
            with quiet():
                c.x0 = np.array(pose, dtype=float)
                c.update_obstacle(k, ts_opt)
                verts = [[list(p) for p in o[:5]] for o in c.dyn_loc[-1]]
                c.sensor()
            flags = [int(o[5]) for o in c.dyn_loc[-1]]
            steps.append({"k": k, "Ts_opt": ts_opt, "pose": tolist(pose), "ego": tolist(c.ego),
                          "senseDis": st.senseDis, "dyn_table": tolist(c.dyn_orignal_info),
                          "dyn_vertices": tolist(verts), "flags": flags, "fixtime": int(c.fixtime),
                          "sensed_info": tolist(st.dyn_obs_info), "dyn_nObs": int(st.dyn_nObs)})
        f56.append({"demo": demo, "dyn_table0": tolist(problemSettingTable(demo)), "steps": steps})
    out["F5F6_sensor_advance"] = f56

    # F7 fixed-time preparation (src/closed_loop.py:360-374)
    f7 = []
    for demo, N in (("demo1", 6), ("demo9", 5), ("demo8", 5)):
        with quiet():
            st = problemSetting(demo)
            c = closedLoop(st)
            ref = c.update_path(0, c.x0, c.xF, 0, "A_star")
        c.N_free = N
        c.N_fix = N
        P = ref.shape[1]
        i0 = 4
        x0 = np.array([ref[0, i0] + 0.2, ref[1, i0] + 0.1, ref[2, i0]])
        xprev = np.array(ref[:, i0:i0 + N + 1], dtype=float) + 0.05
        c.x0 = x0
        c.xOpt = xprev.copy()
        c.Ts_opt = 1.9
        c.Ts = 0.1
        with quiet():
            c.xref = c.update_reference_trajectory(c.N_fix, ref, c.x0)
            win = c.xref.copy()
            for i in range(c.N_fix - 5):
                c.xref[:, i] = c.xOpt[:, i + 1]
            c.xref = c.update_path(0, 0, 0, allAviable=1, type="")
            term = np.array([[c.x0[0] + 5, 99], [1, 9]])
        f7.append({"demo": demo, "N_free": N, "N_fix_in": N, "ref": tolist(ref), "x0": tolist(x0),
                   "xOpt_prev": tolist(xprev), "Ts_opt_in": 1.9, "window": tolist(win),
                   "xref": tolist(c.xref), "N_fix": int(c.N_fix), "Ts_opt": float(c.Ts_opt),
                   "Ts": float(c.Ts), "terminal_set": tolist(term)})
    out["F7_fixtime_prep"] = f7

    # F8 driver trace with a scripted stand-in solver (follows xref exactly, Ts_opt = 2.0)
    f8 = []
    for demo, N in (("demo1", 6), ("demo8", 6), ("demo9", 5)):
        with quiet():
            st = problemSetting(demo)
            c = closedLoop(st)
        c.N_free = N
        c.N_fix = N
        calls = []

        class Fake:
            def _ret(self, name, Ts, N_, x0, xref, nObs, vObs, AObs, bObs, u0, free, term=None):
                calls.append({"variant": name, "Ts": float(Ts), "N": int(N_), "x0": tolist(np.asarray(x0, float)),
                              "u0": tolist(np.asarray(u0, float)), "xref": tolist(np.asarray(xref, float)),
                              "nObs": int(nObs), "vObs": [int(v) for v in vObs],
                              "AObs_shape": list(np.shape(AObs)), "bObs_sum": float(np.sum(bObs)),
                              "AObs_first": tolist(np.asarray(AObs)[:int(sum(vObs[:nObs]) - nObs)]),
                              "terminal_set": tolist(term) if term is not None else None})
                xo = np.array(xref, dtype=float)[:, :N_ + 1].copy()
                xo[:, 0] = np.asarray(x0, float)
                uo = np.tile(np.array([[0.5], [0.01]]), (1, N_))
                return xo, uo, True, (2.0 if free else Ts)

            def obca_mpc4(self, Ts, P, Q, R, N_, x0, xL, xU, uL, uU, xref, nObs, vObs, AObs, bObs, dmin, ego, u0):
                return self._ret("mpc4", Ts, N_, x0, xref, nObs, vObs, AObs, bObs, u0, True)

            def obca_mpc6(self, Ts, P, Q, R, N_, x0, xL, xU, uL, uU, xref, nObs, vObs, AObs, bObs, dmin, ego, u0,
                          uOpt, terminal_set):
                return self._ret("mpc6", Ts, N_, x0, xref, nObs, vObs, AObs, bObs, u0, False, terminal_set)

            def obca_mpc8(self, *a):
                raise AssertionError("mpc8 not expected in the scripted trace")

        c.obca_solver = Fake()
        c.draw.fullDimension_closedLoop_animate = lambda *a, **k: None
        with quiet():
            c.closed_loop_mpc4()
        f8.append({"demo": demo, "N": N, "calls": calls, "x_closed": tolist(np.asarray(c.xOpt)),
                   "Ts_opt_list": tolist(c.Ts_opt)})
    out["F8_driver_trace"] = f8
    return out


def problemSettingTable(demo):
    from demo_setting import problemSetting
    with quiet():
        return [list(r) for r in problemSetting(demo).dyn_obs_info]


def main():
    nc = _install_stubs()
    nlp = nlp_cases(nc)
    with open(os.path.join(HERE, "nlp_eval.json"), "w") as f:
        json.dump(nlp, f)
    har = harness_cases()
    with open(os.path.join(HERE, "harness.json"), "w") as f:
        json.dump(har, f)
    print("nlp cases:", [(c["name"], len(c["points"][0]["cons"])) for c in nlp])
    print("harness keys:", {k: len(v) for k, v in har.items()})


if __name__ == "__main__":
    main()
