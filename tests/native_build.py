"""Builds the CPU exercisers of the portable device cores (tests/native/*.cpp) -- test infrastructure only."""
import ctypes
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
SRC = os.path.join(HERE, "native", "rollout_host.cpp")
OUT = os.path.join(HERE, "native", "_build", "libnative_host.so")
CSRC = os.path.join(ROOT, "vehicle_motion_planning_with_obstacles_avoidance_using_mpc_amd", "csrc")
DEPS = [SRC, os.path.join(HERE, "native", "lpi_host.cpp"), os.path.join(CSRC, "obca_lpi_core.h"),
        os.path.join(CSRC, "obca_rollout_core.h"), os.path.join(CSRC, "obca_astar_core.h"), os.path.join(CSRC, "obca_device.h"),
        os.path.join(ROOT, "include", "obca_mpc.h")]

_lib = None


def load():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(OUT) or any(os.path.getmtime(d) > os.path.getmtime(OUT) for d in DEPS):
        os.makedirs(os.path.dirname(OUT), exist_ok=True)
        subprocess.run(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off", "-fopenmp", "-Wno-unknown-pragmas",
                        SRC, "-o", OUT], check=True)
    _lib = ctypes.CDLL(OUT)
    _lib.lpi_host_solve_batch.restype = ctypes.c_int
    _lib.lpi_host_solve_batch_cert.restype = ctypes.c_int
    _lib.rollout_host_run.restype = ctypes.c_int
    return _lib


def _ptr(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def lpi_solve(variant, N, m, x0, u0, xref, A, b, Ts, term=None, params=None, cert=False):
    """the structured solver of csrc/obca_lpi_core.h on the CPU; arrays as in include/obca_mpc.h.
    cert=True adds the certificate buffers (final primal vector "z" and multipliers "y", include/obca_mpc.h)"""
    from oracle import c_oracle
    lib = load()
    x0 = np.ascontiguousarray(x0, float)
    B, M = x0.shape[0], int(sum(m))
    var = np.ascontiguousarray(np.broadcast_to(np.asarray(variant, np.int32), (B,)))
    term = np.zeros((B, 3)) if term is None else np.ascontiguousarray(term, float)
    params = params or c_oracle.default_params()
    out = dict(xopt=np.zeros((B, 3, N + 1)), uopt=np.zeros((B, 2, N)), ts_opt=np.zeros(B), status=np.zeros(B, np.int32),
               iters=np.zeros(B, np.int32), info=np.zeros((B, 4)))
    arrs = [x0, np.ascontiguousarray(u0, float), np.ascontiguousarray(xref, float).reshape(B, 3, N + 1),
            np.ascontiguousarray(A, float).reshape(B, N + 1, M, 2), np.ascontiguousarray(b, float).reshape(B, N + 1, M),
            np.ascontiguousarray(np.broadcast_to(np.asarray(Ts, float), (B,)))]
    marr = (ctypes.c_int * len(m))(*[int(v) for v in m])
    extra = []
    fn = lib.lpi_host_solve_batch
    if cert:
        nO = len(m)
        n_max = (N + 1) * (3 + M + 4 * nO) + 2 * N + 1
        R_max = 3 + 3 * N + 3 + 2 * (N + 1) + 4 * N + 2 + 2 * (N + 1) * nO + (N + 1) * (M + 4 * nO)
        out["z"], out["y"] = np.zeros((B, n_max)), np.zeros((B, R_max + 2 * (N + 1) * nO))
        extra = [_ptr(out["z"]), _ptr(out["y"])]
        fn = lib.lpi_host_solve_batch_cert
    rc = fn(N, len(m), marr, _ptr(var), B, *[_ptr(a) for a in arrs], _ptr(term),
            ctypes.byref(params), _ptr(out["xopt"]), _ptr(out["uopt"]), _ptr(out["ts_opt"]),
            _ptr(out["status"]), _ptr(out["iters"]), _ptr(out["info"]), *extra)
    assert rc == 0
    return out


class LpiObca:
    """`obca`-shaped object (reference src/obca.py:828,1361,1564) on the CPU build of the structured core, so that the
    Python ``closedLoop`` mirror can run without a GPU in tests."""

    def __init__(self, engine="lpi"):
        self.calls = []
        self.start_order = "default"       # as the drop-in obca class (…_amd/obca.py): "default" | "x0" | "window" | "zeros"
        self.single_start = False
        self.dodge = True
        self.terminal_screen = True
        self.engine = engine               # "lpi": structured core (csrc/obca_lpi_core.h); "oracle": dense C oracle (oracle/obca_oracle.c)

    def _solve(self, variant, Ts, P, Q, R, N, x0, xL, xU, uL, uU, xref, nObs, vObs, AObs, bObs, dmin, ego, u0, term=None, single_start=False, start_order=None):
        from oracle import c_oracle
        from vehicle_motion_planning_with_obstacles_avoidance_using_mpc_amd.solver import pack_reference_call
        m, x0a, u0a, xr, A, b, ts, tm = pack_reference_call(variant, Ts, N, x0, xref, nObs, vObs, AObs, bObs, u0, term)
        kw = dict(xL=xL[:2], xU=xU[:2], uL=uL, uU=uU, ego=ego, dmin=dmin)
        kw.update(start_order=self.start_order if start_order is None else start_order, single_start=bool(single_start or self.single_start), dodge=self.dodge, terminal_screen=self.terminal_screen)
        if variant == 4:
            kw.update(Qf=Q, Pf=P, R1f=R[0], R2f=R[1])
        else:
            kw.update(Qx=Q, Px=P, R1x=R[0], R2x=R[1])
        engine = lpi_solve if self.engine == "lpi" else c_oracle.solve_batch
        o = engine(variant, N, m, x0a[None], u0a[None], xr[None], A[None], b[None], [ts], tm[None],
                   c_oracle.default_params(**kw))
        self.calls.append(dict(variant=variant, x0=x0a.copy(), u0=u0a.copy(), xref=xr.copy(), A=A.copy(), b=b.copy(), Ts=ts, term=tm.copy(), m=m,
                               status=int(o["status"][0]), info=o["info"][0].copy(), iters=int(o["iters"][0])))
        return o["xopt"][0], o["uopt"][0], bool(o["status"][0] in (0, 1)), float(o["ts_opt"][0])

    def obca_mpc4(self, *a, start_order=None):
        return self._solve(4, *a, start_order=start_order)

    def obca_mpc6(self, *a, single_start=False):
        return self._solve(6, *a[:18], term=a[19], single_start=single_start)

    def obca_mpc8(self, *a):
        return self._solve(8, *a[:18])


def rollout_run(w, N, params, n_steps, max_steps=30, Ts0=0.1, warm_mu=0.0, N_fix=None):
    """csrc/obca_rollout_core.h on the CPU for PackedWorlds ``w``; returns the same dict as DeviceRollouts.read()"""
    from vehicle_motion_planning_with_obstacles_avoidance_using_mpc_amd.rollouts import rollout_dims
    lib = load()
    d = rollout_dims(w, N, max_steps, N_fix=N_fix)
    B, S, N1, nd = w.batch, max_steps, max(N, N_fix or N) + 1, w.n_dyn
    out = {"x_closed": np.zeros((B, S + 1, 3)), "u_closed": np.zeros((B, S, 2)), "T_closed": np.zeros((B, S)),
           "x_openloop": np.zeros((B, S, 3, N1)), "variant": np.zeros((B, S), np.int32), "iters": np.zeros((B, S), np.int32),
           "status": np.zeros((B, S), np.int32), "dyn": np.zeros((B, S, max(nd, 1), 4)), "steps": np.zeros(B, np.int32), "flags": np.zeros(B, np.int32),
           "xref": np.zeros((B, S, 3, N1))}
    dyn = np.ascontiguousarray(w.dyn if nd else np.zeros((B, 1, 13)))
    ins = [np.ascontiguousarray(a) for a in (w.start, w.goal, w.path)] + [np.ascontiguousarray(w.path_len, np.int32)] + \
          [np.ascontiguousarray(w.static_A), np.ascontiguousarray(w.static_b), dyn]
    rc = lib.rollout_host_run(ctypes.byref(d), *[_ptr(a) for a in ins], ctypes.c_double(Ts0), ctypes.c_double(w.sense_dis),
                              ctypes.byref(params), ctypes.c_int(n_steps),
                              *[_ptr(out[k]) for k in ("x_closed", "u_closed", "T_closed", "x_openloop", "variant", "iters",
                                                       "status", "dyn", "steps", "flags", "xref")],
                              ctypes.c_double(warm_mu))
    assert rc == 0, rc
    return out


def harness_rows(w, N, ks, Ts_opt, x0, g, ego=(1.7, 0.75, 1.7, 0.75)):
    """csrc/obca_rollout_core.h on the CPU: the harness part of a step alone for rollout 0 of PackedWorlds ``w``, once per step
    counter in ``ks`` (obstacle advance accumulates), with inherited step length Ts_opt and pose x0; returns what it hands
    the solver of group g after the last one: (variant, A [N_g+1, M_g, 2], b [N_g+1, M_g])"""
    from vehicle_motion_planning_with_obstacles_avoidance_using_mpc_amd.rollouts import rollout_dims
    lib = load()
    d = rollout_dims(w, N, 30)
    nd = w.n_dyn
    Mg = w.static_A.shape[1] + 4 * g
    A, b, var = np.zeros((N + 1, Mg, 2)), np.zeros((N + 1, Mg)), np.zeros(1, np.int32)
    dyn = np.ascontiguousarray(w.dyn[:1] if nd else np.zeros((1, 1, 13)))
    ins = [np.ascontiguousarray(a[:1]) for a in (w.start, w.goal, w.path)] + [np.ascontiguousarray(w.path_len[:1], np.int32)] + \
          [np.ascontiguousarray(w.static_A[:1]), np.ascontiguousarray(w.static_b[:1]), dyn]
    ksa = np.ascontiguousarray(ks, np.int32)
    x0a = None if x0 is None else np.ascontiguousarray(x0, float)
    rc = lib.rollout_host_debug_harness(ctypes.byref(d), *[_ptr(a) for a in ins], ctypes.c_double(w.sense_dis),
                                        _ptr(np.ascontiguousarray(ego, float)), _ptr(ksa), ctypes.c_int(len(ksa)), ctypes.c_double(Ts_opt),
                                        None if x0a is None else _ptr(x0a), ctypes.c_int(g), _ptr(var), _ptr(A), _ptr(b))
    assert rc == 0, rc
    return int(var[0]), A, b


def astar_batch(grids, starts, goals, path_max):
    """csrc/obca_astar_core.h on the CPU: grids [B,rows,cols] (1 = occupied), starts/goals [B,2] (row, col)"""
    from vehicle_motion_planning_with_obstacles_avoidance_using_mpc_amd.planner import yaw_table
    lib = load()
    g = np.ascontiguousarray(grids, np.uint8)
    B, rows, cols = g.shape
    st, go = np.ascontiguousarray(starts, np.int32), np.ascontiguousarray(goals, np.int32)
    path = np.zeros((B, 3, path_max))
    plen = np.zeros(B, np.int32)
    yaw = np.ascontiguousarray(yaw_table())
    lib.astar_host_batch(_ptr(g), B, rows, cols, _ptr(st), _ptr(go), _ptr(yaw), path_max, _ptr(path), _ptr(plen))
    return path, plen
