"""ORACLE (test infrastructure): ctypes wrapper of oracle/obca_oracle.c -- the plain-C restatement used as the
checker at larger batch sizes and as bench.py's cpu_baseline.  Host (numpy) arrays in and out."""
import ctypes
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.path.join(HERE, "_build", "libobca_oracle.so")


class OracleParams(ctypes.Structure):
    _fields_ = [(k, ctypes.c_double * n) for k, n in (("Qf", 9), ("Pf", 9), ("R1f", 4), ("R2f", 4), ("Qx", 9), ("Px", 9),
                                                      ("R1x", 4), ("R2x", 4), ("xL", 2), ("xU", 2), ("uL", 2), ("uU", 2),
                                                      ("ego", 4))] + \
               [("dmin", ctypes.c_double), ("tol", ctypes.c_double), ("rho", ctypes.c_double),
                ("feas_tol", ctypes.c_double), ("max_iter_free", ctypes.c_int), ("max_iter_fixed", ctypes.c_int),
                ("max_soc", ctypes.c_int),          # 0 = IPOPT's default (4 second-order-correction trials), < 0 = off
                # as obca_params (include/obca_mpc.h): the start ladder
                ("start_order", ctypes.c_int), ("single_start", ctypes.c_int), ("patience", ctypes.c_int), ("retry_iter", ctypes.c_int),
                # 0 = the default (on), negative = off: the dodge rung of the ladder, the closed-form terminal-set screen of obca_mpc6
                ("dodge", ctypes.c_int), ("terminal_screen", ctypes.c_int)]

START_ORDERS = {"default": 0, "x0": 3, "window": 1, "zeros": 2}


_lib = None


def build():
    subprocess.run(["make", "-s", "-C", HERE], check=True)


def available():
    try:
        load()
        return True
    except Exception:
        return False


def load():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB):
            build()
        _lib = ctypes.CDLL(LIB)
        _lib.obca_oracle_solve_batch.restype = ctypes.c_int
    return _lib


def default_params(**kw):
    p = OracleParams()
    e3, e2 = np.eye(3), np.eye(2)
    vals = dict(Qf=0.1 * e3, Pf=0.1 * e3, R1f=0.01 * e2, R2f=0.1 * e2, Qx=0.001 * e3, Px=0.001 * e3, R1x=0.01 * e2,
                R2x=1.0 * e2, xL=[0, 0], xU=[39, 10], uL=[-0.6, -np.pi / 6], uU=[0.6, np.pi / 6],
                ego=[1.7, 0.75, 1.7, 0.75])
    vals.update({k: v for k, v in kw.items() if k in vals})
    for k, v in vals.items():
        getattr(p, k)[:] = np.asarray(v, float).reshape(-1).tolist()
    p.dmin = float(kw.get("dmin", 0.05))
    p.tol = float(kw.get("tol", 0.0))
    p.rho = float(kw.get("rho", 0.0))
    p.feas_tol = float(kw.get("feas_tol", 0.0))
    p.max_iter_free = int(kw.get("max_iter_free", 0))
    p.max_iter_fixed = int(kw.get("max_iter_fixed", 0))
    p.max_soc = int(kw.get("max_soc", 0))
    order = kw.get("start_order", 0)
    p.start_order = int(START_ORDERS.get(order, order))
    ss = kw.get("single_start", 0)
    p.single_start = int(ss) if isinstance(ss, (int, np.integer)) and not isinstance(ss, bool) else int(bool(ss))     # (an out-of-range integer reaches the library as it is)
    p.patience = int(kw.get("patience", 0))
    p.retry_iter = int(kw.get("retry_iter", 0))
    p.dodge = 0 if kw.get("dodge", True) else -1
    p.terminal_screen = 0 if kw.get("terminal_screen", True) else -1
    return p


def solve_batch(variant, N, m, x0, u0, xref, A, b, Ts, term=None, params=None, threads=1):
    lib = load()
    x0 = np.ascontiguousarray(x0, float)
    B = x0.shape[0]
    M = int(sum(m))
    var = np.ascontiguousarray(np.broadcast_to(np.asarray(variant, np.int32), (B,)))
    u0 = np.ascontiguousarray(u0, float)
    xref = np.ascontiguousarray(xref, float).reshape(B, 3, N + 1)
    A = np.ascontiguousarray(A, float).reshape(B, N + 1, M, 2)
    b = np.ascontiguousarray(b, float).reshape(B, N + 1, M)
    Ts = np.ascontiguousarray(np.broadcast_to(np.asarray(Ts, float), (B,)))
    term = np.zeros((B, 3)) if term is None else np.ascontiguousarray(term, float)
    params = params or default_params()
    marr = (ctypes.c_int * len(m))(*[int(v) for v in m])
    out = dict(xopt=np.zeros((B, 3, N + 1)), uopt=np.zeros((B, 2, N)), ts_opt=np.zeros(B),
               status=np.zeros(B, np.int32), iters=np.zeros(B, np.int32), info=np.zeros((B, 4)))
    ptr = lambda a: a.ctypes.data_as(ctypes.c_void_p)
    rc = lib.obca_oracle_solve_batch(ctypes.c_int(N), ctypes.c_int(len(m)), marr, ptr(var), ctypes.c_int(B), ptr(x0), ptr(u0),
                                ptr(xref), ptr(A), ptr(b), ptr(Ts), ptr(term), ctypes.byref(params), ptr(out["xopt"]),
                                ptr(out["uopt"]), ptr(out["ts_opt"]), ptr(out["status"]), ptr(out["iters"]),
                                ptr(out["info"]), ctypes.c_int(threads))
    if rc != 0:
        raise ValueError("obca_oracle_solve_batch: invalid start options (return code %d)" % rc)
    return out
