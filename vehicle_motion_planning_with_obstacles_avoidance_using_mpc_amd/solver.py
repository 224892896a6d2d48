"""Batched OBCA-MPC solver: torch-ROCm tensors in, torch-ROCm tensors out, HIP kernel underneath.

Mirrors the reference's ``obca.obca_mpc4/6/8`` argument meaning (reference src/obca.py:828, :1361, :1564)
for B instances at once.  PyTorch is used for device memory and streams only; the arithmetic is the
hand-written kernel in csrc/obca_kernel.hip reached through the C ABI of include/obca_mpc.h.
"""
import ctypes
import math

import numpy as np


class _LazyTorch:
    """torch is imported on first use: the host-side mirror of the reference's loop (closed_loop.py) needs
    ``pack_reference_call`` from this module, and the worker processes that replay rollouts on the CPU (tests/independent.py)
    should not pay for -- or depend on -- PyTorch"""

    def __getattr__(self, name):
        import torch as t
        globals()["torch"] = t
        return getattr(t, name)


torch = _LazyTorch()

from . import _lib

DEFAULT_EGO = (1.7, 0.75, 1.7, 0.75)      # reference src/closed_loop.py:63


class SolverParams:
    """Per-rollout constants the reference keeps in ``closedLoop.__init__`` (src/closed_loop.py:32-101)."""

    def __init__(self, xL=(0.0, 0.0), xU=(39.0, 10.0), uL=(-0.6, -math.pi / 6), uU=(0.6, math.pi / 6),
                 ego=DEFAULT_EGO, dmin=0.05,
                 Q_free=None, R_free=None, P_free=None, Q_fix=None, R_fix=None, P_fix=None,
                 tol=0.0, rho=0.0, feas_tol=0.0, max_iter_free=0, max_iter_fixed=0, max_soc=0,
                 start_order=0, single_start=False, patience=0, retry_iter=0, dodge=True, terminal_screen=True):
        self.xL, self.xU, self.uL, self.uU = [tuple(float(v) for v in a[:2]) for a in (xL, xU, uL, uU)]
        self.ego = tuple(float(v) for v in ego)
        self.dmin = float(dmin)
        eye3, eye2 = np.eye(3), np.eye(2)
        self.Q_free = 0.1 * eye3 if Q_free is None else np.asarray(Q_free, float)          # closed_loop.py:77
        self.R_free = [0.01 * eye2, 0.1 * eye2] if R_free is None else [np.asarray(r, float) for r in R_free]
        self.P_free = self.Q_free if P_free is None else np.asarray(P_free, float)
        self.Q_fix = 0.001 * eye3 if Q_fix is None else np.asarray(Q_fix, float)           # closed_loop.py:94
        self.R_fix = [0.01 * eye2, 1.0 * eye2] if R_fix is None else [np.asarray(r, float) for r in R_fix]
        self.P_fix = self.Q_fix if P_fix is None else np.asarray(P_fix, float)
        self.tol, self.rho, self.feas_tol = float(tol), float(rho), float(feas_tol)
        self.max_iter_free, self.max_iter_fixed = int(max_iter_free), int(max_iter_fixed)
        self.max_soc = int(max_soc)            # 0 = IPOPT's default (4), negative = no second-order correction
        # the start ladder (include/obca_mpc.h): start_order "default" (window -> x0 -> zeros for every variant;
        # x0 first for a single start and for warm starts) | "x0" | "window" | "zeros" (that start first for every
        # variant) or the OBCA_START_* value; single_start: the first start only; patience / retry_iter: 0 = the defaults
        # 500 + 10 N / 300 + 10 N; dodge: the ladder's last rung for obca_mpc6 / 8 (the window moved to either side); terminal_screen:
        # obca_mpc6 whose terminal set cannot be reached is answered without a solve
        self.start_order = int(_lib.START_ORDERS.get(start_order, start_order))
        self.single_start = bool(single_start)
        self.patience, self.retry_iter = int(patience), int(retry_iter)
        self.dodge, self.terminal_screen = bool(dodge), bool(terminal_screen)

    def to_c(self):
        p = _lib.ObcaParams()
        for dst, (Q, R, P) in ((p.free_time, (self.Q_free, self.R_free, self.P_free)),
                               (p.fixed_time, (self.Q_fix, self.R_fix, self.P_fix))):
            dst.Q[:] = np.asarray(Q, float).reshape(9).tolist()
            dst.P[:] = np.asarray(P, float).reshape(9).tolist()
            dst.R1[:] = np.asarray(R[0], float).reshape(4).tolist()
            dst.R2[:] = np.asarray(R[1], float).reshape(4).tolist()
        p.xL[:], p.xU[:], p.uL[:], p.uU[:] = self.xL, self.xU, self.uL, self.uU
        p.ego[:] = self.ego
        p.dmin = self.dmin
        p.tol, p.rho, p.feas_tol = self.tol, self.rho, self.feas_tol
        p.max_iter_free, p.max_iter_fixed = self.max_iter_free, self.max_iter_fixed
        p.max_soc = self.max_soc
        p.start_order, p.single_start = self.start_order, int(self.single_start)
        p.patience, p.retry_iter = self.patience, self.retry_iter
        p.dodge, p.terminal_screen = (0 if self.dodge else -1), (0 if self.terminal_screen else -1)
        return p


class BatchResult:
    __slots__ = ("xopt", "uopt", "ts_opt", "status", "iters", "info")

    @property
    def feas(self):
        return (self.status == _lib.STATUS_OK) | (self.status == _lib.STATUS_ACCEPTABLE)


class BatchSolver:
    """One handle = one problem shape (N, obstacle edge counts) on one GPU."""

    def __init__(self, N, m, max_batch, device=None, mode=None):
        if not torch.cuda.is_available():
            raise RuntimeError("BatchSolver needs a ROCm GPU; there is no CPU fallback on the product path")
        self.lib = _lib.load()
        self.device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        self.N = int(N)
        self.m = [int(v) for v in m]
        self.M = sum(self.m)
        self.n_obs = len(self.m)
        self.max_batch = int(max_batch)
        d = _lib.ObcaDims()
        d.N, d.n_obs, d.max_batch = self.N, self.n_obs, self.max_batch
        d.device = self.device.index if self.device.index is not None else torch.cuda.current_device()
        for i, v in enumerate(self.m):
            d.m[i] = v
        self._dims = d
        h = ctypes.c_void_p()
        _lib.check(self.lib.obca_create(ctypes.byref(d), ctypes.byref(h)))
        self._h = h
        self.lds_bytes = int(self.lib.obca_lds_bytes(ctypes.byref(d)))
        if mode is not None:
            self.set_mode(mode)

    MODES = {"auto": 0, "wave": 1, "lane": 2, "multiwave": 3, "global": 4, "global1": 5}

    def set_mode(self, mode):
        """'auto' | 'wave' (one wavefront per instance, LDS) | 'multiwave' (four wavefronts per instance, LDS) |
        'lane' (64 instances per wavefront, HBM workspace) | 'global' / 'global1' (four wavefronts / one wavefront per instance,
        row state in an HBM workspace)"""
        _lib.check(self.lib.obca_set_mode(self._h, self.MODES.get(mode, mode)))

    def set_two_sided_sweep(self, on):
        """four-wavefront kernels: Riccati sweep as two concurrent halves (include/obca_mpc.h, obca_set_two_sided_sweep);
        None / -1 = default (only where the one-wavefront kernels cannot run the shape), False / True = never / always"""
        _lib.check(self.lib.obca_set_two_sided_sweep(self._h, -1 if on is None else int(on)))

    def set_shape_specialisation(self, on):
        """use (default) or bypass the compile-time-shape instantiation of the one-wavefront kernel this handle's shape may
        have (include/obca_mpc.h: obca_set_shape_specialisation); results are the same words either way"""
        _lib.check(self.lib.obca_set_shape_specialisation(self._h, 1 if on else 0))

    @property
    def specialised(self):
        """True if launches of this handle run a compile-time-shape instantiation (csrc/obca_device.h: OBCA_SHAPES)"""
        return self.lib.obca_shape_is_specialised(self._h) == 1

    def enable_certificates(self, on=True):
        """Keep the final primal vector and multipliers of every solve (include/obca_mpc.h, obca_set_certificate_buffers):
        after a solve ``self.cert_z[:B]`` is [B, primal_size] and ``self.cert_y[:B]`` is [B, dual_size]."""
        if on:
            nz = int(self.lib.obca_primal_size(ctypes.byref(self._dims)))
            ny = int(self.lib.obca_dual_size(ctypes.byref(self._dims)))
            self.cert_z = torch.zeros(self.max_batch, nz, dtype=torch.float64, device=self.device)
            self.cert_y = torch.zeros(self.max_batch, ny, dtype=torch.float64, device=self.device)
            _lib.check(self.lib.obca_set_certificate_buffers(self._h, ctypes.c_void_p(self.cert_z.data_ptr()),
                                                             ctypes.c_void_p(self.cert_y.data_ptr())))
        else:
            _lib.check(self.lib.obca_set_certificate_buffers(self._h, None, None))
            self.cert_z = self.cert_y = None

    def close(self):
        if getattr(self, "_h", None):
            self.lib.obca_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _dev(self, t, shape, dtype=None):
        t = torch.as_tensor(t, dtype=dtype or torch.float64, device=self.device).contiguous()
        if tuple(t.shape) != tuple(shape):
            raise ValueError("expected shape %s, got %s" % (tuple(shape), tuple(t.shape)))
        return t

    def solve(self, variant, x0, u0, xref, A, b, Ts, term=None, params=None, out=None, want_info=True):
        """variant [B] int (4/6/8) or a single int; other arguments as in include/obca_mpc.h."""
        params = params or SolverParams()
        x0 = torch.as_tensor(x0, dtype=torch.float64, device=self.device)
        B = x0.shape[0]
        N, M = self.N, self.M
        if isinstance(variant, int):
            variant = torch.full((B,), variant, dtype=torch.int32, device=self.device)
        variant = self._dev(variant, (B,), torch.int32)
        x0 = self._dev(x0, (B, 3))
        u0 = self._dev(u0, (B, 2))
        xref = self._dev(xref, (B, 3, N + 1))
        A = self._dev(A, (B, N + 1, M, 2))
        b = self._dev(b, (B, N + 1, M))
        Ts = self._dev(Ts, (B,))
        term = self._dev(torch.zeros(B, 3) if term is None else term, (B, 3))
        if out is None:
            out = BatchResult()
            out.xopt = torch.empty(B, 3, N + 1, dtype=torch.float64, device=self.device)
            out.uopt = torch.empty(B, 2, N, dtype=torch.float64, device=self.device)
            out.ts_opt = torch.empty(B, dtype=torch.float64, device=self.device)
            out.status = torch.empty(B, dtype=torch.int32, device=self.device)
            out.iters = torch.empty(B, dtype=torch.int32, device=self.device)
            out.info = torch.empty(B, 4, dtype=torch.float64, device=self.device) if want_info else None
        cp = params.to_c() if isinstance(params, SolverParams) else params
        stream = torch.cuda.current_stream(self.device).cuda_stream
        ptr = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else None
        _lib.check(self.lib.obca_solve_batch(self._h, ptr(variant), B, ptr(x0), ptr(u0), ptr(xref), ptr(A), ptr(b),
                                             ptr(Ts), ptr(term), ctypes.byref(cp), ptr(out.xopt), ptr(out.uopt),
                                             ptr(out.ts_opt), ptr(out.status), ptr(out.iters), ptr(out.info),
                                             ctypes.c_void_p(stream)))
        self._keep = (variant, x0, u0, xref, A, b, Ts, term)     # alive until the stream has consumed them
        return out


def pack_reference_call(variant, Ts, N, x0, xref, nObs, vObs, AObs, bObs, u0, terminal_set=None):
    """One reference-style call -> the canonical per-instance arrays (numpy).

    Reproduces the reference's row bookkeeping: obca_mpc4 resets its row counter per horizon step and so only
    ever reads the first M rows of AObs (src/obca.py:969), obca_mpc6/8 walk through all (N+1)*M rows
    (src/obca.py:1482, :1677).
    """
    m = [int(vObs[i]) - 1 for i in range(int(nObs))]
    M = sum(m)
    AObs = np.asarray(AObs, float).reshape(-1, 2)
    bObs = np.asarray(bObs, float).reshape(-1)
    A = np.zeros((N + 1, M, 2))
    b = np.zeros((N + 1, M))
    for k in range(N + 1):
        r0 = 0 if variant == 4 else k * M
        A[k] = AObs[r0:r0 + M]
        b[k] = bObs[r0:r0 + M]
    term = np.zeros(3)
    if variant == 6:
        ts = np.asarray(terminal_set, float)
        term[:] = (ts[0, 0], ts[1, 0], ts[1, 1])               # src/obca.py:1465-1466
    xr = np.asarray(xref, float)[:, :N + 1]
    return m, np.asarray(x0, float).reshape(3), np.asarray(u0, float).reshape(2), xr, A, b, float(Ts), term
