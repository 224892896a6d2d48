"""ORACLE (test infrastructure, never on the product path): numpy restatement of the
OBCA NLP that the reference builds in CasADi.

Follows /root/reference/src/obca.py:828-1071 (obca_mpc4, free time),
:1361-1562 (obca_mpc6, fixed time + terminal set) and :1564-1758 (obca_mpc8,
fixed time, no terminal set).  Only the *effective* variables are kept
(SURVEY.md Appendix A.1): the reference allocates ``l`` with rows(AObs) rows but
only M = sum(vObs[i]-1) of them enter any constraint besides ``l >= 0``; the
N+1 copies of Topt are tied by equalities and are collapsed into one scalar T
whose inequality constraints keep multiplicity N+1.

Pinned by tests/test_oracle_nlp.py against tests/golden/nlp_eval.json, which was
produced by running the reference's own model-building code on numbers
(tests/golden/make_golden.py).  Solver parity with IPOPT itself rests on ONE reference-held output (the demo9 GIF,
tests/test_reference_gif.py) and is UNPINNED beyond it (casadi/IPOPT cannot run in the build container) -- see DESIGN.md.

Variable layout (stage major):
    for k = 0..N:  p_k = (x, y, theta)      3
                   u_k = (v, omega)         2   (k < N only)
                   lam_k                    M
                   mu_k                     4*nObs
    T                                       1   (variant 4 only)
"""
import math

import numpy as np

INF = float("inf")
ACC_MAX = (0.6, math.pi / 6)          # src/obca.py:932-933 (hard-coded in the reference)
T_MIN = 1e-4                          # src/obca.py:963


class Problem:
    """One OBCA NLP instance in canonical (batch-kernel) form."""

    def __init__(self, variant, N, m, x0, u0, xref, A, b, Ts, Q, R1, R2, P, xL, xU, uL, uU, ego, dmin,
                 term=None):
        self.variant = int(variant)
        assert self.variant in (4, 6, 8)
        self.N = int(N)
        self.m = [int(v) for v in m]
        self.nObs = len(self.m)
        self.M = sum(self.m)
        self.off_m = np.concatenate([[0], np.cumsum(self.m)]).astype(int)
        self.x0 = np.asarray(x0, float).reshape(3)
        self.u0 = np.asarray(u0, float).reshape(2)
        self.xref = np.asarray(xref, float)[:, :self.N + 1].copy()
        self.A = np.asarray(A, float).reshape(self.N + 1, self.M, 2)     # per-step rows
        self.b = np.asarray(b, float).reshape(self.N + 1, self.M)
        self.Ts = float(Ts)
        sym = lambda X: 0.5 * (np.asarray(X, float) + np.asarray(X, float).T)
        self.Q, self.R1, self.R2, self.P = sym(Q), sym(R1), sym(R2), sym(P)
        self.xL, self.xU = np.asarray(xL, float)[:2], np.asarray(xU, float)[:2]
        self.uL, self.uU = np.asarray(uL, float)[:2], np.asarray(uU, float)[:2]
        ego = np.asarray(ego, float)
        L, W = ego[0] + ego[2], ego[1] + ego[3]
        self.g = np.array([L / 2, W / 2, L / 2, W / 2])                 # src/obca.py:1018-1024
        self.off = (ego[0] + ego[2]) / 2 - ego[2]                        # src/obca.py:1026
        self.dmin = float(dmin)
        self.term = None if term is None else np.asarray(term, float).reshape(3)  # xmin, ymin, ymax
        if self.variant == 4:
            dis = (self.xref[0, self.N] - self.x0[0]) + (self.xref[1, self.N] - self.x0[1])
            self.Tmax = dis / (self.N * self.uU[0] * self.Ts) + 1.0     # src/obca.py:961-962 (signed sum)
        # layout
        self.ns = 3 + 2 + self.M + 4 * self.nObs                         # full stage width (k < N)
        self.n = (self.N + 1) * (3 + self.M + 4 * self.nObs) + 2 * self.N + (1 if self.variant == 4 else 0)

    # -- construction from the reference calling convention ----------------
    @staticmethod
    def from_reference_args(variant, Ts, P, Q, R, N, x0, xL, xU, uL, uU, xref, nObs, vObs, AObs, bObs, dmin,
                            ego, u0, uOpt=None, terminal_set=None):
        m = [int(vObs[i]) - 1 for i in range(int(nObs))]
        M = sum(m)
        AObs = np.asarray(AObs, float).reshape(-1, 2)
        bObs = np.asarray(bObs, float).reshape(-1)
        A = np.zeros((N + 1, M, 2))
        b = np.zeros((N + 1, M))
        for k in range(N + 1):
            r0 = 0 if variant == 4 else k * M      # q5: mpc4 resets the row counter per k (src/obca.py:969)
            A[k] = AObs[r0:r0 + M]
            b[k] = bObs[r0:r0 + M]
        term = None
        if variant == 6:
            ts = np.asarray(terminal_set, float)
            term = [ts[0, 0], ts[1, 0], ts[1, 1]]                        # src/obca.py:1465-1466
        return Problem(variant, N, m, x0, u0, xref, A, b, Ts, Q, R[0], R[1], P, xL, xU, uL, uU, ego, dmin, term)

    # -- indexing -----------------------------------------------------------
    def ip(self, k):
        return k * self.ns if k < self.N else self.N * self.ns

    def iu(self, k):
        assert k < self.N
        return k * self.ns + 3

    def il(self, k):
        return self.ip(k) + (5 if k < self.N else 3)

    def imu(self, k):
        return self.il(k) + self.M

    def iT(self):
        assert self.variant == 4
        return self.n - 1

    def start_point(self):
        """Opti default initial guess is 0; only Topt is set to 1 (src/obca.py:856)."""
        z = np.zeros(self.n)
        if self.variant == 4:
            z[self.iT()] = 1.0
        return z

    def pack(self, x, u, lam, mu, T=None):
        z = np.zeros(self.n)
        for k in range(self.N + 1):
            z[self.ip(k):self.ip(k) + 3] = np.asarray(x)[:, k]
            if k < self.N:
                z[self.iu(k):self.iu(k) + 2] = np.asarray(u)[:, k]
            z[self.il(k):self.il(k) + self.M] = np.asarray(lam)[:, k]
            z[self.imu(k):self.imu(k) + 4 * self.nObs] = np.asarray(mu)[:, k]
        if self.variant == 4:
            z[self.iT()] = T
        return z

    def unpack_xu(self, z):
        x = np.stack([z[self.ip(k):self.ip(k) + 3] for k in range(self.N + 1)], axis=1)
        u = np.stack([z[self.iu(k):self.iu(k) + 2] for k in range(self.N)], axis=1)
        return x, u

    def h(self, z):
        return z[self.iT()] * self.Ts if self.variant == 4 else self.Ts

    # -- objective -----------------------------------------------------------
    def objective(self, z, grad=False, hess=False):
        N = self.N
        f = 0.0
        g = np.zeros(self.n)
        H = np.zeros((self.n, self.n)) if hess else None
        h = self.h(z)
        free = self.variant == 4
        T = z[self.iT()] if free else 1.0
        for t in range(N):
            i = self.ip(t)
            e = z[i:i + 3] - self.xref[:, t]
            f += e @ self.Q @ e
            g[i:i + 3] += 2 * self.Q @ e
            j = self.iu(t)
            ut = z[j:j + 2]
            f += ut @ self.R1 @ ut
            g[j:j + 2] += 2 * self.R1 @ ut
            if hess:
                H[i:i + 3, i:i + 3] += 2 * self.Q
                H[j:j + 2, j:j + 2] += 2 * self.R1
            if t < N - 1:
                j2 = self.iu(t + 1)
                q = z[j2:j2 + 2] - ut
                Rq = self.R2 @ q
                qq = q @ Rq
                f += qq / h ** 2
                g[j2:j2 + 2] += 2 * Rq / h ** 2
                g[j:j + 2] -= 2 * Rq / h ** 2
                if free:
                    g[self.iT()] += -2 * qq / (h ** 2 * T)
                if hess:
                    B = 2 * self.R2 / h ** 2
                    H[j2:j2 + 2, j2:j2 + 2] += B
                    H[j:j + 2, j:j + 2] += B
                    H[j2:j2 + 2, j:j + 2] -= B
                    H[j:j + 2, j2:j2 + 2] -= B
                    if free:
                        iT = self.iT()
                        c = 4 * Rq / (h ** 2 * T)
                        H[j2:j2 + 2, iT] -= c
                        H[iT, j2:j2 + 2] -= c
                        H[j:j + 2, iT] += c
                        H[iT, j:j + 2] += c
                        H[iT, iT] += 6 * qq / (h ** 2 * T ** 2)
        i = self.ip(N)
        e = z[i:i + 3] - self.xref[:, N]
        f += e @ self.P @ e
        g[i:i + 3] += 2 * self.P @ e
        if hess:
            H[i:i + 3, i:i + 3] += 2 * self.P
        if free:
            f += (N + 1) * (10 * T + T ** 2)                               # src/obca.py:887-888
            g[self.iT()] += (N + 1) * (10 + 2 * T)
            if hess:
                H[self.iT(), self.iT()] += 2 * (N + 1)
        out = [f]
        if grad:
            out.append(g)
        if hess:
            out.append(H)
        return out[0] if len(out) == 1 else tuple(out)

    # -- equality constraints ---------------------------------------------
    def eq_layout(self):
        """names of the equality rows, canonical order"""
        rows = [("init", 0, j) for j in range(3)]
        for k in range(self.N):
            rows += [("dyn", k, j) for j in range(3)]
        if self.variant == 4:
            rows += [("term", self.N, j) for j in range(3)]
        for k in range(self.N + 1):
            for i in range(self.nObs):
                rows += [("rot", k, i, 0), ("rot", k, i, 1)]
        return rows

    def _obst(self, z, k, i):
        o0, o1 = self.off_m[i], self.off_m[i + 1]
        lam = z[self.il(k) + o0:self.il(k) + o1]
        mu = z[self.imu(k) + 4 * i:self.imu(k) + 4 * i + 4]
        A = self.A[k, o0:o1]
        b = self.b[k, o0:o1]
        c = A.T @ lam
        return lam, mu, A, b, c

    def eq(self, z, jac=False, hess_y=None):
        """c(z) = 0.  Optional dense Jacobian; hess_y adds sum_i y_i * Hess c_i to a returned matrix."""
        N, n = self.N, self.n
        rows = self.eq_layout()
        c = np.zeros(len(rows))
        J = np.zeros((len(rows), n)) if jac else None
        H = np.zeros((n, n)) if hess_y is not None else None
        free = self.variant == 4
        h = self.h(z)
        r = 0
        c[0:3] = z[0:3] - self.x0
        if jac:
            J[0:3, 0:3] = np.eye(3)
        r = 3
        for k in range(N):
            i, j, i2 = self.ip(k), self.iu(k), self.ip(k + 1)
            th = z[i + 2]
            v, w = z[j], z[j + 1]
            ct, st = math.cos(th), math.sin(th)
            c[r + 0] = z[i2 + 0] - z[i + 0] - h * v * ct
            c[r + 1] = z[i2 + 1] - z[i + 1] - h * v * st
            c[r + 2] = z[i2 + 2] - z[i + 2] - h * w
            if jac:
                J[r:r + 3, i2:i2 + 3] += np.eye(3)
                J[r:r + 3, i:i + 3] -= np.eye(3)
                J[r + 0, i + 2] += h * v * st
                J[r + 1, i + 2] -= h * v * ct
                J[r + 0, j] -= h * ct
                J[r + 1, j] -= h * st
                J[r + 2, j + 1] -= h
                if free:
                    iT = self.iT()
                    J[r + 0, iT] -= self.Ts * v * ct
                    J[r + 1, iT] -= self.Ts * v * st
                    J[r + 2, iT] -= self.Ts * w
            if H is not None:
                px, py, pt = hess_y[r:r + 3]
                a = h * v * (px * ct + py * st)
                H[i + 2, i + 2] += a
                bq = h * (px * st - py * ct)
                H[i + 2, j] += bq
                H[j, i + 2] += bq
                if free:
                    iT = self.iT()
                    d = -self.Ts * v * (-px * st + py * ct)
                    H[i + 2, iT] += d
                    H[iT, i + 2] += d
                    e = -self.Ts * (px * ct + py * st)
                    H[j, iT] += e
                    H[iT, j] += e
                    H[j + 1, iT] += -self.Ts * pt
                    H[iT, j + 1] += -self.Ts * pt
            r += 3
        if free:
            i = self.ip(N)
            c[r:r + 3] = z[i:i + 3] - self.xref[:, N]
            if jac:
                J[r:r + 3, i:i + 3] = np.eye(3)
            r += 3
        for k in range(N + 1):
            ipk = self.ip(k)
            th = z[ipk + 2]
            ct, st = math.cos(th), math.sin(th)
            for i in range(self.nObs):
                lam, mu, A, b, cc = self._obst(z, k, i)
                o0, o1 = self.off_m[i], self.off_m[i + 1]
                il = self.il(k) + o0
                im = self.imu(k) + 4 * i
                c[r] = mu[0] - mu[2] + ct * cc[0] + st * cc[1]
                c[r + 1] = mu[1] - mu[3] - st * cc[0] + ct * cc[1]
                if jac:
                    J[r, il:il + (o1 - o0)] = ct * A[:, 0] + st * A[:, 1]
                    J[r + 1, il:il + (o1 - o0)] = -st * A[:, 0] + ct * A[:, 1]
                    J[r, im + 0], J[r, im + 2] = 1.0, -1.0
                    J[r + 1, im + 1], J[r + 1, im + 3] = 1.0, -1.0
                    J[r, ipk + 2] = -st * cc[0] + ct * cc[1]
                    J[r + 1, ipk + 2] = -ct * cc[0] - st * cc[1]
                if H is not None:
                    y1, y2 = hess_y[r], hess_y[r + 1]
                    hl = y1 * (-st * A[:, 0] + ct * A[:, 1]) + y2 * (-ct * A[:, 0] - st * A[:, 1])
                    H[ipk + 2, il:il + (o1 - o0)] += hl
                    H[il:il + (o1 - o0), ipk + 2] += hl
                    H[ipk + 2, ipk + 2] += y1 * (-ct * cc[0] - st * cc[1]) + y2 * (st * cc[0] - ct * cc[1])
                r += 2
        out = [c]
        if jac:
            out.append(J)
        if H is not None:
            out.append(H)
        return out[0] if len(out) == 1 else tuple(out)

    # -- inequality constraints -----------------------------------------------
    def ineq_layout(self):
        rows = []
        for k in range(self.N + 1):
            rows += [("xbnd", k, 0), ("xbnd", k, 1)]
        for k in range(self.N):
            rows += [("ubnd", k, 0), ("ubnd", k, 1)]
        for k in range(self.N):
            rows += [("acc", k, 0), ("acc", k, 1)]
        if self.variant == 4:
            for k in range(self.N + 1):
                rows += [("Tpos", k), ("Tbnd", k)]
        if self.variant == 6:
            rows += [("termx",), ("termy",)]
        for k in range(self.N + 1):
            for i in range(self.nObs):
                rows += [("norm", k, i), ("dist", k, i)]
        for k in range(self.N + 1):
            rows += [("lam", k, j) for j in range(self.M)]
            rows += [("mu", k, j) for j in range(4 * self.nObs)]
        return rows

    def ineq_bounds(self):
        rows = self.ineq_layout()
        lb = np.full(len(rows), -INF)
        ub = np.full(len(rows), INF)
        for r, row in enumerate(rows):
            t = row[0]
            if t == "xbnd":
                lb[r], ub[r] = self.xL[row[2]], self.xU[row[2]]
            elif t == "ubnd":
                lb[r], ub[r] = self.uL[row[2]], self.uU[row[2]]
            elif t == "acc":
                lb[r], ub[r] = -ACC_MAX[row[2]], ACC_MAX[row[2]]
            elif t == "Tpos":
                lb[r] = 0.0
            elif t == "Tbnd":
                lb[r], ub[r] = T_MIN, self.Tmax
            elif t == "termx":
                lb[r] = self.term[0]
            elif t == "termy":
                lb[r], ub[r] = self.term[1], self.term[2]
            elif t == "norm":
                ub[r] = 1.0
            elif t == "dist":
                lb[r] = self.dmin
            else:
                lb[r] = 0.0
        return lb, ub

    def ineq(self, z, jac=False, hess_y=None):
        N, n = self.N, self.n
        rows = self.ineq_layout()
        d = np.zeros(len(rows))
        J = np.zeros((len(rows), n)) if jac else None
        H = np.zeros((n, n)) if hess_y is not None else None
        free = self.variant == 4
        h = self.h(z)
        T = z[self.iT()] if free else 1.0
        for r, row in enumerate(rows):
            t = row[0]
            if t == "xbnd":
                idx = self.ip(row[1]) + row[2]
                d[r] = z[idx]
                if jac:
                    J[r, idx] = 1.0
            elif t == "ubnd":
                idx = self.iu(row[1]) + row[2]
                d[r] = z[idx]
                if jac:
                    J[r, idx] = 1.0
            elif t == "acc":
                k, c_ = row[1], row[2]
                cur = self.iu(k) + c_
                prev = self.u0[c_] if k == 0 else z[self.iu(k - 1) + c_]
                q = prev - z[cur]
                d[r] = q / h
                if jac:
                    J[r, cur] = -1.0 / h
                    if k > 0:
                        J[r, self.iu(k - 1) + c_] = 1.0 / h
                    if free:
                        J[r, self.iT()] = -q / (T * h)
                if H is not None and free:
                    y = hess_y[r]
                    iT = self.iT()
                    H[cur, iT] += y / (T * h)
                    H[iT, cur] += y / (T * h)
                    if k > 0:
                        pidx = self.iu(k - 1) + c_
                        H[pidx, iT] -= y / (T * h)
                        H[iT, pidx] -= y / (T * h)
                    H[iT, iT] += y * 2 * q / (T * T * h)
            elif t in ("Tpos", "Tbnd"):
                d[r] = T
                if jac:
                    J[r, self.iT()] = 1.0
            elif t == "termx":
                d[r] = z[self.ip(N)]
                if jac:
                    J[r, self.ip(N)] = 1.0
            elif t == "termy":
                d[r] = z[self.ip(N) + 1]
                if jac:
                    J[r, self.ip(N) + 1] = 1.0
            elif t in ("norm", "dist"):
                k, i = row[1], row[2]
                lam, mu, A, b, cc = self._obst(z, k, i)
                o0, o1 = self.off_m[i], self.off_m[i + 1]
                il = self.il(k) + o0
                im = self.imu(k) + 4 * i
                mm = o1 - o0
                if t == "norm":
                    d[r] = cc[0] ** 2 + cc[1] ** 2
                    if jac:
                        J[r, il:il + mm] = 2 * (A @ cc)
                    if H is not None:
                        H[il:il + mm, il:il + mm] += hess_y[r] * 2 * (A @ A.T)
                else:
                    ipk = self.ip(k)
                    x, y_, th = z[ipk:ipk + 3]
                    ct, st = math.cos(th), math.sin(th)
                    tx, ty = x + ct * self.off, y_ + st * self.off
                    d[r] = -self.g @ mu + tx * cc[0] + ty * cc[1] - b @ lam
                    if jac:
                        J[r, im:im + 4] = -self.g
                        J[r, il:il + mm] = tx * A[:, 0] + ty * A[:, 1] - b
                        J[r, ipk] = cc[0]
                        J[r, ipk + 1] = cc[1]
                        J[r, ipk + 2] = self.off * (-st * cc[0] + ct * cc[1])
                    if H is not None:
                        y = hess_y[r]
                        H[ipk, il:il + mm] += y * A[:, 0]
                        H[il:il + mm, ipk] += y * A[:, 0]
                        H[ipk + 1, il:il + mm] += y * A[:, 1]
                        H[il:il + mm, ipk + 1] += y * A[:, 1]
                        hl = y * self.off * (-st * A[:, 0] + ct * A[:, 1])
                        H[ipk + 2, il:il + mm] += hl
                        H[il:il + mm, ipk + 2] += hl
                        H[ipk + 2, ipk + 2] += y * self.off * (-ct * cc[0] - st * cc[1])
            elif t == "lam":
                idx = self.il(row[1]) + row[2]
                d[r] = z[idx]
                if jac:
                    J[r, idx] = 1.0
            elif t == "mu":
                idx = self.imu(row[1]) + row[2]
                d[r] = z[idx]
                if jac:
                    J[r, idx] = 1.0
        out = [d]
        if jac:
            out.append(J)
        if H is not None:
            out.append(H)
        return out[0] if len(out) == 1 else tuple(out)
