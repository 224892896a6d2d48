"""ORACLE (test infrastructure): numpy blueprint of the STRUCTURED Newton-step solve that the HIP
kernel performs, checked against the dense solve of oracle/ipm_dense.py (tests/test_kkt_structured.py).

Given the condensed Hessian H (x-space: Lagrangian Hessian + delta_w + J^T E^-1 J of every condensed
elastic row), gradient b, the hard rotation rows, and the soft (elastic) initial-state and dynamics rows
with their E and ghat, the step solves

    min 1/2 dx'H dx + b'dx + sum_{soft rows} 1/2 |J dx + ghat|^2_{E^-1}    s.t.  J_rot dx = -c_rot

in two levels:
  1. per (stage k, obstacle i): LDL^T of the local block over (lambda_i, mu_i, nu_i) and a 3x3 Schur
     complement onto the pose p_k;
  2. a backward Riccati sweep over the augmented stage state xi_k = (dp_k, du_{k-1}, dT) with input du_k,
     where the elastic dynamics rows enter through P~ = (P^-1 + E)^-1 evaluated as (I + P E)^-1 P
     (no subtraction of large numbers), followed by a forward sweep.
The step is accepted only if the pivots show IPOPT's inertia for the augmented system (inertia additivity):
each local block must have exactly two negative pivots (counted, not tested by position: an indefinite
(lambda, mu) block that is positive definite on the null space of its rotation rows gives one negative primal
and one positive dual pivot), every (I + P E) pivot, input-block pivot and the time-scale pivot must be positive.
"""
import numpy as np


def ldl_nopivot(K):
    """in-place style LDL^T without pivoting; returns L (unit lower), d"""
    n = K.shape[0]
    L = np.eye(n)
    d = np.zeros(n)
    A = K.copy()
    for j in range(n):
        d[j] = A[j, j]
        L[j + 1:, j] = A[j + 1:, j] / d[j]
        A[j + 1:, j + 1:] -= np.outer(L[j + 1:, j], A[j + 1:, j])
    return L, d


def ldl_solve(L, d, B):
    Y = np.linalg.solve(L, B)
    Y = Y / d[:, None] if Y.ndim == 2 else Y / d
    return np.linalg.solve(L.T, Y)


def lu3_nopivot(Mx):
    """solve helper: returns inverse of a small matrix by LU without pivoting and its pivots"""
    n = Mx.shape[0]
    A = Mx.copy()
    inv = np.eye(n)
    piv = np.zeros(n)
    for j in range(n):
        piv[j] = A[j, j]
        for i in range(j + 1, n):
            f = A[i, j] / A[j, j]
            A[i, :] -= f * A[j, :]
            inv[i, :] -= f * inv[j, :]
    for j in range(n - 1, -1, -1):
        inv[j, :] /= A[j, j]
        for i in range(j):
            inv[i, :] -= A[i, j] * inv[j, :]
    return inv, piv


def soft_min(P, q, E):
    """V~(phat, o) = min_p' 1/2 (p'-phat)' E^-1 (p'-phat) + V(p', o); P is 6x6 over (p(3), o(3))."""
    Ppp, Ppo, Poo = P[:3, :3], P[:3, 3:], P[3:, 3:]
    Mi, piv = lu3_nopivot(np.eye(3) + Ppp * E[None, :])      # (I + Ppp E)^-1
    Pt = np.zeros((6, 6))
    Pt[:3, :3] = Mi @ Ppp
    Pt[:3, 3:] = Mi @ Ppo
    Pt[3:, :3] = Pt[:3, 3:].T
    EM = E[:, None] * Mi                                      # E M
    Pt[3:, 3:] = Poo - Ppo.T @ EM @ Ppo
    qt = np.zeros(6)
    qt[:3] = Mi @ q[:3]
    qt[3:] = q[3:] - Ppo.T @ EM @ q[:3]
    Pt[:3, :3] = 0.5 * (Pt[:3, :3] + Pt[:3, :3].T)
    return Pt, qt, Mi, piv


def structured_step(p, H, b, Jrot, crot, A, B, tcol, E_dyn, gh_dyn, E_init, gh_init, free_T, split=0, kernel_check=None):
    """
    H, b     dense condensed Hessian / gradient in x-space (blocks are read out of it)
    Jrot     dense Jacobian of the hard rotation rows (rows ordered k-major, i, 2), crot residuals
    A,B,tcol per-stage dynamics linearisation: row function J dx = dp_{k+1} - A dp_k - B du_k - tcol dT
    returns dx, dnu (rot multipliers step), dy_init(3), dy_dyn(N,3), pivots-ok flag

    split = m > 0: TWO-SIDED sweep (blueprint of the parallel Riccati): the backward recursion runs from stage N down to m
    and gives the cost-to-go V_m(xi_m); a forward recursion runs from stage 0 up to m and gives the cost-to-arrive W_m(xi_m)
    (stages 0..m-1 eliminated: per stage the 5 variables (dp_k, du_{k-1}) -- 3 at stage 0, where du_{-1} = 0 -- by LDL^T, the
    Schur complement lands on (dp_{k+1}, du_k, dT)); the two meet in one 6x6 solve for xi_m, and the two halves are
    recovered outwards independently.  Same inertia test by additivity: every eliminated block positive definite.
    """
    N, n = p.N, p.n
    nO = p.nObs
    ok = True
    Hpp = [H[p.ip(k):p.ip(k) + 3, p.ip(k):p.ip(k) + 3].copy() for k in range(N + 1)]
    bp = [b[p.ip(k):p.ip(k) + 3].copy() for k in range(N + 1)]
    iT = p.iT() if free_T else None
    # ---- level 1: local elimination ------------------------------------------------------------
    loc = {}
    r = 0
    for k in range(N + 1):
        ipk = p.ip(k)
        for i in range(nO):
            o0, o1 = p.off_m[i], p.off_m[i + 1]
            idx = list(range(p.il(k) + o0, p.il(k) + o1)) + list(range(p.imu(k) + 4 * i, p.imu(k) + 4 * i + 4))
            nw = len(idx)
            Kloc = np.zeros((nw + 2, nw + 2))
            Kloc[:nw, :nw] = H[np.ix_(idx, idx)]
            Jw = Jrot[r:r + 2][:, idx]
            Kloc[nw:, :nw] = Jw
            Kloc[:nw, nw:] = Jw.T
            G = np.zeros((nw + 2, 3))
            G[:nw] = H[np.ix_(idx, range(ipk, ipk + 3))]
            G[nw:] = Jrot[r:r + 2][:, ipk:ipk + 3]
            rloc = np.concatenate([-b[idx], -crot[r:r + 2]])
            L, d = ldl_nopivot(Kloc)
            if np.any(d == 0) or np.sum(d < 0) != 2:        # inertia by count (Sylvester), not by position
                ok = False
            Y = ldl_solve(L, d, np.column_stack([G, rloc]))
            Hpp[k] -= G.T @ Y[:, :3]
            bp[k] += G.T @ Y[:, 3]
            loc[(k, i)] = (idx, Y, r)
            r += 2
    # ---- level 2: Riccati over xi = (dp, du_prev, dT) -----------------------------------------------
    def blk(i0, n0, i1, n1):
        return H[i0:i0 + n0, i1:i1 + n1]

    P = np.zeros((6, 6))
    q = np.zeros(6)
    P[:3, :3] = Hpp[N]
    q[:3] = bp[N]
    if not free_T:
        pass
    gains = [None] * N
    m_split = int(split)
    assert 0 <= m_split < N or N == 0

    def stage_blocks(k):
        ipk, iuk = p.ip(k), p.iu(k)
        F = np.zeros((6, 6))
        F[:3, :3] = A[k]
        if free_T:
            F[:3, 5] = tcol[k]
        F[5, 5] = 1.0
        G = np.zeros((6, 2))
        G[:3] = B[k]
        G[3:5] = np.eye(2)
        f = np.concatenate([-gh_dyn[k], np.zeros(3)])
        Lxx = np.zeros((6, 6))
        Lxx[:3, :3] = Hpp[k]
        Lxu = np.zeros((6, 2))
        Lxu[:3] = blk(ipk, 3, iuk, 2)
        lx = np.zeros(6)
        lx[:3] = bp[k]
        if k >= 1:
            Lxu[3:5] = blk(p.iu(k - 1), 2, iuk, 2)
        if free_T:
            Lxx[:3, 5] = H[ipk:ipk + 3, iT]
            Lxx[5, :3] = H[ipk:ipk + 3, iT]
            Lxu[5] = H[iT, iuk:iuk + 2]
            if k == 0:
                Lxx[5, 5] = H[iT, iT]
                lx[5] = b[iT]
        elif k == 0:
            Lxx[5, 5] = 1.0
        Luu = blk(iuk, 2, iuk, 2)
        lu = b[iuk:iuk + 2]
        return F, G, f, Lxx, Lxu, Luu, lx, lu

    for k in range(N - 1, m_split - 1, -1):
        Pt, qt, Mi, piv = soft_min(P, q, E_dyn[k])
        if np.any(piv <= 0):
            ok = False
        F, G, f, Lxx, Lxu, Luu, lx, lu = stage_blocks(k)
        Pf = Pt @ f + qt
        Mxx = Lxx + F.T @ Pt @ F
        Mxu = Lxu + F.T @ Pt @ G
        Muu = Luu + G.T @ Pt @ G
        mx = lx + F.T @ Pf
        mu_ = lu + G.T @ Pf
        Lc, dc = ldl_nopivot(Muu)
        if np.any(dc <= 0):
            ok = False
        Kg = -ldl_solve(Lc, dc, Mxu.T)          # 2x6
        kap = -ldl_solve(Lc, dc, mu_)
        Pn = Mxx + Mxu @ Kg
        qn = mx + Mxu @ kap
        gains[k] = (Kg, kap, Mi, P.copy(), q.copy(), F, G, f)
        P, q = 0.5 * (Pn + Pn.T), qn
    dx = np.zeros(n)
    dy_dyn = np.zeros((N, 3))
    if m_split == 0:
        # stage 0: u_prev step is zero (u_{-1} = u0 is data); soft initial condition; then dT
        Pt, qt, Mi0, piv = soft_min(P, q, E_init)
        if np.any(piv <= 0):
            ok = False
        phat0 = -gh_init
        dT = 0.0
        if free_T:
            if Pt[5, 5] <= 0:
                ok = False
            dT = -(qt[5] + Pt[5, :3] @ phat0) / Pt[5, 5]
        o = np.array([0.0, 0.0, dT])
        dp = Mi0.T @ (phat0 - E_init * (P[:3, 3:] @ o) - E_init * q[:3])
        dy_init = -(P[:3, :3] @ dp + P[:3, 3:] @ o + q[:3])
        uprev = np.zeros(2)
    else:
        # ---- forward half: cost-to-arrive W_k(xi) = 1/2 xi' Pi xi + pi' xi over xi = (dp_k, du_{k-1}, dT) -------------
        # s = (dp(0:3), du_prev(3:5), dT(5), du(6:8), dp'(8:11)); eliminated: (dp, du_prev) -- at stage 0 only dp
        Pi = np.zeros((6, 6))
        pi = np.zeros(6)
        D0 = 1.0 / E_init
        Pi[:3, :3] = np.diag(D0)
        pi[:3] = D0 * gh_init                   # 1/2 (dp + gh)' D0 (dp + gh)
        fw = [None] * m_split
        for k in range(m_split):
            F, G, f, Lxx, Lxu, Luu, lx, lu = stage_blocks(k)
            Q = np.zeros((11, 11))
            c = np.zeros(11)
            Q[:6, :6] = Pi + Lxx
            Q[:6, 6:8] = Lxu
            Q[6:8, :6] = Lxu.T
            Q[6:8, 6:8] = Luu
            c[:6] = pi + lx
            c[6:8] = lu
            Rr = np.zeros((3, 11))                # residual p' - A p - tcol T - B u + gh
            Rr[:, 0:3] = -F[:3, :3]
            Rr[:, 5] = -F[:3, 5]
            Rr[:, 6:8] = -G[:3]
            Rr[:, 8:11] = np.eye(3)
            Dk = 1.0 / E_dyn[k]
            Q += Rr.T @ (Dk[:, None] * Rr)
            c += Rr.T @ (Dk * gh_dyn[k])
            v = [0, 1, 2] if k == 0 else [0, 1, 2, 3, 4]
            w = [8, 9, 10, 6, 7, 5]              # (dp', du' = du, dT): the next stage's xi
            Lv, dv = ldl_nopivot(Q[np.ix_(v, v)])
            if np.any(dv <= 0):
                ok = False
            Qvw = Q[np.ix_(v, w)]
            Z = ldl_solve(Lv, dv, np.column_stack([Qvw, c[v]]))
            fw[k] = (v, Z)                        # v* = -(Z[:, :6] w + Z[:, 6])
            if kernel_check is not None:
                kernel_check(k, Pi, pi, F, G, Lxx, Lxu, Luu, lx, lu, Dk, gh_dyn[k], Z,
                             Q[np.ix_(w, w)] - Qvw.T @ Z[:, :6], c[w] - Qvw.T @ Z[:, 6])
            Pi = Q[np.ix_(w, w)] - Qvw.T @ Z[:, :6]
            Pi = 0.5 * (Pi + Pi.T)
            pi = c[w] - Qvw.T @ Z[:, 6]
        # ---- the two halves meet at stage m
        Lm, dm = ldl_nopivot(Pi + P)
        if np.any(dm <= 0):
            ok = False
        xi_m = -ldl_solve(Lm, dm, pi + q)
        if not free_T:
            pass                                  # Lxx[5,5] = 1 at stage 0 and no coupling: dT comes out as 0
        dT = xi_m[5]
        if free_T:
            dx[iT] = dT
        # forward half recovered backwards; costates of the elastic rows from the cost-to-arrive
        xi_next = xi_m.copy()
        Pis, pis = Pi, pi
        # (the cost-to-arrive of every stage is needed for the costates: recompute them on the way -- the blueprint keeps it simple)
        Pi_list = []
        Pi2 = np.zeros((6, 6)); pi2 = np.zeros(6)
        Pi2[:3, :3] = np.diag(D0); pi2[:3] = D0 * gh_init
        for k in range(m_split):
            F, G, f, Lxx, Lxu, Luu, lx, lu = stage_blocks(k)
            Q = np.zeros((11, 11)); c = np.zeros(11)
            Q[:6, :6] = Pi2 + Lxx; Q[:6, 6:8] = Lxu; Q[6:8, :6] = Lxu.T; Q[6:8, 6:8] = Luu
            c[:6] = pi2 + lx; c[6:8] = lu
            Rr = np.zeros((3, 11)); Rr[:, 0:3] = -F[:3, :3]; Rr[:, 5] = -F[:3, 5]; Rr[:, 6:8] = -G[:3]; Rr[:, 8:11] = np.eye(3)
            Dk = 1.0 / E_dyn[k]
            Q += Rr.T @ (Dk[:, None] * Rr); c += Rr.T @ (Dk * gh_dyn[k])
            v, Z = fw[k]
            w = [8, 9, 10, 6, 7, 5]
            Qvw = Q[np.ix_(v, w)]
            Pi2 = Q[np.ix_(w, w)] - Qvw.T @ Z[:, :6]; Pi2 = 0.5 * (Pi2 + Pi2.T)
            pi2 = c[w] - Qvw.T @ Z[:, 6]
            Pi_list.append((Pi2.copy(), pi2.copy()))
        for k in range(m_split - 1, -1, -1):
            v, Z = fw[k]
            vs = -(Z[:, :6] @ xi_next + Z[:, 6])
            Pk1, pk1 = Pi_list[k]
            dy_dyn[k] = (Pk1 @ xi_next + pk1)[:3]
            dx[p.ip(k + 1):p.ip(k + 1) + 3] = xi_next[:3]
            dx[p.iu(k):p.iu(k) + 2] = xi_next[3:5]
            xi_k = np.zeros(6)
            xi_k[:len(vs)] = vs
            xi_k[5] = dT
            xi_next = xi_k
        dx[p.ip(0):p.ip(0) + 3] = xi_next[:3]
        dy_init = D0 * (xi_next[:3] + gh_init)
        dp, uprev = xi_m[:3].copy(), xi_m[3:5].copy()
    if m_split == 0:
        dx[p.ip(0):p.ip(0) + 3] = dp
        if free_T:
            dx[iT] = dT
    for k in range(m_split, N):
        Kg, kap, Mi, Pn1, qn1, F, G, f = gains[k]
        xi = np.concatenate([dp, uprev, [dT]])
        u = Kg @ xi + kap
        ph = F @ xi + G @ u + f          # (phat, du_k, dT)
        o = ph[3:]
        dpn = Mi.T @ (ph[:3] - E_dyn[k] * (Pn1[:3, 3:] @ o) - E_dyn[k] * qn1[:3])
        dy_dyn[k] = -(Pn1[:3, :3] @ dpn + Pn1[:3, 3:] @ o + qn1[:3])
        dx[p.iu(k):p.iu(k) + 2] = u
        dx[p.ip(k + 1):p.ip(k + 1) + 3] = dpn
        dp, uprev = dpn, u
    # local recovery
    dnu = np.zeros(crot.size)
    for (k, i), (idx, Y, r) in loc.items():
        sol = Y[:, 3] - Y[:, :3] @ dx[p.ip(k):p.ip(k) + 3]
        dx[idx] = sol[:len(idx)]
        dnu[r:r + 2] = sol[len(idx):]
    return dx, dnu, dy_init, dy_dyn, ok
