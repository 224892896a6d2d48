"""ORACLE (test infrastructure, never on the product path): the OBCA NLP of oracle/obca_nlp.py handed to
CasADi/IPOPT -- the solver the reference itself uses (/root/reference/src/obca.py:1044-1056 obca_mpc4 with IPOPT's
defaults, :1538-1550 / :1745-1746 obca_mpc6/8 with max_iter 1000, acceptable_tol 1e-8, acceptable_obj_change_tol 1e-6).

casadi is NOT installed in the build container or on the GPU box (no wheel, no network: SURVEY.md section 8c), so this
module cannot run there: `available()` is False, tests/test_ipopt_optional.py skips itself and bench.py reports
"ipopt": "unavailable".  It exists so that wherever `import casadi` does succeed, the product's answers can be put next
to IPOPT's on identical inputs.  It is written from the NLP statement in SURVEY.md Appendix A / oracle/obca_nlp.py (which
tests/test_oracle_nlp.py pins to the reference's model code), not from the reference's source text.

What IPOPT is given is what the reference gives it: decision variables x (3,N+1), u (2,N), lambda with rows(AObs) rows per
stage (only M of them enter a constraint besides >= 0, Appendix A.1), mu (4 nObs, N+1) and N+1 copies of Topt tied by
equalities; every bound is a general constraint (Opti has no simple bounds); start point zero with Topt = 1.
"""
import numpy as np


def available():
    try:
        import casadi  # noqa: F401
        return True
    except Exception:
        return False


def solve(p, dummy_lambda_rows=None, print_level=0):
    """p: oracle.obca_nlp.Problem.  Returns dict(x [3,N+1], u [2,N], Ts_opt, feas, iters) like the reference's
    4-tuple (plus IPOPT's iteration count)."""
    import casadi as ca
    N, M, nO = p.N, p.M, p.nObs
    free = p.variant == 4
    opti = ca.Opti()
    x = opti.variable(3, N + 1)
    u = opti.variable(2, N)
    n_dummy = (N * M if dummy_lambda_rows is None else int(dummy_lambda_rows))      # rows(AObs) = (N+1) M in the reference
    lam = opti.variable(M + n_dummy, N + 1)
    mu = opti.variable(4 * nO, N + 1)
    if free:
        T = opti.variable(N + 1)
        opti.set_initial(T, 1)
        step = lambda k: T[k] * p.Ts
    else:
        step = lambda k: p.Ts

    def lam_rows(k):
        """first row of stage k's effective multipliers: 0 for obca_mpc4 (row counter reset per stage), k*M for
        obca_mpc6/8 (row counter runs across stages) -- SURVEY Appendix A.3-q5"""
        return 0 if free else k * M
    assert free or n_dummy == N * M, "obca_mpc6/8 index lambda rows k*M..(k+1)*M-1 of column k"

    # ---- objective (Appendix A.1 / A.2) -----------------------------------------------------------------------
    f = 0
    for t in range(N):
        e = x[:, t] - p.xref[:, t]
        f += ca.mtimes([e.T, ca.DM(p.Q), e]) + ca.mtimes([u[:, t].T, ca.DM(p.R1), u[:, t]])
        if t < N - 1:
            q = (u[:, t + 1] - u[:, t]) / step(t)
            f += ca.mtimes([q.T, ca.DM(p.R2), q])
    e = x[:, N] - p.xref[:, N]
    f += ca.mtimes([e.T, ca.DM(p.P), e])
    if free:
        for k in range(N + 1):
            f += 10 * T[k] + T[k] ** 2
    opti.minimize(f)

    # ---- constraints, in the order the reference issues them ------------------------------------------------------
    for k in range(N):
        opti.subject_to(x[0, k + 1] == x[0, k] + step(k) * (u[0, k] * ca.cos(x[2, k])))
        opti.subject_to(x[1, k + 1] == x[1, k] + step(k) * (u[0, k] * ca.sin(x[2, k])))
        opti.subject_to(x[2, k + 1] == x[2, k] + step(k) * u[1, k])
        if free:
            opti.subject_to(T[k] == T[k + 1])
    for i in range(2):
        opti.subject_to(opti.bounded(p.xL[i], x[i, :], p.xU[i]))
    for i in range(2):
        opti.subject_to(opti.bounded(p.uL[i], u[i, :], p.uU[i]))
    from .obca_nlp import ACC_MAX, T_MIN
    for k in range(N):
        for c in range(2):
            prev = p.u0[c] if k == 0 else u[c, k - 1]
            opti.subject_to(opti.bounded(-ACC_MAX[c], (prev - u[c, k]) / step(k), ACC_MAX[c]))
    opti.subject_to(x[:, 0] == p.x0)
    if free:
        opti.subject_to(x[:, N] == p.xref[:, N])
    if p.variant == 6:
        opti.subject_to(x[0, N] >= p.term[0])
        opti.subject_to(opti.bounded(p.term[1], x[1, N], p.term[2]))
    for k in range(N + 1):
        opti.subject_to(lam[:, k] >= 0)
        opti.subject_to(mu[:, k] >= 0)
        if free:
            opti.subject_to(T[k] > 0)
            opti.subject_to(opti.bounded(T_MIN, T[k], p.Tmax))
    for k in range(N + 1):
        r0 = lam_rows(k)
        ct, st = ca.cos(x[2, k]), ca.sin(x[2, k])
        for i in range(nO):
            o0, o1 = int(p.off_m[i]), int(p.off_m[i + 1])
            A = ca.DM(p.A[k, o0:o1])
            b = ca.DM(p.b[k, o0:o1])
            li = lam[r0 + o0:r0 + o1, k]
            mi = mu[4 * i:4 * i + 4, k]
            c = ca.mtimes(A.T, li)
            opti.subject_to(c[0] ** 2 + c[1] ** 2 <= 1)
            opti.subject_to(mi[0] - mi[2] + ct * c[0] + st * c[1] == 0)
            opti.subject_to(mi[1] - mi[3] - st * c[0] + ct * c[1] == 0)
            dist = -ca.dot(ca.DM(p.g), mi) + (x[0, k] + ct * p.off) * c[0] + (x[1, k] + st * p.off) * c[1] - ca.dot(b, li)
            opti.subject_to(dist >= p.dmin)

    opts = {"ipopt.print_level": int(print_level), "print_time": 0}
    if not free:                                   # obca.py:1538-1539
        opts.update({"ipopt.max_iter": 1000, "ipopt.acceptable_tol": 1e-8, "ipopt.acceptable_obj_change_tol": 1e-6})
    opti.solver("ipopt", opts)
    out = {}
    try:
        sol = opti.solve()
        val, out["feas"] = sol.value, True
        out["iters"] = int(sol.stats().get("iter_count", -1))
    except Exception:                              # the reference's bare except: last iterate, feas = False
        val, out["feas"] = opti.debug.value, False
        out["iters"] = int(opti.debug.stats().get("iter_count", -1)) if hasattr(opti.debug, "stats") else -1
    out["x"] = np.asarray(val(x)).reshape(3, N + 1)
    out["u"] = np.asarray(val(u)).reshape(2, N)
    out["Ts_opt"] = float(val(T[0] * p.Ts)) if free else p.Ts
    return out
