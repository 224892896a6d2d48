"""ORACLE (test infrastructure, never on the product path): dense numpy specification
of the elastic primal-dual interior-point method used for the OBCA NLP.

The reference hands the NLP to IPOPT through CasADi's Opti
(/root/reference/src/obca.py:1044-1056, 1538-1550, 1745-1746).  IPOPT itself is
not available in the build container, and the reference's
cold start (all variables 0, Topt 1; src/obca.py:856) is a point where the
terminal-constraint Jacobian is rank deficient (v = 0), from which a plain
Newton/filter IPM without IPOPT's feasibility-restoration phase stalls.  This
file therefore restates IPOPT's published algorithm (Waechter & Biegler, Math.
Prog. 106, 2006; option defaults of IPOPT 3.12-3.14) applied to the *elastic*
form of the NLP -- the same l1 relaxation IPOPT's own restoration phase solves,
kept on throughout with the true objective:

    min f(x) + rho * sum_r (p_r + n_r)
    s.t. c_h(x) = 0                          hard rows: the rotation equalities G'mu + R'A'lambda = 0 only
         g_r(x) - s_r - p_r + n_r = 0        elastic rows: every inequality and every other equality
         L_r <= s_r <= U_r,  p_r, n_r >= 0   (initial state, dynamics, terminal equalities: s_r == 0)

Opti poses every inequality -- simple bounds included -- as a general
constraint, so all of them are rows here; decision variables are free.  With
rho above the multiplier norm the elastic problem has the NLP's KKT points with
p = n = 0 (exact penalty); a converged point with p + n > 0 is reported
infeasible, which is what the reference's ``feas=False`` means.

Kept from IPOPT: monotone barrier update (mu_init 0.1, kappa_mu 0.2, theta_mu
1.5, kappa_eps 10), fraction-to-boundary tau = max(0.99, 1-mu), slack/bound push
1e-2, filter line search with second-order correction, inertia-correcting
delta_w ladder, dual reset (kappa_sigma 1e10), scaled optimality error (s_max
100), gradient-based objective scaling (max gradient 100), tol 1e-8,
acceptable-point logic with the reference's options for mpc6/mpc8.

PARITY at the solver boundary: IPOPT cannot run here; the one IPOPT output the
reference repository holds -- its GIF of the demo9 closed loop, 83 chained solves
whose Ts_opt the frame titles carry (tests/golden/reference_gif_demo9.json) --
pins the C twin of this file on 69 consecutive steps with the default start ladder
(x0 -> window -> zeros; 42 with the reference's literal zero start first:
tests/test_reference_gif.py; at step 70 the GIF's own answer is the one that is not the
best optimum).  The second output the reference holds -- the picture of its open-loop
plan of demo9 at N = 50 -- pins the structured core and the kernels, which follow this
file (tests/test_reference_openloop.py; this dense version is too slow at that size).
Beyond those two parity is UNPINNED.  What else pins this file: (1) the NLP functions are pinned to the
reference's model code (tests/test_oracle_nlp.py); (2) KKT certificates on the
ORIGINAL NLP and the independent known answers of SURVEY.md Appendix C
(tests/test_oracle_ipm.py).

This dense version is the readable specification; oracle/obca_oracle.c is the
same algorithm in C (the CPU baseline); the HIP kernel is checked against both.
"""
import math

import numpy as np
import scipy.linalg

STATUS_OK = 0
STATUS_ACCEPTABLE = 1
STATUS_INFEASIBLE = 2          # converged, but elastic variables did not vanish
STATUS_MAXITER = -1
STATUS_LINESEARCH = -2
STATUS_NUMERIC = -3
STATUS_BAD_BOUNDS = -4

DEFAULTS = dict(
    tol=1e-8, max_iter=3000, acceptable_tol=1e-6, acceptable_iter=15,
    acceptable_constr_viol_tol=1e-2, acceptable_dual_inf_tol=1e10, acceptable_compl_inf_tol=1e-2,
    acceptable_obj_change_tol=1e20,
    dual_inf_tol=1.0, constr_viol_tol=1e-4, compl_inf_tol=1e-4,
    mu_init=0.1, kappa_mu=0.2, theta_mu=1.5, kappa_eps=10.0, tau_min=0.99,
    bound_push=1e-2, bound_frac=1e-2, kappa_d=1e-5, kappa_sigma=1e10, s_max=100.0,
    gamma_theta=1e-5, gamma_phi=1e-8, delta=1.0, s_theta=1.1, s_phi=2.3, eta_phi=1e-8, gamma_alpha=0.05,
    kappa_soc=0.99, max_soc=4, theta_max_fact=1e4, theta_min_fact=1e-4,
    delta_w_min=1e-20, delta_w_0=1e-4, delta_w_max=1e40, kappa_w_plus=8.0,
    kappa_w_plus_bar=100.0, kappa_w_minus=1.0 / 3.0,
    nlp_scaling_max_gradient=100.0,
    rho=1e4, feas_tol=1e-6,
)


def options_for(variant):
    o = dict(DEFAULTS)
    if variant in (6, 8):   # src/obca.py:1538-1539, 1734-1735
        o.update(max_iter=1000, acceptable_tol=1e-8, acceptable_obj_change_tol=1e-6)
    return o


HARD_KINDS = ('rot',)


class Result:
    pass


def _ldl_inertia(K):
    """(#positive, #negative) eigenvalues from a Bunch-Kaufman LDL^T."""
    _, D, _ = scipy.linalg.ldl(K, lower=True)
    n = D.shape[0]
    pos = neg = 0
    i = 0
    while i < n:
        if i + 1 < n and D[i + 1, i] != 0.0:
            a, b, c = D[i, i], D[i + 1, i], D[i + 1, i + 1]
            det = a * c - b * b
            tr = a + c
            if det < 0:
                pos += 1
                neg += 1
            elif tr > 0:
                pos += 2
            else:
                neg += 2
            i += 2
        else:
            if D[i, i] > 0:
                pos += 1
            elif D[i, i] < 0:
                neg += 1
            i += 1
    return pos, neg


def split_rows(p):
    """hard equality rows (rot) and elastic equality rows (init, dyn, terminal); the inequalities are all elastic"""
    lay = p.eq_layout()
    kinds = HARD_KINDS
    hard = np.array([i for i, r in enumerate(lay) if r[0] in kinds], dtype=int)
    term = np.array([i for i, r in enumerate(lay) if r[0] not in kinds], dtype=int)
    return hard, term


RHO_ESCALATION = (100.0, 1000.0)   # csrc/obca_device.h: OBCA_RHO_ESCALATION -- the penalties tried after the base one, one after the other
RESTART_MU = 1.0               # barrier parameter the window start begins with (csrc/obca_device.h: OBCA_RESTART_MU)
WINDOW_SPEED_FRAC = 0.9
# the three starts of the ladder and the three orders (include/obca_mpc.h: start_order; csrc/obca_device.h: OBCA_START_KIND)
KIND_X0, KIND_WINDOW, KIND_ZEROS = 0, 1, 2
START_ORDERS = {1: (KIND_WINDOW, KIND_X0, KIND_ZEROS), 2: (KIND_ZEROS, KIND_WINDOW, KIND_X0), 3: (KIND_X0, KIND_WINDOW, KIND_ZEROS)}
ORDER_NAMES = {"default": 0, "x0": 3, "window": 1, "zeros": 2}
# the dodge rung (csrc/obca_device.h: OBCA_KIND_DODGE_R / _L, OBCA_DODGE_OFFSET, OBCA_DODGE_RAMP)
KIND_DODGE_R, KIND_DODGE_L = 3, 4
DODGE_OFFSET = 3.0
DODGE_RAMP = 3
DODGE_MIN_SPARE = 0.1          # obca_mpc6: metres of reach beyond the terminal set below which the rung is not tried (OBCA_DODGE_MIN_SPARE)
# second level of the rung (round 6; obca_mpc8 only -- the variant with no fallback behind it: where it fails the reference's closed loop
# stops, src/closed_loop.py:401-413): the same two starts with IPOPT's own initial barrier parameter (mu_init 0.1) instead of
# RESTART_MU.  The dodge starts are nearly feasible, with active rows; a barrier parameter of 1 pushes the first iterates far enough
# from them to fall back into the basin of the infeasible stationary point.  Seen on C5 world 667 (step 20), then on the only two solver
# failures of 8192 worlds no rule was looked at on (7225, 11188: profiles/r06_c5_heldout_failures.json): all three end feasible.
KIND_DODGE_R2, KIND_DODGE_L2 = 5, 6
DODGE_LEVEL2_MU = 0.1


def retry_iter(N):
    """iteration limit of the passes of every start after the first (csrc/obca_device.h: OBCA_RETRY_ITER: starts that succeed
    there take 16-117 iterations at N <= 20)"""
    return 300 + 10 * N


def patience(N):
    """iteration limit of the first start's passes while further starts remain (csrc/obca_device.h: OBCA_PATIENCE)"""
    return 500 + 10 * N


def x0_start(p):
    """The default first start: the reference's cold start (all variables 0, Topt = 1; src/obca.py:856) with every pose at
    x0 -- the iterate IPOPT's first full Newton step reaches from the all-zero start (linearised at v = 0 the dynamics read
    x_{k+1} = x_k, the initial condition x_0 = x0).  Uses nothing but x0."""
    z = p.start_point()
    for k in range(p.N + 1):
        z[p.ip(k):p.ip(k) + 3] = p.x0
    return z


def window_start(p):
    """The window start of the ladder: the poses of the reference window the call was given (first pose = x0), the
    inputs that drive from pose to pose (finite differences, clipped to the input box), for the free-time problem the
    time scale at which the window is driven at 0.9 of the speed bound (at least 1, at most max_Topt); lambda = mu = 0."""
    z = np.zeros(p.n)
    pts = p.xref.copy()
    pts[:, 0] = p.x0
    T = 1.0
    if p.variant == 4:
        length = 0.0
        for k in range(p.N):
            length += math.sqrt((pts[0, k + 1] - pts[0, k]) ** 2 + (pts[1, k + 1] - pts[1, k]) ** 2)
        T = min(max(1.0, length / (p.N * WINDOW_SPEED_FRAC * p.uU[0] * p.Ts)), max(1.0, p.Tmax))
        z[p.iT()] = T
    h = p.Ts * T
    for k in range(p.N + 1):
        z[p.ip(k):p.ip(k) + 3] = pts[:, k]
        if k < p.N:
            d = pts[:, k + 1] - pts[:, k]
            z[p.iu(k)] = min(max(math.sqrt(d[0] * d[0] + d[1] * d[1]) / h, p.uL[0]), p.uU[0])
            z[p.iu(k) + 1] = min(max(d[2] / h, p.uL[1]), p.uU[1])
    return z


def dodge_start(p, side):
    """The dodge starts of the ladder's last rung (fixed-time problems on which every start of the order ended without a
    feasible point).  Where the reference window runs head-on into an obstacle the l1 penalty problem has a stationary point
    that is symmetric about the window -- the plan brakes in front of the obstacle, the terminal set and the distance rows share
    the violation -- and x0, window and zeros all end there although a plan around the obstacle exists (the reference's own
    demo11 run, steps 21-25: IPOPT drives around, tests/test_reference_demo11.py).  side = -1 / +1: the window moved to the
    right / left of the direction of travel by DODGE_OFFSET metres, ramped in over the first DODGE_RAMP stages; headings along
    the moved poses (continued from x0's heading without a jump by 2 pi), inputs by differences clipped to their box, the
    free-time scale as window_start's; lambda, mu of every (stage, obstacle) pair: the half-space row with the largest gap to
    the car at that pose, lambda = 1 / ||A_j|| on it, mu from the rotation equalities G'mu = -R A'lambda."""
    N = p.N
    z = np.zeros(p.n)
    base = p.xref.copy()
    base[:, 0] = p.x0
    pts = base.copy()
    for k in range(1, N + 1):
        kb = min(k + 1, N)
        ax, ay = base[0, kb] - base[0, k - 1], base[1, kb] - base[1, k - 1]
        ln = math.sqrt(ax * ax + ay * ay)
        th = p.xref[2, k]
        nx, ny = (-ay / ln, ax / ln) if ln > 1e-9 else (-math.sin(th), math.cos(th))
        w = side * DODGE_OFFSET * (k / DODGE_RAMP if k < DODGE_RAMP else 1.0)
        pts[0, k] = base[0, k] + w * nx
        pts[1, k] = base[1, k] + w * ny
    pts[2, 0] = p.x0[2]
    for k in range(1, N + 1):
        prev = pts[2, k - 1]
        th = prev
        if k < N:
            ddx, ddy = pts[0, k + 1] - pts[0, k], pts[1, k + 1] - pts[1, k]
            if ddx * ddx + ddy * ddy > 1e-18:
                d = math.atan2(ddy, ddx) - prev
                d -= 2.0 * math.pi * math.floor(d / (2.0 * math.pi) + 0.5)
                th = prev + d
        pts[2, k] = th
    T = 1.0
    if p.variant == 4:
        length = 0.0
        for k in range(N):
            length += math.sqrt((pts[0, k + 1] - pts[0, k]) ** 2 + (pts[1, k + 1] - pts[1, k]) ** 2)
        T = min(max(1.0, length / (N * WINDOW_SPEED_FRAC * p.uU[0] * p.Ts)), max(1.0, p.Tmax))
        z[p.iT()] = T
    h = p.Ts * T
    for k in range(N + 1):
        z[p.ip(k):p.ip(k) + 3] = pts[:, k]
        if k < N:
            d = pts[:, k + 1] - pts[:, k]
            z[p.iu(k)] = min(max(math.sqrt(d[0] * d[0] + d[1] * d[1]) / h, p.uL[0]), p.uU[0])
            z[p.iu(k) + 1] = min(max(d[2] / h, p.uL[1]), p.uU[1])
    for k in range(N + 1):
        th = pts[2, k]
        ct, st = math.cos(th), math.sin(th)
        tx, ty = pts[0, k] + ct * p.off, pts[1, k] + st * p.off
        ks = 0 if p.variant == 4 else k                      # obca_mpc4 reads the rows of step 0 only
        for i in range(p.nObs):
            o0, o1 = p.off_m[i], p.off_m[i + 1]
            best = None
            for j in range(o0, o1):
                a0, a1 = p.A[ks, j]
                nrm = math.sqrt(a0 * a0 + a1 * a1)
                if not nrm > 0.0:
                    continue
                v0, v1 = a0 / nrm, a1 / nrm
                r0, r1 = ct * v0 + st * v1, -st * v0 + ct * v1
                mu = (max(-r0, 0.0), max(-r1, 0.0), max(r0, 0.0), max(r1, 0.0))
                gap = -(p.g[0] * mu[0] + p.g[1] * mu[1] + p.g[2] * mu[2] + p.g[3] * mu[3]) + (a0 * tx + a1 * ty - p.b[ks, j]) / nrm
                if best is None or gap > best[0]:
                    best = (gap, j, 1.0 / nrm, mu)
            if best is not None:
                z[p.il(k) + best[1]] = best[2]
                z[p.imu(k) + 4 * i:p.imu(k) + 4 * i + 4] = best[3]
    return z


def terminal_set_shortfall(p, feas_tol=None):
    """Closed-form screen of obca_mpc6 (csrc/obca_device.h: obca_terminal_shortfall): by how much NO trajectory the rows allow can
    reach the terminal set's x_N >= term[0] (src/obca.py:1465).  x_0 = x0 and the heading of the first step is x0's; the speeds are
    bounded by the input box and, from u0, by the acceleration rows (src/obca.py:928-939):
        vhi_k = min(uU, vhi_{k-1} + a Ts),  vlo_k = max(uL, vlo_{k-1} - a Ts),  vhi_{-1} = vlo_{-1} = u0[0],  a = 0.6
        x_N - x_0 <= Ts max(vhi_0 cos th0, vlo_0 cos th0) + Ts sum_{k >= 1} max(|vhi_k|, |vlo_k|),   x_N <= xU
    Returns term[0] - (largest reachable x_N) - margin, where the margin is everything elastic variables of size feas_tol on the
    rows involved can add (initial state, N dynamics rows, the terminal row, N input-box rows, the acceleration rows, whose
    slack accumulates): feas_tol (N + 2 + N Ts + Ts^2 N (N + 1) / 2), doubled.  Positive: the solve would end 'infeasible'
    whatever the obstacles do -- it is not run (status 2, zero iterations, the x0 start as the iterate).  The closed loop meets
    this constantly: its terminal set is x0 + 5 (src/closed_loop.py:371) and N_fix Ts_opt uU = 5.000 m exactly while the car
    follows the 1 m lattice path at full speed, so after the first dodge (heading != 0, or a braked step) obca_mpc6 cannot
    succeed and obca_mpc8 answers (src/closed_loop.py:393-398)."""
    if p.variant != 6:
        return -math.inf
    tol = DEFAULTS["feas_tol"] if feas_tol is None else feas_tol
    h, a = p.Ts, 0.6                                   # oracle/obca_nlp.py: ACC_MAX[0]
    vhi = vlo = p.u0[0]
    reach = 0.0
    c0 = math.cos(p.x0[2])
    for k in range(p.N):
        vhi = min(p.uU[0], vhi + a * h)
        vlo = max(p.uL[0], vlo - a * h)
        reach += h * (max(vhi * c0, vlo * c0) if k == 0 else max(abs(vhi), abs(vlo)))
    xN = min(p.x0[0] + reach, p.xU[0])
    margin = 2.0 * tol * (p.N + 2 + p.N * h + h * h * p.N * (p.N + 1) / 2.0)
    return p.term[0] - xN - margin


def _screened(p, shortfall):
    """what a screened-out obca_mpc6 returns: status 'infeasible', no iteration, the x0 start as the iterate"""
    r = Result()
    r.x = x0_start(p)
    r.status, r.iters, r.nfact, r.feas = STATUS_INFEASIBLE, 0, 0, False
    r.f, r.elastic, r.E0, r.mu = 0.0, float(shortfall), 0.0, 0.0
    r.xopt, r.uopt = p.unpack_xu(r.x)
    r.Ts_opt = p.Ts
    r.starts_used, r.restarted, r.screened = 0, False, True
    return r


def solve(p, opts=None, trace=None):
    """The elastic IPM run through the START LADDER (same rule in oracle/obca_oracle.c, csrc/obca_lpi_core.h:run_instance and
    the wave kernels; include/obca_mpc.h: start_order, single_start, patience, retry_iter):

    * the starts of the order (default: reference window -> x0 -> zeros) are tried one after the other until one ends at a
      feasible point.  IPOPT answers a solve that ends at an infeasible stationary point of its merit function, in a
      line-search failure or at the iteration limit with its feasibility-restoration phase; measured on such instances the
      restoration problem min ||c||_1 + zeta/2 ||D(x - x_R)||^2 started AT the stationary point x_R does not move (the l1
      merit is at a local minimum there: a plan that dives under / through a moving box), whereas the same method from
      another start the call's own inputs describe converges on 97 % of them.  A genuinely infeasible problem stays
      infeasible.
    * penalty escalation (free-time problem only): the l1 penalty is exact only while rho exceeds the multipliers; if
      obca_mpc4 converges with elastic variables left (what "infeasible" looks like, but also what a too small rho looks
      like -- seen on the open-loop problem of demo1, N = 10) the SAME start is repeated with rho * 100 and, if elastic
      variables still remain, with rho * 1000 (the reference's open-loop plan of demo9 at N = 50 needs 1e7: tests/golden/
      reference_openloop_demo9.json); the next start begins at the base penalty again (measured on the reference's GIF run: with the raised penalty kept, the window and the
      x0 start fail on a problem both solve at the base penalty).

    opts: ``start_order`` 0 / 1 / 2 / 3 or "default" / "window" / "zeros" / "x0" (include/obca_mpc.h: OBCA_START_*); ``single_start``; ``patience``; ``retry_iter``;
    ``no_escalation``.  With ``single_start`` a pass runs to ``max_iter``; otherwise the first start's passes stop after
    ``patience`` iterations, the later starts' after ``retry_iter``."""
    opts = dict(opts or {})
    short = terminal_set_shortfall(p, opts.get("feas_tol"))          # -inf unless obca_mpc6
    if short > 0.0 and opts.get("terminal_screen", True):
        return _screened(p, short)
    order = opts.get("start_order", 0)
    order = ORDER_NAMES.get(order, order)
    if order == 0:                 # the default: the window first for every variant; x0 first for a single-start call
        order = 3 if opts.get("single_start") else 1          # (csrc/obca_device.h: OBCA_EFFECTIVE_ORDER)
    kinds = START_ORDERS[order]
    if opts.get("single_start"):
        kinds = kinds[:1]
    max_v = opts.get("max_iter", options_for(p.variant)["max_iter"])
    rho0 = opts.get("rho", DEFAULTS["rho"])
    pat = opts.get("patience") or patience(p.N)
    ret = opts.get("retry_iter") or retry_iter(p.N)

    def run(s, kind, rho):
        cap = max_v if len(kinds) == 1 else min(max_v, pat if s == 0 else ret)
        o = dict(opts, rho=rho, max_iter=cap)
        if kind == KIND_WINDOW:
            return _solve_once(p, dict(o, mu_init=RESTART_MU), trace, x_start=window_start(p))
        if kind in (KIND_DODGE_R, KIND_DODGE_L, KIND_DODGE_R2, KIND_DODGE_L2):
            return _solve_once(p, dict(o, mu_init=RESTART_MU if kind <= KIND_DODGE_L else DODGE_LEVEL2_MU, max_iter=min(max_v, ret)), trace,
                               x_start=dodge_start(p, -1.0 if kind in (KIND_DODGE_R, KIND_DODGE_R2) else 1.0))
        return _solve_once(p, o, trace, x_start=x0_start(p) if kind == KIND_X0 else None)

    # WHICH pass's answer an exhausted ladder returns (status, iterate, objective; round 6 -- until then: the last pass's): a pass
    # that ends at a feasible point always; otherwise the first one, replaced by a later one only if that one CONVERGED
    # (STATUS_INFEASIBLE: a stationary point of the penalty problem with elastic variables left -- a statement about the problem)
    # where the held one did not (iteration limit, line-search failure: a statement about the solver), or if it is the same
    # start's repetition with a raised penalty (the sharper statement of the same kind).  Iteration counts: the whole sequence.
    r = held = None
    it = nf = 0
    for s, kind in enumerate(kinds):
        if r is not None and r.status in (STATUS_OK, STATUS_ACCEPTABLE, STATUS_BAD_BOUNDS):
            break
        for level in range(1 + len(RHO_ESCALATION)):
            if level > 0 and not (r.status == STATUS_INFEASIBLE and p.variant == 4 and not opts.get("no_escalation")):
                break
            r = run(s, kind, rho0 * (RHO_ESCALATION[level - 1] if level else 1.0))
            it, nf = it + r.iters, nf + getattr(r, "nfact", 0)
            r.start_index = s
            if _replaces(r, held):
                held = r
        starts_used = s + 1
    r = held
    r.iters, r.nfact, r.starts_used = it, nf, starts_used
    r.restarted = r.starts_used > 1
    # the dodge rung: fixed-time problems only, after the order is exhausted; both sides run, the feasible answer with the lower
    # objective stays (a failed rung leaves the answer of the order's last pass, with the iterations added)
    # (obca_mpc6: only with room to dodge -- a detour of lateral offset d over a path of length L is about 2 d^2 / L longer, and
    # with less than DODGE_MIN_SPARE of reach beyond the terminal set no offset that clears anything fits: the closed loop's
    # terminal set x0 + 5 m is 1e-9 m inside the reach while the car runs straight at full speed)
    if p.variant != 4 and opts.get("dodge", True) and short < -DODGE_MIN_SPARE and r.status not in (STATUS_OK, STATUS_ACCEPTABLE, STATUS_BAD_BOUNDS):
        best = None
        it, nf = r.iters, getattr(r, "nfact", 0)
        for level in ((KIND_DODGE_R, KIND_DODGE_L), (KIND_DODGE_R2, KIND_DODGE_L2)):
            if best is not None or (level[0] == KIND_DODGE_R2 and p.variant != 8):
                break
            for kind in level:
                d = run(1, kind, rho0)
                it, nf = it + d.iters, nf + getattr(d, "nfact", 0)
                if d.status in (STATUS_OK, STATUS_ACCEPTABLE) and (best is None or d.f < best.f):
                    best = d
        if best is not None:
            best.starts_used, best.restarted = r.starts_used, True
            r = best
        r.iters, r.nfact, r.dodged = it, nf, best is not None
    return r


def _replaces(new, held):
    """the rule of solve(): does the pass `new` take the place of the answer `held` (csrc/obca_device.h: OBCA_LADDER_REPLACES)"""
    if held is None or new.status in (STATUS_OK, STATUS_ACCEPTABLE, STATUS_BAD_BOUNDS):
        return True
    return new.status == STATUS_INFEASIBLE and (held.status != STATUS_INFEASIBLE or held.start_index == new.start_index)


def _solve_once(p, opts=None, trace=None, x_start=None):
    o = options_for(p.variant)
    if opts:
        o.update(opts)
    n = p.n
    x = p.start_point() if x_start is None else np.array(x_start, float)
    hard, term = split_rows(p)
    lbI, ubI = p.ineq_bounds()
    nt = term.size
    # elastic rows: terminal equalities first (no slack), then the inequalities
    lb = np.concatenate([np.zeros(nt), lbI])
    ub = np.concatenate([np.zeros(nt), ubI])
    iseq = np.concatenate([np.ones(nt, bool), np.zeros(lbI.size, bool)])
    hasL = np.isfinite(lb) & ~iseq
    hasU = np.isfinite(ub) & ~iseq
    res = Result()
    if np.any(lb[hasL & hasU] >= ub[hasL & hasU]):
        res.status, res.iters, res.x = STATUS_BAD_BOUNDS, 0, x
        xs, us = p.unpack_xu(x)
        res.xopt, res.uopt, res.Ts_opt, res.feas = xs, us, p.Ts, False
        return res
    me = lb.size
    mh = hard.size
    onlyL = hasL & ~hasU
    onlyU = hasU & ~hasL
    kd = o["kappa_d"]

    g0 = p.objective(x, grad=True)[1]
    # IPOPT's gradient-based scaling applied to the elastic objective f + rho*sum(p+n): its gradient holds rho
    gmax = max(np.max(np.abs(g0)), o["rho"])
    sf = o["nlp_scaling_max_gradient"] / gmax if gmax > o["nlp_scaling_max_gradient"] else 1.0
    rho = o["rho"] * sf

    def rows(x, jac=False):
        """hard residuals ch, elastic row functions ge (and Jacobians)"""
        if jac:
            c, Jc = p.eq(x, jac=True)
            d, Jd = p.ineq(x, jac=True)
            return c[hard], np.concatenate([c[term], d]), Jc[hard], np.vstack([Jc[term], Jd])
        c = p.eq(x)
        d = p.ineq(x)
        return c[hard], np.concatenate([c[term], d])

    def evals(x):
        f, g = p.objective(x, grad=True)
        ch, ge, Jh, Je = rows(x, jac=True)
        return sf * f, sf * g, ch, ge, Jh, Je

    f, g, ch, ge, Jh, Je = evals(x)
    # slacks with bound push
    s = ge.copy()
    k1, k2 = o["bound_push"], o["bound_frac"]
    for i in range(me):
        if iseq[i]:
            s[i] = 0.0
        elif hasL[i] and hasU[i]:
            pL = min(k1 * max(1.0, abs(lb[i])), k2 * (ub[i] - lb[i]))
            pU = min(k1 * max(1.0, abs(ub[i])), k2 * (ub[i] - lb[i]))
            s[i] = min(max(s[i], lb[i] + pL), ub[i] - pU)
        elif hasL[i]:
            s[i] = max(s[i], lb[i] + k1 * max(1.0, abs(lb[i])))
        elif hasU[i]:
            s[i] = min(s[i], ub[i] - k1 * max(1.0, abs(ub[i])))
    mu = o["mu_init"]
    # elastic variables on the central path of their own 1-d problem (IPOPT restoration initialisation)
    r = ge - s
    a = (mu - rho * r) / (2 * rho)
    en = a + np.sqrt(a * a + mu * r / (2 * rho))
    ep = r + en
    zp = mu / ep
    zn = mu / en
    ye = rho - zp
    yh = np.zeros(mh)
    zL = np.where(hasL, 1.0, 0.0)
    zU = np.where(hasU, 1.0, 0.0)
    nz = int(np.sum(hasL) + np.sum(hasU)) + 2 * me

    def barrier(f, s, ep, en, mu):
        v = f + rho * (np.sum(ep) + np.sum(en)) - mu * (np.sum(np.log(ep)) + np.sum(np.log(en)))
        v -= mu * np.sum(np.log(s[hasL] - lb[hasL]))
        v -= mu * np.sum(np.log(ub[hasU] - s[hasU]))
        v += kd * mu * (np.sum(s[onlyL] - lb[onlyL]) + np.sum(ub[onlyU] - s[onlyU]))
        return v

    def theta_of(ch, ge, s, ep, en):
        return np.sum(np.abs(ch)) + np.sum(np.abs(ge - s - ep + en))

    def errors(g, ch, ge, Jh, Je, s, ep, en, yh, ye, zL, zU, zp, zn, mu):
        rx = g + Jh.T @ yh + Je.T @ ye
        rs = np.where(iseq, 0.0, -ye - zL + zU)
        rp = rho - ye - zp
        rn = rho + ye - zn
        dual = max(np.max(np.abs(rx)), np.max(np.abs(rs)), np.max(np.abs(rp)), np.max(np.abs(rn)))
        prim = max(np.max(np.abs(ch)) if mh else 0.0, np.max(np.abs(ge - s - ep + en)))
        comp = max(np.max(np.abs(ep * zp - mu)), np.max(np.abs(en * zn - mu)))
        if np.any(hasL):
            comp = max(comp, np.max(np.abs((s[hasL] - lb[hasL]) * zL[hasL] - mu)))
        if np.any(hasU):
            comp = max(comp, np.max(np.abs((ub[hasU] - s[hasU]) * zU[hasU] - mu)))
        zsum = np.sum(zL) + np.sum(zU) + np.sum(zp) + np.sum(zn)
        sd = max(o["s_max"], (np.sum(np.abs(yh)) + np.sum(np.abs(ye)) + zsum) / (mh + me + nz)) / o["s_max"]
        sc = max(o["s_max"], zsum / nz) / o["s_max"]
        return max(dual / sd, prim, comp / sc), dual, prim, comp

    filt = []
    th0 = theta_of(ch, ge, s, ep, en)
    theta_max = o["theta_max_fact"] * max(1.0, th0)
    theta_min = o["theta_min_fact"] * max(1.0, th0)
    delta_w_last = 0.0
    tau = max(o["tau_min"], 1.0 - mu)
    acc_count = 0
    f_prev = None
    status = STATUS_MAXITER
    nfact = 0
    it = 0
    E0 = np.inf
    for it in range(o["max_iter"] + 1):
        E0, dual, prim, comp = errors(g, ch, ge, Jh, Je, s, ep, en, yh, ye, zL, zU, zp, zn, 0.0)
        if trace is not None:
            trace.append(dict(it=it, f=f / sf, E0=E0, dual=dual, prim=prim, comp=comp, mu=mu, x=x.copy(),
                              dw=delta_w_last, pn=float(np.max(ep + en))))
        if E0 <= o["tol"] and dual <= o["dual_inf_tol"] and prim <= o["constr_viol_tol"] \
                and comp <= o["compl_inf_tol"]:
            status = STATUS_OK
            break
        fobj = f + rho * (np.sum(ep) + np.sum(en))
        objchg = abs(fobj - f_prev) / max(1.0, abs(fobj)) if f_prev is not None else np.inf
        if E0 <= o["acceptable_tol"] and dual <= o["acceptable_dual_inf_tol"] and \
                prim <= o["acceptable_constr_viol_tol"] and comp <= o["acceptable_compl_inf_tol"] and \
                objchg <= o["acceptable_obj_change_tol"]:
            acc_count += 1
            if acc_count >= o["acceptable_iter"]:
                status = STATUS_ACCEPTABLE
                break
        else:
            acc_count = 0
        if it == o["max_iter"]:
            break
        # barrier parameter update (monotone, possibly several decreases at once)
        mu_floor = o["tol"] / (o["kappa_eps"] + 1.0)
        while mu > mu_floor:
            Emu = errors(g, ch, ge, Jh, Je, s, ep, en, yh, ye, zL, zU, zp, zn, mu)[0]
            if Emu > o["kappa_eps"] * mu:
                break
            mu = max(mu_floor, min(o["kappa_mu"] * mu, mu ** o["theta_mu"]))
            tau = max(o["tau_min"], 1.0 - mu)
            filt = []
        # ---- search direction ---------------------------------------------------
        c_all = np.zeros(len(p.eq_layout()))
        c_all[hard] = yh
        c_all[term] = ye[:nt]
        W = sf * p.objective(x, hess=True)[1] + p.eq(x, hess_y=c_all)[1] + p.ineq(x, hess_y=ye[nt:])[1]
        sL = np.where(hasL, s - lb, 1.0)
        sU = np.where(hasU, ub - s, 1.0)
        Sig_s = np.where(hasL, zL / sL, 0.0) + np.where(hasU, zU / sU, 0.0)
        Sig_p = zp / ep
        Sig_n = zn / en
        gs = -np.where(hasL, mu / sL, 0.0) + np.where(hasU, mu / sU, 0.0) + kd * mu * (onlyL * 1.0 - onlyU * 1.0)
        r_s = np.where(iseq, 0.0, -ye + gs)
        r_p = rho - ye - mu / ep
        r_n = rho + ye - mu / en
        r_x = g + Jh.T @ yh + Je.T @ ye
        r_g = ge - s - ep + en
        cond = ~iseq                      # rows condensed into the Hessian; terminal rows stay bordered

        def build(delta_w, rg_rhs, ch_rhs):
            Ds = Sig_s + delta_w
            Dp = Sig_p + delta_w
            Dn = Sig_n + delta_w
            E = np.where(iseq, 0.0, 1.0 / np.where(iseq, 1.0, Ds)) + 1.0 / Dp + 1.0 / Dn
            ghat = rg_rhs + np.where(iseq, 0.0, r_s / np.where(iseq, 1.0, Ds)) + r_p / Dp - r_n / Dn
            H = W + delta_w * np.eye(n)
            if p.variant == 4:            # the N+1 tied Topt copies each carry delta_w
                H[n - 1, n - 1] += delta_w * p.N
            Jc_ = Je[cond]
            H = H + Jc_.T @ (Jc_ / E[cond][:, None])
            rhs_x = -(r_x + Jc_.T @ (ghat[cond] / E[cond]))
            Jt = Je[iseq]
            K = np.zeros((n + mh + nt, n + mh + nt))
            K[:n, :n] = H
            K[:n, n:n + mh] = Jh.T
            K[n:n + mh, :n] = Jh
            K[:n, n + mh:] = Jt.T
            K[n + mh:, :n] = Jt
            K[n + mh:, n + mh:] = -np.diag(E[iseq])
            rhs = np.concatenate([rhs_x, -ch_rhs, -ghat[iseq]])
            return K, rhs, (Ds, Dp, Dn, E, ghat)

        def back(K, rhs, aux):
            Ds, Dp, Dn, E, ghat = aux
            sol = np.linalg.solve(K, rhs)
            dx = sol[:n]
            dyh = sol[n:n + mh]
            dye = (Je @ dx + ghat) / E
            dye[iseq] = sol[n + mh:]
            ds = np.where(iseq, 0.0, (dye - r_s) / np.where(iseq, 1.0, Ds))
            dp = (dye - r_p) / Dp
            dn = (-dye - r_n) / Dn
            return dx, ds, dp, dn, dyh, dye

        delta_w = 0.0
        K, rhs, aux = build(0.0, r_g, ch)
        pos, neg = _ldl_inertia(K)
        nfact += 1
        if not (pos == n and neg == mh + nt):
            delta_w = o["delta_w_0"] if delta_w_last == 0.0 else max(o["delta_w_min"],
                                                                      o["kappa_w_minus"] * delta_w_last)
            while True:
                K, rhs, aux = build(delta_w, r_g, ch)
                pos, neg = _ldl_inertia(K)
                nfact += 1
                if pos == n and neg == mh + nt:
                    break
                delta_w *= o["kappa_w_plus_bar"] if delta_w_last == 0.0 else o["kappa_w_plus"]
                if delta_w > o["delta_w_max"]:
                    status = STATUS_NUMERIC
                    break
            if status == STATUS_NUMERIC:
                break
            delta_w_last = delta_w
        dx, ds, dp, dn, dyh, dye = back(K, rhs, aux)
        if o.get("probe") is not None:
            o["probe"](dict(it=it, W=W, Je=Je, Jh=Jh, E=aux[3], ghat=aux[4], r_x=r_x, ch=ch, iseq=iseq, dx=dx,
                            dyh=dyh, dye=dye, delta_w=delta_w, hard=hard, term=term, mu=mu))
        dzL = np.where(hasL, (mu - zL * ds) / sL - zL, 0.0)
        dzU = np.where(hasU, (mu + zU * ds) / sU - zU, 0.0)
        dzp = (mu - zp * dp) / ep - zp
        dzn = (mu - zn * dn) / en - zn

        def ftb(vals, steps, mask=None):
            m = steps < 0
            if mask is not None:
                m = m & mask
            return np.min(-tau * vals[m] / steps[m]) if np.any(m) else 1.0

        def ftb_primal(ds_, dp_, dn_):
            return min(1.0, ftb(sL, ds_, hasL), ftb(sU, -ds_, hasU), ftb(ep, dp_), ftb(en, dn_))

        a_max = ftb_primal(ds, dp, dn)
        if o.get("verbose", 0) > 1:
            names = [("term", i) for i in range(nt)] + p.ineq_layout()
            for nm, vals, steps, mask in (("sL", sL, ds, hasL), ("sU", sU, -ds, hasU), ("p", ep, dp, None), ("n", en, dn, None)):
                m = steps < 0
                if mask is not None:
                    m = m & mask
                if np.any(m):
                    rat = np.where(m, -tau * vals / np.where(m, steps, -1.0), np.inf)
                    i = int(np.argmin(rat))
                    print("    ftb %s: row %s val %.3e step %.3e ratio %.3e" % (nm, names[i], vals[i], steps[i], rat[i]))
        a_z = min(1.0, ftb(zL, dzL, hasL), ftb(zU, dzU, hasU), ftb(zp, dzp), ftb(zn, dzn))

        # ---- filter line search ------------------------------------------------
        th = theta_of(ch, ge, s, ep, en)
        phi = barrier(f, s, ep, en, mu)
        dphi = g @ dx + gs @ ds + (rho - mu / ep) @ dp + (rho - mu / en) @ dn
        if dphi < 0:
            cand = [o["gamma_theta"], o["gamma_phi"] * th / (-dphi)]
            if th <= theta_min:
                cand.append(o["delta"] * th ** o["s_theta"] / (-dphi) ** o["s_phi"])
            alpha_min = o["gamma_alpha"] * min(cand)
        else:
            alpha_min = o["gamma_alpha"] * o["gamma_theta"]

        def in_filter(th_t, phi_t):
            if th_t >= theta_max:
                return True
            return any(th_t >= tf and phi_t >= pf for (tf, pf) in filt)

        def acceptable(alpha, th_t, phi_t):
            if not np.isfinite(phi_t) or in_filter(th_t, phi_t):
                return False, False
            switching = dphi < 0 and alpha * (-dphi) ** o["s_phi"] > o["delta"] * th ** o["s_theta"]
            if th <= theta_min and switching:
                return (phi_t <= phi + o["eta_phi"] * alpha * dphi + 10 * np.finfo(float).eps * abs(phi)), False
            return (th_t <= (1 - o["gamma_theta"]) * th or phi_t <= phi - o["gamma_phi"] * th), True

        def trial(alpha, dx_, ds_, dp_, dn_):
            xt = x + alpha * dx_
            st = s + alpha * ds_
            pt = ep + alpha * dp_
            nt_ = en + alpha * dn_
            ft = sf * p.objective(xt)
            cht, get = rows(xt)
            if not (np.isfinite(ft) and np.all(np.isfinite(cht)) and np.all(np.isfinite(get))):
                return (xt, st, pt, nt_), np.inf, np.inf, None
            return (xt, st, pt, nt_), theta_of(cht, get, st, pt, nt_), barrier(ft, st, pt, nt_, mu), (cht, get)

        alpha = a_max
        accepted = False
        first_trial = True
        aug = False
        step = None
        while True:
            pt_, th_t, phi_t, ev = trial(alpha, dx, ds, dp, dn)
            ok_, aug = acceptable(alpha, th_t, phi_t)
            if ok_:
                accepted = True
                step = (alpha, dx, ds, dp, dn, dyh, dye)
                break
            if first_trial and th_t >= th and ev is not None and o["max_soc"] > 0:
                # second-order correction
                res.soc_tried = getattr(res, "soc_tried", 0) + 1
                c_soc = alpha * ch + ev[0]
                g_soc = alpha * r_g + (ev[1] - pt_[1] - pt_[2] + pt_[3])
                th_old = th_t
                for _ in range(o["max_soc"]):
                    Ks, rs_, auxs = build(delta_w, g_soc, c_soc)
                    dxs, dss, dps, dns, dyhs, dyes = back(Ks, rs_, auxs)
                    a_soc = ftb_primal(dss, dps, dns)
                    pts, th_s, phi_s, evs = trial(a_soc, dxs, dss, dps, dns)
                    ok_s, aug_s = acceptable(alpha, th_s, phi_s)
                    if ok_s:
                        res.soc_accepted = getattr(res, "soc_accepted", 0) + 1
                        accepted = True
                        pt_, th_t, phi_t, ev, aug = pts, th_s, phi_s, evs, aug_s
                        step = (a_soc, dxs, dss, dps, dns, dyhs, dyes)
                        break
                    if evs is None or th_s > o["kappa_soc"] * th_old:
                        break
                    th_old = th_s
                    c_soc = a_soc * c_soc + evs[0]
                    g_soc = a_soc * g_soc + (evs[1] - pts[1] - pts[2] + pts[3])
                if accepted:
                    break
            first_trial = False
            alpha *= 0.5
            if alpha < alpha_min:
                break
        if o.get("verbose"):
            print("it %3d f %.6e pn %.2e th %.2e phi %.6e dphi %.2e mu %.1e dw %.1e a %.2e az %.2e |dx| %.2e E0 %.2e %s"
                  % (it, f / sf, np.max(ep + en), th, phi, dphi, mu, delta_w, step[0] if accepted else -1, a_z,
                     np.max(np.abs(dx)), E0, "" if accepted else "LS-FAIL"))
        if not accepted:
            status = STATUS_LINESEARCH
            break
        if aug:
            tn, pn_ = (1 - o["gamma_theta"]) * th, phi - o["gamma_phi"] * th
            filt = [(tf, pf) for (tf, pf) in filt if not (tf >= tn and pf >= pn_)]   # drop dominated entries
            filt.append((tn, pn_))               # unbounded, as IPOPT's (the kernels hold 64 / 128 entries: OBCA_FILTER_CAP)
            res.max_filter = max(getattr(res, "max_filter", 0), len(filt))
        a_used = step[0]
        f_prev = fobj
        x, s, ep, en = pt_
        yh = yh + a_used * step[5]
        ye = ye + a_used * step[6]
        zL = zL + a_z * dzL
        zU = zU + a_z * dzU
        zp = zp + a_z * dzp
        zn = zn + a_z * dzn
        sL = np.where(hasL, s - lb, 1.0)
        sU = np.where(hasU, ub - s, 1.0)
        ks = o["kappa_sigma"]
        zL = np.where(hasL, np.maximum(np.minimum(zL, ks * mu / sL), mu / (ks * sL)), 0.0)
        zU = np.where(hasU, np.maximum(np.minimum(zU, ks * mu / sU), mu / (ks * sU)), 0.0)
        zp = np.maximum(np.minimum(zp, ks * mu / ep), mu / (ks * ep))
        zn = np.maximum(np.minimum(zn, ks * mu / en), mu / (ks * en))
        f, g, ch, ge, Jh, Je = evals(x)
    res.iters = it
    res.nfact = nfact
    res.x = x
    res.f = p.objective(x)
    res.mu = mu
    res.E0 = E0
    res.elastic = float(np.max(ep + en))
    res.yh, res.ye = yh / sf, ye / sf
    if status in (STATUS_OK, STATUS_ACCEPTABLE) and res.elastic > o["feas_tol"]:
        status = STATUS_INFEASIBLE
    res.status = status
    res.feas = status in (STATUS_OK, STATUS_ACCEPTABLE)
    xs, us = p.unpack_xu(x)
    res.xopt, res.uopt = xs, us
    res.Ts_opt = (x[p.iT()] * p.Ts) if p.variant == 4 else p.Ts
    return res


def kkt_certificate(p, res):
    """Residuals of the ORIGINAL NLP's first-order conditions at res (multipliers from the elastic solve)."""
    hard, term = split_rows(p)
    x = res.x
    g = p.objective(x, grad=True)[1]
    c, Jc = p.eq(x, jac=True)
    d, Jd = p.ineq(x, jac=True)
    lb, ub = p.ineq_bounds()
    y = np.zeros(c.size)
    y[hard] = res.yh
    y[term] = res.ye[:term.size]
    yd = res.ye[term.size:]
    stat = np.max(np.abs(g + Jc.T @ y + Jd.T @ yd))
    prim = max(np.max(np.abs(c)), np.max(np.maximum(lb - d, 0.0)), np.max(np.maximum(d - ub, 0.0)))
    slackL = np.where(np.isfinite(lb), d - lb, np.inf)
    slackU = np.where(np.isfinite(ub), ub - d, np.inf)
    comp = np.max(np.abs(yd) * np.minimum(slackL, slackU))
    return dict(stationarity=float(stat), primal=float(prim), complementarity=float(comp))
