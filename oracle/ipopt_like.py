"""ORACLE (test infrastructure, never on the product path; NOT the product's algorithm): IPOPT's published algorithm restated from
the paper -- Waechter & Biegler, "On the implementation of an interior-point filter line-search algorithm for large-scale
nonlinear programming", Math. Prog. 106 (2006), sections 2.1-3.3, with the option defaults of IPOPT 3.12-3.14 -- applied to the
reference's NLP AS THE REFERENCE POSES IT (src/obca.py:1044-1056: ``opti.solver('ipopt')``, no options besides print_level):

    min f(x)   s.t.  c(x) = 0  (initial state, dynamics, terminal pose, rotation equalities -- ALL hard),
                     d_L <= d(x) <= d_U  as  d(x) - s = 0,  d_L <= s <= d_U  (every inequality, simple bounds included: CasADi's Opti
                     hands them to IPOPT as general constraints), x free,

from the reference's literal start (every variable 0, Topt = 1; src/obca.py:856).  Unlike oracle/ipm_dense.py -- the specification of
the PRODUCT's method, the l1-elastic form kept on throughout with a ladder of starts -- nothing here is shared with the kernels:
hard equalities, slack bounds, least-squares multiplier initialisation, inertia correction with delta_c, and a FEASIBILITY
RESTORATION PHASE that moves (section 3.3: min rho ||(p, n)||_1 + zeta / 2 ||D_R (v - v_R)||^2 over the relaxed constraints by the same
interior-point method, left as soon as the original filter accepts a point with 90 % of the infeasibility).  Round 4's experiment
with hard equalities inside ipm_dense.py ended where IPOPT enters restoration; this file has the missing piece.

What it is for: an oracle that cannot share a mistake with the product (round-5 review).  tests/test_ipopt_like.py runs it on the
known answers of SURVEY.md Appendix C and on steps of the reference-held runs; tools/ipopt_like_study.py produces the table "steps of
the reference's own runs matched from the reference's own start" quoted in DESIGN.md.

Not restated (IPOPT defaults that stay off the path of these problems or are implementation detail): the watchdog procedure,
tiny-step handling, the quality-function mu oracle (mu_strategy is monotone by default), MUMPS' pivoting (dense Bunch-Kaufman
here: inertia from the D blocks).  Dense linear algebra, Python loops in the model functions: seconds per solve.
"""
import math

import numpy as np
import scipy.linalg

OPT = dict(
    tol=1e-8, max_iter=3000, dual_inf_tol=1.0, constr_viol_tol=1e-4, compl_inf_tol=1e-4,
    acceptable_tol=1e-6, acceptable_iter=15, acceptable_dual_inf_tol=1e10, acceptable_constr_viol_tol=1e-2,
    acceptable_compl_inf_tol=1e-2, acceptable_obj_change_tol=1e20,
    mu_init=0.1, kappa_mu=0.2, theta_mu=1.5, kappa_eps=10.0, tau_min=0.99, mu_min=1e-11,
    bound_push=1e-2, bound_frac=1e-2, bound_relax_factor=1e-8, bound_mult_init_val=1.0, constr_mult_init_max=1e3,
    kappa_sigma=1e10, s_max=100.0, kappa_d=1e-5,
    gamma_theta=1e-5, gamma_phi=1e-8, delta=1.0, s_theta=1.1, s_phi=2.3, eta_phi=1e-8, gamma_alpha=0.05,
    kappa_soc=0.99, max_soc=4, theta_max_fact=1e4, theta_min_fact=1e-4,
    delta_w_min=1e-20, delta_w_0=1e-4, delta_w_max=1e40, kappa_w_plus=8.0, kappa_w_plus_bar=100.0, kappa_w_minus=1.0 / 3.0,
    delta_c_bar=1e-8, kappa_c=0.25,
    nlp_scaling_max_gradient=100.0,
    resto_rho=1000.0, kappa_resto=0.9, bound_mult_reset_threshold=1000.0, constr_mult_reset_threshold=0.0,
)

OK, ACCEPTABLE, INFEASIBLE, MAXITER, RESTO_FAILED, NUMERIC = 0, 1, 2, -1, -2, -3
STATUS_NAMES = {0: "Solve_Succeeded", 1: "Solved_To_Acceptable_Level", 2: "Infeasible_Problem_Detected", -1: "Maximum_Iterations_Exceeded",
                -2: "Restoration_Failed", -3: "Error_In_Step_Computation"}


class Result:
    pass


# ------------------------------------------------------------------------------------------------------ problems in IPOPT's internal form
class SlackForm:
    """v = (x, s):  min f(x)  s.t.  C(v) = [c(x); d(x) - s] = 0,  vL <= v <= vU  (bounds on the slacks only), with IPOPT's
    gradient-based scaling of the objective and of every constraint row, fixed at the starting point"""

    def __init__(self, p, x0, o):
        self.p = p
        self.nx = p.n
        dL, dU = p.ineq_bounds()
        rel = o["bound_relax_factor"]
        self.dL = np.where(np.isfinite(dL), dL - rel * np.maximum(1.0, np.abs(dL)), dL)
        self.dU = np.where(np.isfinite(dU), dU + rel * np.maximum(1.0, np.abs(dU)), dU)
        self.md = dL.size
        g0 = p.objective(x0, grad=True)[1]
        gmax = o["nlp_scaling_max_gradient"]
        self.sf = min(1.0, gmax / np.max(np.abs(g0))) if np.max(np.abs(g0)) > gmax else 1.0
        Jc = p.eq(x0, jac=True)[1]
        Jd = p.ineq(x0, jac=True)[1]
        rowmax = lambda J: np.max(np.abs(J), axis=1) if J.size else np.zeros(0)
        self.sc = np.where(rowmax(Jc) > gmax, gmax / np.maximum(rowmax(Jc), 1e-300), 1.0)
        self.sd = np.where(rowmax(Jd) > gmax, gmax / np.maximum(rowmax(Jd), 1e-300), 1.0)
        self.mc = self.sc.size
        self.n = self.nx + self.md
        self.m = self.mc + self.md
        self.vL = np.concatenate([np.full(self.nx, -np.inf), self.dL * self.sd])
        self.vU = np.concatenate([np.full(self.nx, np.inf), self.dU * self.sd])

    def slack_of(self, x):
        return self.p.ineq(x) * self.sd

    def f(self, v):
        return self.sf * self.p.objective(v[:self.nx])

    def grad(self, v):
        g = np.zeros(self.n)
        g[:self.nx] = self.sf * self.p.objective(v[:self.nx], grad=True)[1]
        return g

    def c(self, v):
        x = v[:self.nx]
        return np.concatenate([self.p.eq(x) * self.sc, self.p.ineq(x) * self.sd - v[self.nx:]])

    def jac(self, v):
        x = v[:self.nx]
        J = np.zeros((self.m, self.n))
        J[:self.mc, :self.nx] = self.p.eq(x, jac=True)[1] * self.sc[:, None]
        J[self.mc:, :self.nx] = self.p.ineq(x, jac=True)[1] * self.sd[:, None]
        J[self.mc:, self.nx:] = -np.eye(self.md)
        return J

    def hess(self, v, y, sigma=1.0):
        x = v[:self.nx]
        H = np.zeros((self.n, self.n))
        Hx = sigma * self.sf * self.p.objective(x, grad=True, hess=True)[2] if sigma != 0.0 else np.zeros((self.nx, self.nx))
        Hx = Hx + self.p.eq(x, hess_y=y[:self.mc] * self.sc)[1] + self.p.ineq(x, hess_y=y[self.mc:] * self.sd)[1]
        H[:self.nx, :self.nx] = Hx
        return H


class RestoForm:
    """the restoration problem of section 3.3 for a problem `base` in the form above, at the point vR with barrier parameter mu:
    w = (v, p, n):  min rho sum(p + n) + zeta / 2 ||D_R (v - vR)||^2  s.t.  C(v) - p + n = 0,  vL <= v <= vU,  p, n >= 0"""

    def __init__(self, base, vR, mu, o):
        self.b, self.vR = base, vR.copy()
        self.rho, self.zeta = o["resto_rho"], math.sqrt(mu)
        self.DR2 = np.minimum(1.0, 1.0 / np.maximum(np.abs(vR), 1e-300)) ** 2
        self.nv, self.m = base.n, base.m
        self.n = self.nv + 2 * self.m
        self.vL = np.concatenate([base.vL, np.zeros(2 * self.m)])
        self.vU = np.concatenate([base.vU, np.full(2 * self.m, np.inf)])

    def split(self, w):
        return w[:self.nv], w[self.nv:self.nv + self.m], w[self.nv + self.m:]

    def f(self, w):
        v, p, n = self.split(w)
        return self.rho * (p.sum() + n.sum()) + 0.5 * self.zeta * np.sum(self.DR2 * (v - self.vR) ** 2)

    def grad(self, w):
        v, p, n = self.split(w)
        return np.concatenate([self.zeta * self.DR2 * (v - self.vR), np.full(2 * self.m, self.rho)])

    def c(self, w):
        v, p, n = self.split(w)
        return self.b.c(v) - p + n

    def jac(self, w):
        v, _, _ = self.split(w)
        return np.hstack([self.b.jac(v), -np.eye(self.m), np.eye(self.m)])

    def hess(self, w, y, sigma=1.0):
        v, _, _ = self.split(w)
        H = np.zeros((self.n, self.n))
        H[:self.nv, :self.nv] = self.b.hess(v, y, 0.0) + sigma * self.zeta * np.diag(self.DR2)
        return H


# ------------------------------------------------------------------------------------------------------ linear algebra
def _inertia(K):
    """(positive, negative, zero) eigenvalue counts from a Bunch-Kaufman LDL^T"""
    _, D, _ = scipy.linalg.ldl(K, lower=True)
    n = D.shape[0]
    pos = neg = zero = 0
    i = 0
    while i < n:
        if i + 1 < n and D[i + 1, i] != 0.0:
            a, b, c = D[i, i], D[i + 1, i], D[i + 1, i + 1]
            det, tr = a * c - b * b, a + c
            if det < 0:
                pos += 1
                neg += 1
            elif tr > 0:
                pos += 2
            else:
                neg += 2
            i += 2
        else:
            d = D[i, i]
            if d == 0.0 or not np.isfinite(d):
                zero += 1
            elif d > 0:
                pos += 1
            else:
                neg += 1
            i += 1
    return pos, neg, zero


class KKT:
    """the augmented system of (13) with the inertia correction of Algorithm IC; keeps delta_w of the last success"""

    def __init__(self, o):
        self.o, self.dw_last = o, 0.0

    def factor(self, W, Sigma, J, mu):
        o = self.o
        n, m = W.shape[0], J.shape[0]
        A = W + np.diag(Sigma)
        dw, dc = 0.0, 0.0
        tries = 0
        while True:
            K = np.zeros((n + m, n + m))
            K[:n, :n] = A + dw * np.eye(n)
            K[:n, n:] = J.T
            K[n:, :n] = J
            K[n:, n:] = -dc * np.eye(m)
            try:
                pos, neg, zero = _inertia(K)
            except Exception:      # noqa: BLE001
                pos, neg, zero = -1, -1, 1
            if pos == n and neg == m and zero == 0:
                self.K, self.dw, self.dc, self.n = K, dw, dc, n
                self.lu = scipy.linalg.lu_factor(K)
                if dw > 0.0:
                    self.dw_last = dw
                return True
            # "singular" (what the linear solver reports for a rank-deficient Jacobian: fewer than m negative eigenvalues, or exact
            # zeros): first delta_c alone, then delta_w as well; wrong inertia otherwise: delta_w
            if (neg < m or zero > 0) and dc == 0.0:
                dc = o["delta_c_bar"] * mu ** o["kappa_c"]
                continue
            if tries == 0:
                dw = o["delta_w_0"] if self.dw_last == 0.0 else max(o["delta_w_min"], o["kappa_w_minus"] * self.dw_last)
            else:
                dw *= o["kappa_w_plus_bar"] if self.dw_last == 0.0 else o["kappa_w_plus"]
            tries += 1
            if dw > o["delta_w_max"]:
                return False

    def solve(self, r1, r2):
        s = scipy.linalg.lu_solve(self.lu, np.concatenate([r1, r2]))
        return s[:self.n], s[self.n:]


def _least_squares_y(g, J):
    """multipliers minimising ||g + J'y||: [[I, J'], [J, 0]] [w; y] = -[g; 0]"""
    n, m = g.size, J.shape[0]
    K = np.zeros((n + m, n + m))
    K[:n, :n] = np.eye(n)
    K[:n, n:] = J.T
    K[n:, :n] = J
    try:
        s = np.linalg.lstsq(K, -np.concatenate([g, np.zeros(m)]), rcond=None)[0]
    except Exception:       # noqa: BLE001
        return np.zeros(m)
    return s[n:]


# ------------------------------------------------------------------------------------------------------ the method
def _push(v, vL, vU, o):
    hasL, hasU = np.isfinite(vL), np.isfinite(vU)
    both = hasL & hasU
    vL, vU = np.where(hasL, vL, 0.0), np.where(hasU, vU, 0.0)          # (placeholders where there is no bound: never used)
    pL = np.where(both, np.minimum(o["bound_push"] * np.maximum(1.0, np.abs(vL)), o["bound_frac"] * (vU - vL)), o["bound_push"] * np.maximum(1.0, np.abs(vL)))
    pU = np.where(both, np.minimum(o["bound_push"] * np.maximum(1.0, np.abs(vU)), o["bound_frac"] * (vU - vL)), o["bound_push"] * np.maximum(1.0, np.abs(vU)))
    v = np.where(hasL, np.maximum(v, np.where(hasL, vL + pL, v)), v)
    v = np.where(hasU, np.minimum(v, np.where(hasU, vU - pU, v)), v)
    return v


def _ipm(nlp, v, zL, zU, y, mu, o, it0, orig=None, trace=None):
    """Algorithm A on `nlp` from (v, zL, zU, y, mu).  orig = (base problem, its filter, theta_R, phi function of the original
    barrier problem) when this IS a restoration phase: it then ends as soon as a trial point's (x, s) part is acceptable to the
    original filter with theta <= kappa_resto theta_R.  Returns (status, v, zL, zU, y, mu, iterations used, info)."""
    hasL, hasU = np.isfinite(nlp.vL), np.isfinite(nlp.vU)
    oneL, oneU = hasL & ~hasU, hasU & ~hasL
    nb = int(hasL.sum() + hasU.sum())
    vL, vU = np.where(hasL, nlp.vL, 0.0), np.where(hasU, nlp.vU, 0.0)
    kd = o["kappa_d"]
    kkt = KKT(o)

    def phi(vv, mu_):
        sl, su = np.where(hasL, vv - vL, 1.0), np.where(hasU, vU - vv, 1.0)
        if np.any(sl <= 0) or np.any(su <= 0):
            return np.inf
        return nlp.f(vv) - mu_ * (np.log(sl)[hasL].sum() + np.log(su)[hasU].sum()) + kd * mu_ * (sl[oneL].sum() + su[oneU].sum())

    def theta(vv):
        return float(np.sum(np.abs(nlp.c(vv))))

    def errors(g, J, c, vv, zL_, zU_, y_, mu_):
        sl, su = np.where(hasL, vv - vL, 1.0), np.where(hasU, vU - vv, 1.0)
        m = c.size
        sd = max(o["s_max"], (np.abs(y_).sum() + zL_[hasL].sum() + zU_[hasU].sum()) / max(1, m + nb)) / o["s_max"]
        sc = max(o["s_max"], (zL_[hasL].sum() + zU_[hasU].sum()) / max(1, nb)) / o["s_max"]
        dual = float(np.max(np.abs(g + J.T @ y_ - zL_ + zU_)))
        prim = float(np.max(np.abs(c))) if m else 0.0
        comp = max(float(np.max(np.abs(sl * zL_ - mu_)[hasL], initial=0.0)), float(np.max(np.abs(su * zU_ - mu_)[hasU], initial=0.0)))
        return max(dual / sd, prim, comp / sc), dual, prim, comp

    th0 = theta(v)
    th_max, th_min = o["theta_max_fact"] * max(1.0, th0), o["theta_min_fact"] * max(1.0, th0)
    filt = []
    acc_count, f_prev = 0, None
    tau = max(o["tau_min"], 1.0 - mu)
    it = it0
    info = dict(restorations=0, resto_iters=0)
    while True:
        g, J, c = nlp.grad(v), nlp.jac(v), nlp.c(v)
        E0, dual, prim, comp = errors(g, J, c, v, zL, zU, y, 0.0)
        if trace is not None:
            trace.append(dict(it=it, resto=orig is not None, f=nlp.f(v), th=float(np.abs(c).sum()), mu=mu, E0=E0))
        if orig is None:
            if E0 <= o["tol"] and dual <= o["dual_inf_tol"] and prim <= o["constr_viol_tol"] and comp <= o["compl_inf_tol"]:
                return OK, v, zL, zU, y, mu, it, info
            fobj = nlp.f(v)
            chg = abs(fobj - f_prev) / max(1.0, abs(fobj)) if f_prev is not None else np.inf
            if E0 <= o["acceptable_tol"] and dual <= o["acceptable_dual_inf_tol"] and prim <= o["acceptable_constr_viol_tol"] and \
                    comp <= o["acceptable_compl_inf_tol"] and chg <= o["acceptable_obj_change_tol"]:
                acc_count += 1
                if acc_count >= o["acceptable_iter"]:
                    return ACCEPTABLE, v, zL, zU, y, mu, it, info
            else:
                acc_count = 0
            f_prev = fobj
        elif E0 <= o["tol"]:
            return INFEASIBLE, v, zL, zU, y, mu, it, info          # the restoration problem converged: a local minimiser of the infeasibility
        if it >= o["max_iter"]:
            return MAXITER, v, zL, zU, y, mu, it, info
        # ---- barrier parameter (7)
        while mu > o["mu_min"] and errors(g, J, c, v, zL, zU, y, mu)[0] <= o["kappa_eps"] * mu:
            mu = max(o["mu_min"], min(o["kappa_mu"] * mu, mu ** o["theta_mu"]))
            tau = max(o["tau_min"], 1.0 - mu)
            filt = []
        # ---- search direction (13), inertia correction
        sl, su = np.where(hasL, v - vL, 1.0), np.where(hasU, vU - v, 1.0)
        Sigma = np.where(hasL, zL / sl, 0.0) + np.where(hasU, zU / su, 0.0)
        W = nlp.hess(v, y, 1.0)
        gphi = g - np.where(hasL, mu / sl, 0.0) + np.where(hasU, mu / su, 0.0) + kd * mu * (oneL.astype(float) - oneU.astype(float))
        need_resto = not kkt.factor(W, Sigma, J, mu)
        th, ph = float(np.abs(c).sum()), phi(v, mu)
        accepted = False
        if not need_resto:
            dv, dy = kkt.solve(-(gphi + J.T @ y), -c)
            dzL = np.where(hasL, mu / sl - zL - zL / sl * dv, 0.0)
            dzU = np.where(hasU, mu / su - zU + zU / su * dv, 0.0)

            def frac(vv, d):
                a = 1.0
                neg, posm = hasL & (d < 0), hasU & (d > 0)
                if neg.any():
                    a = min(a, float(np.min(-tau * (vv - vL)[neg] / d[neg])))
                if posm.any():
                    a = min(a, float(np.min(tau * (vU - vv)[posm] / d[posm])))
                return a
            a_max = frac(v, dv)
            a_z = 1.0
            for z_, dz_, has in ((zL, dzL, hasL), (zU, dzU, hasU)):
                mneg = has & (dz_ < 0)
                if mneg.any():
                    a_z = min(a_z, float(np.min(-tau * z_[mneg] / dz_[mneg])))
            dphi = float(gphi @ dv)
            if dphi < 0:
                amin = min(o["gamma_theta"], o["gamma_phi"] * th / (-dphi))
                if th <= th_min:
                    amin = min(amin, o["delta"] * th ** o["s_theta"] / (-dphi) ** o["s_phi"])
            else:
                amin = o["gamma_theta"]
            amin *= o["gamma_alpha"]

            def acceptable(th_t, ph_t, alpha):
                """filter + sufficient progress w.r.t. the current iterate; -> (ok, augment the filter)"""
                if not (np.isfinite(th_t) and np.isfinite(ph_t)) or th_t >= th_max:
                    return False, False
                if any(th_t >= tf and ph_t >= pf for tf, pf in filt):
                    return False, False
                switching = dphi < 0 and alpha * (-dphi) ** o["s_phi"] > o["delta"] * th ** o["s_theta"]
                if th <= th_min and switching:
                    return ph_t <= ph + o["eta_phi"] * alpha * dphi + 10 * np.finfo(float).eps * abs(ph), False
                ok = th_t <= (1 - o["gamma_theta"]) * th or ph_t <= ph - o["gamma_phi"] * th
                return ok, ok
            alpha = a_max
            first = True
            step = None
            while alpha >= amin:
                vt = v + alpha * dv
                th_t, ph_t = theta(vt), phi(vt, mu)
                ok, aug = acceptable(th_t, ph_t, alpha)
                if ok:
                    step = (vt, alpha, dy, aug)
                    break
                if first and th_t >= th and o["max_soc"] > 0 and np.isfinite(th_t):      # second-order correction (A-5.5 .. A-5.9)
                    c_soc, th_old = alpha * c + nlp.c(vt), th_t
                    for _ in range(o["max_soc"]):
                        dvs, dys = kkt.solve(-(gphi + J.T @ y), -c_soc)
                        a_s = frac(v, dvs)
                        vs = v + a_s * dvs
                        th_s, ph_s = theta(vs), phi(vs, mu)
                        ok, aug = acceptable(th_s, ph_s, alpha)
                        if ok:
                            step = (vs, a_s, dys, aug)
                            break
                        if not np.isfinite(th_s) or th_s > o["kappa_soc"] * th_old:
                            break
                        c_soc, th_old = a_s * c_soc + nlp.c(vs), th_s
                    if step is not None:
                        break
                first = False
                alpha *= 0.5
            if step is not None:
                vt, a_used, dy_used, aug = step
                if aug:
                    tn, pn = (1 - o["gamma_theta"]) * th, ph - o["gamma_phi"] * th
                    filt = [(tf, pf) for tf, pf in filt if not (tf >= tn and pf >= pn)] + [(tn, pn)]
                v = vt
                y = y + a_used * dy_used
                zL, zU = zL + a_z * dzL, zU + a_z * dzU
                accepted = True
            else:
                need_resto = True
        if need_resto and not accepted:
            if orig is not None:
                return RESTO_FAILED, v, zL, zU, y, mu, it, info           # no restoration inside the restoration phase
            # ---- feasibility restoration phase (section 3.3)
            filt = [(tf, pf) for tf, pf in filt] + [((1 - o["gamma_theta"]) * th, ph - o["gamma_phi"] * th)]
            mu_r = max(mu, float(np.max(np.abs(c))))
            rn = RestoForm(nlp, v, mu_r, o)
            rho = rn.rho
            a = (mu_r - rho * c) / (2 * rho)
            n0 = a + np.sqrt(a * a + mu_r * c / (2 * rho))
            p0 = c + n0
            w = np.concatenate([v, p0, n0])
            zLr = np.concatenate([np.where(hasL, np.minimum(rho, zL), 0.0), mu_r / p0, mu_r / n0])
            zUr = np.concatenate([np.where(hasU, np.minimum(rho, zU), 0.0), np.zeros(2 * rn.m)])
            base_phi = lambda vv: phi(vv, mu)
            st, w, _, _, _, _, it_r, _ = _ipm(rn, w, zLr, zUr, np.zeros(rn.m), mu_r, o, it, orig=(nlp, list(filt), th, base_phi, th_max), trace=trace)
            info["restorations"] += 1
            info["resto_iters"] += it_r - it
            it = it_r
            if st != OK:
                return (INFEASIBLE if st == INFEASIBLE else st if st == MAXITER else RESTO_FAILED), v, zL, zU, y, mu, it, info
            v_new = rn.split(w)[0]
            # bound multipliers: a Newton step for complementarity with the change over the whole phase as the primal step;
            # all reset to 1 when the largest exceeds bound_mult_reset_threshold.  Constraint multipliers: constr_mult_reset_threshold = 0
            dvr = v_new - v
            zLn = np.where(hasL, zL + (mu / sl - zL - zL / sl * dvr), 0.0)
            zUn = np.where(hasU, zU + (mu / su - zU + zU / su * dvr), 0.0)
            if max(np.max(zLn, initial=0.0), np.max(zUn, initial=0.0)) > o["bound_mult_reset_threshold"] or np.any(zLn[hasL] <= 0) or np.any(zUn[hasU] <= 0):
                zLn, zUn = np.where(hasL, 1.0, 0.0), np.where(hasU, 1.0, 0.0)
            v, zL, zU = v_new, zLn, zUn
            y = np.zeros_like(y)
            continue
        # ---- after an accepted step: keep the bound multipliers within kappa_sigma of mu / slack (16)
        sl, su = np.where(hasL, v - vL, 1.0), np.where(hasU, vU - v, 1.0)
        ks = o["kappa_sigma"]
        zL = np.where(hasL, np.maximum(np.minimum(zL, ks * mu / sl), mu / (ks * sl)), 0.0)
        zU = np.where(hasU, np.maximum(np.minimum(zU, ks * mu / su), mu / (ks * su)), 0.0)
        it += 1
        if orig is not None:
            base, ofilt, thR, base_phi, th_max_o = orig
            vx = v[:base.n]
            th_o = float(np.sum(np.abs(base.c(vx))))
            if th_o <= o["kappa_resto"] * thR and th_o < th_max_o:
                ph_o = base_phi(vx)
                if np.isfinite(ph_o) and not any(th_o >= tf and ph_o >= pf for tf, pf in ofilt):
                    return OK, v, zL, zU, y, mu, it, info


def solve(p, x_start=None, opts=None, trace=None):
    """IPOPT's method on the reference's NLP `p` (oracle/obca_nlp.py: Problem) from x_start (default: the reference's literal
    start).  -> Result: status (0 Solve_Succeeded, 1 acceptable, 2 infeasible problem detected, < 0 failures), feas (what the
    reference's try / except sees: status in (0, 1)), x, xopt, uopt, Ts_opt, f, iters, restorations"""
    try:            # one BLAS thread: which optimum the method lands in from the zero start depends on ROUNDOFF on these problems (measured:
        from threadpoolctl import threadpool_limits      # the same code with 1 / 2 / 8 threads takes different optima at steps 4-5 of demo1)
    except ImportError:
        return _solve(p, x_start, opts, trace)
    with threadpool_limits(limits=1):
        return _solve(p, x_start, opts, trace)


def _solve(p, x_start, opts, trace):
    o = dict(OPT)
    if p.variant in (6, 8):          # src/obca.py:1538-1539, 1734-1735
        o.update(max_iter=1000, acceptable_tol=1e-8, acceptable_obj_change_tol=1e-6)
    if opts:
        o.update(opts)
    x0 = p.start_point() if x_start is None else np.asarray(x_start, float)[:p.n].copy()
    nlp = SlackForm(p, x0, o)
    v = _push(np.concatenate([x0, nlp.slack_of(x0)]), nlp.vL, nlp.vU, o)
    hasL, hasU = np.isfinite(nlp.vL), np.isfinite(nlp.vU)
    zL, zU = np.where(hasL, o["bound_mult_init_val"], 0.0), np.where(hasU, o["bound_mult_init_val"], 0.0)
    y = _least_squares_y(nlp.grad(v) - zL + zU, nlp.jac(v))
    if np.max(np.abs(y), initial=0.0) > o["constr_mult_init_max"]:
        y = np.zeros_like(y)
    st, v, zL, zU, y, mu, it, info = _ipm(nlp, v, zL, zU, y, o["mu_init"], o, 0, trace=trace)
    r = Result()
    r.status, r.status_name, r.feas, r.iters = st, STATUS_NAMES.get(st, str(st)), st in (OK, ACCEPTABLE), it
    r.restorations, r.resto_iters, r.mu = info["restorations"], info["resto_iters"], mu
    r.x = v[:p.n]
    r.f = float(p.objective(r.x))
    r.y = y
    r.xopt, r.uopt = p.unpack_xu(r.x)
    r.Ts_opt = (r.x[p.iT()] * p.Ts) if p.variant == 4 else p.Ts
    c, d = p.eq(r.x), p.ineq(r.x)
    lb, ub = p.ineq_bounds()
    r.viol = float(max(np.max(np.abs(c)), np.max(np.maximum(lb - d, 0.0)), np.max(np.maximum(d - ub, 0.0))))
    return r


def _solve_c2_instance(job):
    """worker of solve_c2_sample: (seeded C2 batch size B, horizon N, instance index) -> (status, Ts_opt, f, xopt, uopt, restorations)"""
    import warnings
    warnings.filterwarnings("ignore", category=RuntimeWarning)
    B, N, i = job[:3]
    gen = job[3] if len(job) > 3 else "c2"
    from tests import kkt_check
    from vehicle_motion_planning_with_obstacles_avoidance_using_mpc_amd import scenarios as sc
    p = kkt_check.problem_of(sc.make_batch(B, N) if gen == "c2" else sc.make_batch_c3(B, N, gated=(gen == "c3gated")), i, N)
    r = solve(p, opts=dict(max_iter=job[4]) if len(job) > 4 and job[4] else None)
    return int(r.status), float(r.Ts_opt), float(r.f), r.xopt, r.uopt, int(r.restorations)


def solve_c2_sample(B, N, idx, procs=1, gen="c2", max_iter=None):
    """instances `idx` of the seeded C2 batch (scenarios.make_batch(B, N); gen = "c3free": the free-time half of C3, make_batch_c3(B, N,
    gated=False); "c3gated": its gated half, obca_mpc6) by this file's method from the reference's zero start, in `procs` spawned processes (never a fork of a process that may
    hold a HIP context); returns the list of _solve_c2_instance tuples"""
    jobs = [(B, N, int(i), gen, max_iter) for i in idx]
    if procs <= 1:
        return [_solve_c2_instance(j) for j in jobs]
    import multiprocessing as mp
    import os
    env = {k: os.environ.get(k) for k in ("OMP_NUM_THREADS", "OPENBLAS_NUM_THREADS")}
    os.environ["OMP_NUM_THREADS"] = os.environ["OPENBLAS_NUM_THREADS"] = "1"
    try:
        with mp.get_context("spawn").Pool(min(procs, len(jobs))) as pool:
            return pool.map(_solve_c2_instance, jobs)
    finally:
        for k, v in env.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


def _solve_packed_call(job):
    """worker of solve_calls: one recorded closed-loop call (variant, N, m, x0, u0, xref, A, b, Ts, term, weights / bounds as SolverParams
    fields) by this file's method from the zero start -> (status, f, xopt, Ts_opt, restorations)"""
    import warnings
    warnings.filterwarnings("ignore", category=RuntimeWarning)
    from oracle.obca_nlp import Problem
    v, N, q, W, box, max_iter = job
    p = Problem(v, N, q["m"], q["x0"], q["u0"], q["xref"], q["A"], q["b"], q["Ts"], W[0], W[1][0], W[1][1], W[2], *box, term=q["term"] if v == 6 else None)
    r = solve(p, opts=dict(max_iter=max_iter) if max_iter else None)
    return int(r.status), float(r.f), r.xopt, float(r.Ts_opt), int(r.restorations)


def solve_calls(jobs, procs=1):
    """recorded calls (tuples as _solve_packed_call takes them) in `procs` spawned processes"""
    if procs <= 1:
        return [_solve_packed_call(j) for j in jobs]
    import multiprocessing as mp
    import os
    env = {k: os.environ.get(k) for k in ("OMP_NUM_THREADS", "OPENBLAS_NUM_THREADS")}
    os.environ["OMP_NUM_THREADS"] = os.environ["OPENBLAS_NUM_THREADS"] = "1"
    try:
        with mp.get_context("spawn").Pool(min(procs, len(jobs))) as pool:
            return pool.map(_solve_packed_call, jobs)
    finally:
        for k, v in env.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
