"""An INDEPENDENT solver on the reference-pinned model (oracle/obca_nlp.py): SciPy SLSQP with analytic Jacobians -- test
infrastructure, the strongest stand-in for the reference's `opti.solve()` (src/obca.py:1056) this image allows (casadi /
IPOPT are not installed).  Used three ways:

  polish(p, z)        started AT an answer of the product: a KKT point cannot be improved -- SLSQP must not find a feasible
                      point with an objective lower by more than 1e-6 (relative);
  from_starts(p, z)   started from the reference window as a trajectory and from two perturbations of it: does an
                      independent method reach the SAME optimum, another one, a better one?  (SLSQP cannot start from the
                      reference's own all-zero point: "singular matrix C", SURVEY Appendix C.)
  classify(p, z)      for an instance the product gave up on: any feasible point found (seven starts, four of them beside the
                      window) => "solver failure", none => "no feasible point found" (likely genuinely infeasible; not a proof).
Worker functions are module-level so that they can run in a process pool (spawn context: no HIP state is inherited)."""
import numpy as np
from scipy.optimize import minimize

FEAS_TOL = 1e-6
REL_F = 1e-6


def primal_violation(p, z):
    c, d = p.eq(z), p.ineq(z)
    lb, ub = p.ineq_bounds()
    return float(max(np.max(np.abs(c)), np.max(np.maximum(lb - d, 0)), np.max(np.maximum(d - ub, 0))))


def dual_guess(p, z):
    """lambda, mu of every (stage, obstacle) pair for given poses: the separating row with the largest gap, lambda on
    it scaled to ||A'lambda|| = 1, mu from the rotation equalities (a feasible dual point whenever the pose is clear)"""
    for k in range(p.N + 1):
        x, y, th = z[p.ip(k):p.ip(k) + 3]
        ct, st = np.cos(th), np.sin(th)
        t = np.array([x + ct * p.off, y + st * p.off])
        for i in range(p.nObs):
            o0, o1 = p.off_m[i], p.off_m[i + 1]
            A, b = p.A[k, o0:o1], p.b[k, o0:o1]
            nrm = np.linalg.norm(A, axis=1)
            R = np.array([[ct, st], [-st, ct]])
            gaps = []
            for j in range(o1 - o0):
                lam = np.zeros(o1 - o0)
                lam[j] = 1.0 / nrm[j]
                r = R @ (A.T @ lam)                     # mu0 - mu2 = -r0, mu1 - mu3 = -r1
                mu = np.array([max(-r[0], 0), max(-r[1], 0), max(r[0], 0), max(r[1], 0)])
                gaps.append((-(p.g @ mu) + (A[j] @ t - b[j]) / nrm[j], lam, mu))
            g, lam, mu = max(gaps, key=lambda q: q[0])
            z[p.il(k) + o0:p.il(k) + o1] = lam
            z[p.imu(k) + 4 * i:p.imu(k) + 4 * i + 4] = mu
    return z


def trajectory_start(p, pts):
    """primal vector from poses pts (3, N+1): inputs by differences, time scale for ~0.55 m/s, duals by dual_guess"""
    z = np.zeros(p.n)
    pts = np.array(pts, float)
    pts[:, 0] = p.x0
    if p.variant == 4:
        seg = np.hypot(*np.diff(pts[:2], axis=1))
        z[p.iT()] = min(max(1.0, seg.max() / (0.55 * p.Ts)), p.Tmax)
    h = p.Ts * (z[p.iT()] if p.variant == 4 else 1.0)
    for k in range(p.N + 1):
        z[p.ip(k):p.ip(k) + 3] = pts[:, k]
        if k < p.N:
            d = pts[:, k + 1] - pts[:, k]
            z[p.iu(k)] = np.clip(np.hypot(d[0], d[1]) / h, -0.6, 0.6)
            z[p.iu(k) + 1] = np.clip(d[2] / h, -np.pi / 6, np.pi / 6)
    return dual_guess(p, z)


def slsqp(p, z0, maxiter=400):
    lb, ub = p.ineq_bounds()
    hasL, hasU = np.isfinite(lb), np.isfinite(ub)

    def gfun(z):
        d = p.ineq(z)
        return np.concatenate([d[hasL] - lb[hasL], ub[hasU] - d[hasU]])

    def gjac(z):
        J = p.ineq(z, jac=True)[1]
        return np.vstack([J[hasL], -J[hasU]])
    cons = [dict(type="eq", fun=lambda z: p.eq(z), jac=lambda z: p.eq(z, jac=True)[1]),
            dict(type="ineq", fun=gfun, jac=gjac)]
    try:
        r = minimize(lambda z: p.objective(z, grad=True), np.asarray(z0, float)[:p.n].copy(), jac=True, constraints=cons,
                     method="SLSQP", options=dict(maxiter=maxiter, ftol=1e-12))
        return dict(z=r.x, f=float(r.fun), viol=primal_violation(p, r.x), nit=int(r.nit))
    except Exception as e:            # noqa: BLE001
        return dict(z=np.asarray(z0, float)[:p.n], f=np.inf, viol=np.inf, nit=-1, err=repr(e))


def duality_gap(p, z, y):
    """sum of the complementarity products of the certificate (multipliers y as the C ABI hands them out, kernel row order):
    the first-order estimate of how far an interior-point answer -- which stops at a barrier parameter > 0 -- sits above the
    optimum it converges to"""
    from tests import kkt_check
    z = np.asarray(z, float)[:p.n]
    _, yin = kkt_check.split_duals(p, y)
    d = p.ineq(z)
    lb, ub = p.ineq_bounds()
    sU = np.where(np.isfinite(ub), np.maximum(ub - d, 0.0), 0.0)
    sL = np.where(np.isfinite(lb), np.maximum(d - lb, 0.0), 0.0)
    return float(np.sum(np.maximum(yin, 0.0) * sU + np.maximum(-yin, 0.0) * sL))


def polish(job):
    """job = (p, z of the product's answer[, y its multipliers]): SLSQP from that point.  improved: it found a FEASIBLE point
    whose objective is lower than the answer's by more than REL_F (relative to max(1, |f|)) plus twice the answer's own
    duality-gap estimate -- a KKT point of the pinned model does not allow that"""
    p, z = job[0], job[1]
    gap = 0.0
    if len(job) > 2 and job[2] is not None:                 # multipliers in kernel order, or the gap itself
        gap = float(job[2]) if np.ndim(job[2]) == 0 else duality_gap(p, z, job[2])
    f0 = float(p.objective(np.asarray(z, float)[:p.n]))
    r = slsqp(p, z, maxiter=100)
    better = r["viol"] <= FEAS_TOL and r["f"] < f0 - REL_F * max(1.0, abs(f0)) - 2.0 * gap
    return dict(f0=f0, f=r["f"], viol=r["viol"], nit=r["nit"], gap=gap, improved=bool(better))


def from_starts(job):
    """job = (p, z of the product's answer, seed): SLSQP from the reference window and two perturbations of it (poses +-
    0.3 m / 0.15 rad).  Per start: 'same' (feasible, objective within 1e-4 relative of the product's), 'better' (feasible,
    lower by more than that), 'other' (feasible, higher), 'none' (no feasible point reached)"""
    p, z, seed = job
    f0 = float(p.objective(np.asarray(z, float)[:p.n]))
    rng = np.random.default_rng(seed)
    out = []
    for j in range(3):
        pts = p.xref.copy()
        if j:
            pts[:2, 1:] += rng.uniform(-0.3, 0.3, size=(2, p.N))
            pts[2, 1:] += rng.uniform(-0.15, 0.15, size=p.N)
            if p.variant == 4:
                pts[:, -1] = p.xref[:, -1]
        r = slsqp(p, trajectory_start(p, pts))
        tol = 1e-4 * max(1.0, abs(f0))
        kind = "none" if r["viol"] > FEAS_TOL else "same" if abs(r["f"] - f0) <= tol else "better" if r["f"] < f0 else "other"
        out.append(dict(kind=kind, f=r["f"], viol=r["viol"], nit=r["nit"]))
    return dict(f0=f0, starts=out)


def lateral_points(p, d, ramp=3):
    """the reference window moved sideways by d metres (ramped in over `ramp` stages), headings along the moved poses -- a start in
    the basin of a plan AROUND what the window runs into (round 5: the window, the straight line and the solver's last iterate
    are all symmetric about a head-on obstacle, and SLSQP from them called steps 21-25 of the reference's demo11 run infeasible,
    which they are not)"""
    base = p.xref.copy().astype(float)
    base[:, 0] = p.x0
    out = base.copy()
    for k in range(1, p.N + 1):
        a = base[:2, min(k + 1, p.N)] - base[:2, k - 1]
        n = np.array([-a[1], a[0]]) / max(np.hypot(*a), 1e-9)
        out[:2, k] = base[:2, k] + d * min(1.0, k / ramp) * n
    for k in range(1, p.N):
        dd = out[:2, k + 1] - out[:2, k]
        out[2, k] = np.arctan2(dd[1], dd[0])
    out[2, p.N] = out[2, p.N - 1]
    return out


def classify(job):
    """job = (p, last iterate of the product, tag): a feasible point from the window, a straight line, the last iterate, or the
    window moved 1.5 / 3 m to either side"""
    p, z_last, tag = job[:3]
    kinds = job[3] if len(job) > 3 else ("window", "line", "last_iterate", "right 1.5", "left 1.5", "right 3", "left 3")
    best = None
    for kind in kinds:
        if kind == "window":
            z0 = trajectory_start(p, p.xref)
        elif kind == "line":
            end = p.xref[:, p.N].copy()
            if p.variant == 6:
                end[0] = max(end[0], p.term[0] + 0.1)
            z0 = trajectory_start(p, np.linspace(p.x0, end, p.N + 1).T)
        elif kind == "last_iterate":
            z0 = np.asarray(z_last, float)[:p.n].copy()
        else:
            side, d = kind.split()
            z0 = trajectory_start(p, lateral_points(p, (-1.0 if side == "right" else 1.0) * float(d)))
        r = slsqp(p, z0, maxiter=250)
        cand = dict(start=kind, viol=r["viol"], f=r["f"], nit=r["nit"])
        if best is None or (cand["viol"] <= FEAS_TOL and (best["viol"] > FEAS_TOL or cand["f"] < best["f"])) or \
                (best["viol"] > FEAS_TOL and cand["viol"] < best["viol"]):
            best = cand
        if best["viol"] <= FEAS_TOL:
            break                          # one feasible point settles the classification
    best["tag"] = tag
    best["feasible_point_found"] = bool(best["viol"] <= FEAS_TOL)
    return best


def classify_stopped_world(job):
    """job = (world index, n_dyn, steps the device rollout completed): the closed loop of C5 world i replayed on the host
    (Python mirror of the reference's loop driving the CPU build of the structured core -- same iterates as the device);
    the solve on which the rollout stopped is handed to classify().  Returns the classification, or a note when the replay
    does not stop at the same step (roundoff-sensitive path)."""
    i, n_dyn, steps_dev = job
    from oracle.obca_nlp import Problem
    from tests import native_build
    from vehicle_motion_planning_with_obstacles_avoidance_using_mpc_amd import scenarios as sc
    from vehicle_motion_planning_with_obstacles_avoidance_using_mpc_amd.closed_loop import closedLoop
    from vehicle_motion_planning_with_obstacles_avoidance_using_mpc_amd.solver import SolverParams
    sp = SolverParams()
    s = native_build.LpiObca()
    cl = closedLoop(sc.make_world_c5(i, n_dyn=n_dyn), solver=s)
    cl.N_free = cl.N_fix = 5
    cl.closed_loop_mpc4()
    c = s.calls[-1]
    if c["status"] in (0, 1) or cl.k != steps_dev:
        return dict(tag=(i, cl.k, steps_dev), replay_differs=True, feasible_point_found=False)
    v = c["variant"]
    W = (sp.Q_free, sp.R_free, sp.P_free) if v == 4 else (sp.Q_fix, sp.R_fix, sp.P_fix)
    p = Problem(v, c["xref"].shape[1] - 1, c["m"], c["x0"], c["u0"], c["xref"], c["A"], c["b"], c["Ts"], W[0], W[1][0], W[1][1], W[2],
                sp.xL, sp.xU, sp.uL, sp.uU, sp.ego, sp.dmin, term=c["term"] if v == 6 else None)
    r = classify((p, np.zeros(p.n), (i, cl.k, v, c["status"])))
    r["replay_differs"] = False
    return r


def pool_map(fn, jobs, procs):
    """fn over jobs in `procs` worker PROCESSES started from scratch (``python -m tests.independent``: never a fork of a
    process that holds a HIP context, no dependence on how the caller's __main__ was started); serial for one process"""
    if procs <= 1 or len(jobs) <= 1:
        return [fn(j) for j in jobs]
    import os
    import pickle
    import subprocess
    import sys
    import tempfile
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    n = min(procs, len(jobs))
    with tempfile.TemporaryDirectory() as tmp:
        running = []
        for w in range(n):
            fin, fout = os.path.join(tmp, "in%d.pkl" % w), os.path.join(tmp, "out%d.pkl" % w)
            with open(fin, "wb") as fh:
                pickle.dump(jobs[w::n], fh)
            running.append((w, fout, subprocess.Popen([sys.executable, "-m", "tests.independent", fn.__name__, fin, fout], cwd=root,
                                                      env=dict(os.environ, OMP_NUM_THREADS="1", OPENBLAS_NUM_THREADS="1"))))
        out = [None] * len(jobs)
        for w, fout, pr in running:
            if pr.wait() != 0:
                raise RuntimeError("independent-solver worker failed")
            with open(fout, "rb") as fh:
                out[w::n] = pickle.load(fh)
    return out


def summarise(polished, started):
    """shares for the bench line / test messages"""
    kinds = [s["kind"] for r in started for s in r["starts"]]
    n = max(1, len(kinds))
    return dict(instances=len(polished), improved_from_the_answer=int(sum(r["improved"] for r in polished)),
                starts=len(kinds), same_optimum=kinds.count("same") / n, other_optimum=kinds.count("other") / n,
                better_optimum=kinds.count("better") / n, no_feasible_point=kinds.count("none") / n)


if __name__ == "__main__":            # worker of pool_map
    import pickle
    import sys
    with open(sys.argv[2], "rb") as fh:
        todo = pickle.load(fh)
    res = [globals()[sys.argv[1]](j) for j in todo]
    with open(sys.argv[3], "wb") as fh:
        pickle.dump(res, fh)
